"""Drop-in counterparts of the reference `models` package (same class names, constructor and forward signatures)."""
from .encoder import Encoder  # noqa: F401
from .decoder import Decoder  # noqa: F401
from .merger import Merger  # noqa: F401
from .refiner import Refiner  # noqa: F401
from .swin_transformer import SwinTransformer  # noqa: F401
from .cross_view_attention import CrossViewAttention  # noqa: F401
