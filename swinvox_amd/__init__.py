"""swinvox_amd: the SwinVox Encoder -> Decoder -> Merger -> Refiner forward/backward path as hand-written HIP
kernels for MI355X (gfx950) behind the reference's nn.Module surface.  Importing never touches the GPU."""
from .config import Cfg, cfg, default_cfg  # noqa: F401
from .ops import get_math, get_storage, set_attention_fp8, set_fused_mlp, set_math, set_overlap, set_storage  # noqa: F401
from . import models  # noqa: F401
