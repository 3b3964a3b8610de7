"""Config tree mirroring reference config.py:8-142 (hot-path knobs NETWORK.* :83-94, CONST :62-65, TRAIN gates
:109-122, TEST.VOXEL_THRESH :134).  `easydict` is not a dependency: Cfg is a small attribute-access dict, so a
reference-style `cfg.NETWORK.USE_MERGER` works unchanged, and an EasyDict built by the reference's own config.py can
be passed to every module as is."""
from __future__ import annotations


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg() -> Cfg:
    c = Cfg()
    c.DATASET = Cfg(MEAN=[0.5, 0.5, 0.5], STD=[0.5, 0.5, 0.5])                                     # config.py:48-49
    c.CONST = Cfg(DEVICE="0", RNG_SEED=0, IMG_W=224, IMG_H=224, BATCH_SIZE=32, N_VIEWS_RENDERING=1, CROP_IMG_W=128, CROP_IMG_H=128)
    c.NETWORK = Cfg(LEAKY_VALUE=0.2, TCONV_USE_BIAS=False, USE_REFINER=True, USE_MERGER=True, USE_SWIN_T_MULTI_STAGE=True,
                    SWIN_T_STAGES=[0, 1, 2, 3], USE_CROSS_VIEW_ATTENTION=True, CROSS_ATT_REDUCTION_RATIO=4,
                    ATT_SPATIAL_DOWNSAMPLE_RATIO=2, CROSS_ATT_NUM_HEADS=4)
    c.TRAIN = Cfg(POLICY="adam", EPOCH_START_USE_REFINER=0, EPOCH_START_USE_MERGER=0,
                  ENCODER_LEARNING_RATE=3.834299021554089e-06, DECODER_LEARNING_RATE=2.4966084898328403e-05,
                  REFINER_LEARNING_RATE=1.6418272442716922e-06, MERGER_LEARNING_RATE=0.00022177181973320365,
                  BETAS=(0.8500000000000001, 0.993), MOMENTUM=0.9, GAMMA=0.8830819189779433,
                  WEIGHT_DECAY=0.0003370779562775397, ENCODER_LR_MILESTONES=[150], DECODER_LR_MILESTONES=[150],
                  REFINER_LR_MILESTONES=[150], MERGER_LR_MILESTONES=[150],
                  BRIGHTNESS=0.13746317606570424, CONTRAST=0.3365401951623921, SATURATION=0.20370660036548005,   # config.py:103-107
                  NOISE_STD=0.0850409938037522, RANDOM_BG_COLOR_RANGE=[[225, 255], [225, 255], [225, 255]])
    c.TEST = Cfg(VOXEL_THRESH=[0.2, 0.3, 0.4, 0.5], RANDOM_BG_COLOR_RANGE=[[240, 240], [240, 240], [240, 240]])      # config.py:131-134
    return c


cfg = default_cfg()
