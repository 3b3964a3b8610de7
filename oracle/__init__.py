"""CPU oracle for the SwinVox hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain-PyTorch fp32 CPU restatement of the reference's
Encoder -> Decoder -> Merger -> Refiner forward/backward path.  It exists only
so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
check / time the HIP path against it.  Nothing under swinvox_amd/ may import
it; the product path fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * Decoder / Merger / Refiner / CrossViewAttention: pinned against the
    reference's own modules imported from /root/reference in the build
    container (tests/golden/make_golden.py), max|diff| recorded in
    tests/golden/manifest.json.
  * Encoder plumbing (neck, stage heads, fusion): pinned against the reference
    models/encoder.py + models/swin_transformer.py imported with the oracle's
    own backbones standing in for the absent timm/torchvision packages.
  * Swin-T arithmetic (timm 1.0.15, absent from /root/reference): pinned
    against transformers' independent SwinModel built from a local config, and
    by the reference notebook's parameter-count KAT and state-dict key list.
  * ResNet-50[:layer3] arithmetic (torchvision 0.21, absent): parameter-count
    KAT + key list only -> "parity unpinned" for that branch's arithmetic.
"""
from .model import (  # noqa: F401
    Cfg, default_cfg, Encoder, Decoder, Merger, Refiner, SwinTransformer,
    CrossViewAttention, SwinBackbone, ResNetTrunk, init_weights, calibrate_,
    seeded_weights_, bce_logits, iou_at_thresholds, fscore_at_thresholds, train_step_loss,
)
