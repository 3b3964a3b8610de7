"""Counterpart of the reference's utils/helpers.py for the hot path: the weight recipe every run starts from.

`init_weights` follows utils/helpers.py:20-44 (applied to all four nets by core/train.py:91-94, the pretrained
backbones included): Conv*/ConvTranspose* weights Kaiming-normal (fan_out, leaky_relu, a = 0.02) scaled by 0.1 with zero
bias, BatchNorm gain 1 / bias 0, Linear weights N(0, 0.01) scaled by 0.1 with zero bias.  It is dispatch by
`isinstance` on stock torch holders, so `net.apply(init_weights)` works on the HIP-backed modules exactly as on the
reference's; the draws come from torch's global generator in module-registration order, so with the same seed the two
model families start from identical weights (tests/test_cpu_oracle_and_abi.py::test_init_weights_matches_reference_recipe).
`var_or_cuda` (utils/helpers.py:15-18) and `count_parameters` (:46-47) are the two other helpers the hot loops call.
"""
from __future__ import annotations

import torch

_CONVS = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d)


def init_weights(m: torch.nn.Module) -> None:
    if isinstance(m, _CONVS):
        torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu", a=0.02)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)
        m.weight.data *= 0.1
    elif isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
        torch.nn.init.constant_(m.weight, 1)
        torch.nn.init.constant_(m.bias, 0)
    elif isinstance(m, torch.nn.Linear):
        torch.nn.init.normal_(m.weight, 0, 0.01)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)
        m.weight.data *= 0.1


def var_or_cuda(x: torch.Tensor) -> torch.Tensor:
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x


def count_parameters(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())
