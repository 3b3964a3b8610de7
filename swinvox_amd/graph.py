"""hipGraph replay of a whole step of the hot path.

One forward + loss + backward of the four modules is ~1 300 kernel launches (on three HIP streams when the branches overlap);
enqueueing them through ctypes costs ~21 ms of host time per step.  `GraphedStep` captures the launch sequence of one step into ONE
hipGraph and replays it: the host cost of a step becomes one hipGraphLaunch (0.4 ms).  It pays when the host is the bottleneck
(small per-GPU batches); at the benchmark batch the GPU needs ~70 ms per step and the eager three-stream path is faster (numbers
in __init__).

The captured step is the same code path as the eager one (the modules do not know about the capture).  What a capture
freezes, and how it is handled:
  * buffer addresses: inputs are copied into static tensors (`copy_inputs`), outputs and parameter gradients are static
    tensors of the graph's private memory pool (set `p.grad = None` inside `fn`, as core/train.py:265-268 zero_grad does,
    so the gradients are allocated inside the pool);
  * scalar kernel arguments: dropout / drop-path seeds are mixed with a device word (`ops.set_seed_epoch`) that the graph
    itself advances at the start of every replay, so each replay draws new masks;
  * host decisions (which kernels, which shapes): fixed - build one GraphedStep per (shape, mode, configuration).
Nothing may synchronise with the host inside `fn` (no .item(), no host-side reads): the reference's per-step `.item()` calls
(core/train.py:300-306) belong after the replay.

The process that captures only replays afterwards (no re-exec of a GPU-initialised process is involved anywhere).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import hip, ops


class GraphedStep:
    def __init__(self, fn: Callable[[], object], static_inputs: Sequence[torch.Tensor] = (), warmup: int = 3,
                 device: Optional[torch.device] = None, single_stream: bool = True):
        """fn(): one step on the static inputs (closure); its return value (tensor / tuple of tensors / None) is kept as the
        static output.  `warmup` eager calls run first on a side stream (first-use paths: weight-pack registration, arena
        sizing, lazy stream creation must not be captured)."""
        if hip.TRACE is not None:
            raise RuntimeError("GraphedStep: timing events (hip.TRACE) cannot be recorded inside a capture")
        self.fn = fn
        self.static_inputs = list(static_inputs)
        dev = device or (self.static_inputs[0].device if self.static_inputs else torch.device("cuda", torch.cuda.current_device()))
        self.device = dev
        self.epoch = torch.zeros(1, dtype=torch.int32, device=dev)
        prev = ops._STATE.get("seed_epoch")
        prev_overlap = ops.overlap_enabled()
        ops.set_seed_epoch(self.epoch)
        # single_stream (default): the capture runs with set_overlap(False).  Measured on ROCm 7.2 / MI355X at B = 32 x V = 8: eager on three
        # streams 72.7 ms per step (host 21 ms with an idle queue), a graph captured across the three streams 96.8 ms (fork / join edges
        # replay slower than eager launches), a single-stream graph 77.6 ms at 0.4 ms of host time - the form to use when the host is
        # the bottleneck (small batches).
        if single_stream:
            ops.set_overlap(False)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self.epoch.add_(1)
                    fn()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.epoch.add_(1)              # captured: every replay advances the seed word first
                self.output = fn()
        finally:
            ops.set_seed_epoch(prev)
            ops.set_overlap(prev_overlap)
        self.replays = 0

    def copy_inputs(self, *tensors: torch.Tensor) -> None:
        if len(tensors) != len(self.static_inputs):
            raise ValueError(f"GraphedStep: expected {len(self.static_inputs)} inputs")
        for dst, src in zip(self.static_inputs, tensors):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("GraphedStep: input shape / dtype differs from the captured one - capture a new GraphedStep")
            dst.copy_(src, non_blocking=True)

    def __call__(self, *tensors: torch.Tensor):
        """Replay (after copying `tensors` into the static inputs, when given); returns the static output."""
        if tensors:
            self.copy_inputs(*tensors)
        self.graph.replay()
        self.replays += 1
        return self.output
