"""hipGraph capture of the per-step network evaluation.

One VideoUNet evaluation is ~660 kernel launches of 20-400 us each; launched one by one from Python/ctypes the GPU idles
~3.3 ms per evaluation (4 % of the step, profiles/r01b_kernel_stats_v2.txt).  The evaluation is shape-static and free of
host synchronisation (every scalar the kernels need lives in device memory), so it is captured once into a HIP graph and
replayed for the remaining 24 sampler steps (and for every later sample of the same shape).

`graphed(fn)` wraps any callable `fn(*tensors_or_dicts_of_tensors) -> tensor`:
  * first call per input signature (shapes / dtypes / dict keys): inputs are copied into static buffers, `fn` runs once
    un-captured on a side stream (warm-up: lazy allocations, library loading), then once under capture;
  * later calls copy the inputs into the static buffers and replay.
The returned tensor is the graph's static output buffer: it is overwritten by the next call (the sampler consumes it
before the next evaluation; clone it to keep it).

The ctypes launches go to `torch.cuda.current_stream()`, which is the capture stream inside `torch.cuda.graph`, and the
activation buffers come from torch's caching allocator, which serves captures from a private pool - nothing else is
needed for the HIP kernels to be captured.

Measured on MI355X / ROCm 7.2 (bench.py --graph): 9.59 frames/s with replay vs 9.62 without - graph replay does not close
the gaps between dependent kernels, so callers keep it opt-in (V3D_GRAPH=1 or enabled=True).
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, Tuple

import torch


def _flatten(args) -> Tuple[list, Any]:
    """Tensors of a nested (tuple / list / dict) argument structure in a fixed order + a hashable structure key."""
    flat, key = [], []

    def walk(a):
        if isinstance(a, torch.Tensor):
            flat.append(a)
            key.append(("T", tuple(a.shape), str(a.dtype), str(a.device)))
        elif isinstance(a, dict):
            key.append(("D", tuple(sorted(a))))
            for k in sorted(a):
                walk(a[k])
        elif isinstance(a, (tuple, list)):
            key.append(("L", len(a)))
            for v in a:
                walk(v)
        else:
            key.append(("C", a if isinstance(a, (int, float, str, bool, type(None))) else id(a)))

    walk(args)
    return flat, tuple(key)


def _rebuild(args, it):
    if isinstance(args, torch.Tensor):
        return next(it)
    if isinstance(args, dict):
        return {k: _rebuild(args[k], it) for k in sorted(args)}
    if isinstance(args, (tuple, list)):
        return type(args)(_rebuild(v, it) for v in args)
    return args


class _Captured:
    def __init__(self, fn, args, flat):
        self.static_in = [t.clone() for t in flat]
        self.static_args = _rebuild(args, iter(self.static_in))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(*self.static_args)            # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: the library keeps its split-K / stream-K workspaces per stream and cannot allocate
        # inside a capture - on a fresh stream those launches would fall back to their unsplit forms (other summation order: the replay
        # would differ from eager in the last bits)
        with torch.cuda.graph(self.graph, stream=side):
            self.out = fn(*self.static_args)

    def __call__(self, flat):
        for dst, src in zip(self.static_in, flat):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.out


def graphed(fn: Callable, enabled: bool | None = None) -> Callable:
    """Wrap `fn` so that repeated calls with the same input signature replay a captured HIP graph."""
    if enabled is None:
        enabled = os.environ.get("V3D_GRAPH", "0") not in ("", "0")
    if not enabled:
        return fn
    cache: Dict[Any, _Captured] = {}

    def call(*args):
        flat, key = _flatten(args)
        if not flat or not all(t.is_cuda for t in flat):
            return fn(*args)
        cap = cache.get(key)
        if cap is None:
            cap = cache[key] = _Captured(fn, args, flat)
        return cap(flat)

    call.graph_cache = cache
    call.__wrapped__ = fn
    return call
