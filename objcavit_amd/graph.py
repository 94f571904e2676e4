"""hipGraph replay of the forward (BASELINE config 5 asks for a "hipGraph-captured forward").

Every C-ABI entry point only enqueues on the caller's stream (no allocation, no sync), so the whole forward
(~280 launches at bs = 16, all of them our own kernels plus a handful of element-wise glue ops) is capturable with
``torch.cuda.CUDAGraph`` (= hipGraph on ROCm).  Measured on MI355X / ROCm 7.2 (bench.py): replay 24.2 ms per step
against 25.9 ms for eager dispatch -- the ~1 ms the GPU idles between dependent launches of the eager stream.  (While
the decoder still ran on MIOpen / ATen kernels replay was SLOWER than eager, 77 vs 51 ms; that went away with them.)

``GraphedGraphBins`` captures ``GraphBins.forward_until_head`` (shapes fixed by the example image; object boxes /
features come from the model's provider and are baked in as static device tensors) and runs the fused bin-head
kernel eagerly after each replay.  Launches named in ``eager_ops`` (timing names of hip_ops, e.g.
``"conv3x3|16,240,320,280,128"``) are kept out of the graph as EAGER ISLANDS: capture ends in front of them and a new
graph segment starts behind them, so a step is  segment, island, segment, ..., head  -- that is how bench.py times
its roofline kernel live with HIP events inside the timed region (events recorded inside a captured graph cannot be
read back on ROCm 7.2).
"""
from __future__ import annotations

import threading
import warnings
from typing import Callable, List, Sequence, Tuple, Union

import torch

from . import hip_ops

# Stream capture is a process-wide affair on ROCm 7.2: two threads capturing at the same time abort inside capture_end
# (measured: tests, round 3), whatever capture_error_mode says.  Captures are therefore serialised; everything a capture
# hands to hip_ops (island hook, scratch store) is thread-local state, so a thread that merely LAUNCHES meanwhile is safe.
_CAPTURE_LOCK = threading.Lock()


class GraphedGraphBins:
    def __init__(self, model, example_image: torch.Tensor, warmup: int = 2, eager_ops: Sequence[str] = ()):
        if example_image.device.type != "cuda":
            raise RuntimeError("graph capture needs a GPU tensor")
        self.model = model
        self.static_image = example_image.clone()
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        # the graph OWNS its scratch: every workspace requested during warm-up, capture and replay comes from this
        # store, so no eager call or later capture at other shapes can free a buffer whose address is baked in here
        self.scratch = hip_ops.WorkspaceStore()
        with hip_ops.workspace_scope(self.scratch), torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):                      # sizes every workspace / weight cache before capture
                model(self.static_image)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()

        self.segments: List[Union[torch.cuda.CUDAGraph, Tuple[str, Callable]]] = []
        pool = torch.cuda.graph_pool_handle()
        state = {"g": None}

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: calls made by OTHER threads while we capture (the RCCL watchdog of a data-parallel job
            # polls events) must not invalidate the capture
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            state["g"] = g

        def end():
            # a segment without a single node (two adjacent islands, or nothing behind the last one) is DROPPED: torch
            # reports it with a warning at capture_end, and replaying an empty hipGraph on every step is pure overhead
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                state["g"].capture_end()
            empty = any("empty" in str(w.message).lower() for w in seen)
            for w in seen:
                if "empty" not in str(w.message).lower():
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            if empty:
                self.empty_segments_dropped += 1
            else:
                self.segments.append(state["g"])
            state["g"] = None

        def on_break(name, call):
            end()
            with hip_ops.timed(name):
                call()                                   # the capture run executes the island once, eagerly
            self.segments.append((name, call))
            begin()

        self.empty_segments_dropped = 0
        # the hook is an object handed to hip_ops for the duration of THIS capture on THIS thread (thread-local scope)
        with _CAPTURE_LOCK, hip_ops.island_scope(hip_ops.IslandHook(eager_ops, on_break)), hip_ops.workspace_scope(self.scratch), \
                torch.cuda.stream(self.stream), torch.no_grad():
            begin()
            parts = model.forward_until_head(self.static_image)
            end()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.scratch.freeze()
        self.feat, self.queries, self.centers, self.bin_edges, self.detections = parts
        self.ReturnType = model.ReturnType
        self.islands = [s[0] for s in self.segments if isinstance(s, tuple)]

    @torch.no_grad()
    def __call__(self, image: torch.Tensor):
        if image.shape != self.static_image.shape:
            raise ValueError(f"captured for {tuple(self.static_image.shape)}, got {tuple(image.shape)}")
        if image.data_ptr() != self.static_image.data_ptr():
            self.static_image.copy_(image)
        with hip_ops.workspace_scope(self.scratch):
            for seg in self.segments:
                if isinstance(seg, tuple):
                    with hip_ops.timed(seg[0]):
                        seg[1]()
                else:
                    seg.replay()
            depth = self.model.head(self.feat, self.queries, self.centers)
        return self.ReturnType(depth_pred=depth, bin_edges=self.bin_edges, detections=self.detections)
