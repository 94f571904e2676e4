"""hipGraph replay of the forward (BASELINE config 5 asks for a "hipGraph-captured forward").

Every C-ABI entry point only enqueues on the caller's stream (no allocation, no sync), so the whole forward
(~280 launches at bs = 16, all of them our own kernels plus a handful of element-wise glue ops) is capturable with
``torch.cuda.CUDAGraph`` (= hipGraph on ROCm).  Measured on MI355X / ROCm 7.2 (bench.py): replay 24.2 ms per step
against 25.9 ms for eager dispatch -- the ~1 ms the GPU idles between dependent launches of the eager stream.  (While
the decoder still ran on MIOpen / ATen kernels replay was SLOWER than eager, 77 vs 51 ms; that went away with them.)

``GraphedGraphBins`` captures ``GraphBins.forward_until_head`` (shapes fixed by the example image) and runs the fused bin-head
kernel eagerly after each replay.  Objects: by default whatever the model's provider returned at capture time is baked in as
static device tensors (the benchmark's synthetic objects).  With ``object_capacity = N`` the graph owns static PADDED object
buffers ([B, N, 512] features, [B, N, 4] boxes, int32 counts[B]) and every replay takes LIVE objects -- ``g(image, features_list,
xywh_list)`` / ``g(image, PaddedObjects)`` / the provider's output for ``image`` -- copied into those buffers first: the masks and
the front-padding of the key rows are formed on the device from ``counts`` (csrc/objects_pad.hip), nothing on the captured path
depends on a count on the host, so ONE capture per (B, N) serves any ragged object set (BASELINE configs[4]: a detector + CLIP
in front of a hipGraph-captured forward, modules/GraphBins.py:90-107, modules/ObjCAViT.py:311-330,180-194).  Launches named in ``eager_ops`` (timing names of hip_ops, e.g.
``"conv3x3|16,240,320,280,128"``) are kept out of the graph as EAGER ISLANDS: capture ends in front of them and a new
graph segment starts behind them, so a step is  segment, island, segment, ..., head  -- that is how bench.py times
its roofline kernel live with HIP events inside the timed region (events recorded inside a captured graph cannot be
read back on ROCm 7.2).
"""
from __future__ import annotations

import contextlib
import ctypes
import threading
import warnings
from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch

from . import graph_topology, hip_ops
from .modules.ObjCAViT import PaddedObjects

# Stream capture is a process-wide affair on ROCm 7.2: two threads capturing at the same time abort inside capture_end
# (measured: tests, round 3), whatever capture_error_mode says.  Captures are therefore serialised; everything a capture
# hands to hip_ops (island hook, scratch store) is thread-local state, so a thread that merely LAUNCHES meanwhile is safe.
_CAPTURE_LOCK = threading.Lock()


_HIP = None


def _graph_node_count(g: "torch.cuda.CUDAGraph") -> Optional[int]:
    """Number of nodes of a captured (kept, not yet instantiated) graph, asked of the HIP runtime itself
    (hipGraphGetNodes); None when the runtime library or the raw handle is not reachable."""
    global _HIP
    try:
        if _HIP is None:
            _HIP = ctypes.CDLL("libamdhip64.so")
            _HIP.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
            _HIP.hipGraphGetNodes.restype = ctypes.c_int
        n = ctypes.c_size_t(0)
        if _HIP.hipGraphGetNodes(ctypes.c_void_p(int(g.raw_cuda_graph())), None, ctypes.byref(n)) != 0:
            return None
        return int(n.value)
    except (OSError, AttributeError, RuntimeError, TypeError):
        return None


class GraphedGraphBins:
    images_are_independent = True      # an image's result does not depend on its batch mates (per object group: SURVEY.md Q3)

    def __init__(self, model, example_image: torch.Tensor, warmup: int = 2, eager_ops: Sequence[str] = (),
                 object_capacity: Optional[int] = None, object_group: Optional[int] = None, check_topology: bool = True,
                 pairs: Optional[str] = None, in_flight: int = 1, stream: Optional["torch.cuda.Stream"] = None):
        """``stream``: the stream to capture on and replay on (default: a new one).  Callers that keep several graphs in flight pass
        streams from ``hip_ops.independent_streams`` -- two slots whose streams share a hardware queue serialise.
        ``in_flight``: how many batches the caller keeps in flight on this GPU (one graph per slot): the capture forks side streams
        inside the forward for a lone batch only (hip_ops.batches_in_flight).  ``pairs``: None = the element type the model's decoder settled on (fp16 pairs unless its weights / first batch said
        otherwise); "bf16" = warm up and capture under ``hip_ops.bf16_pairs()`` (fp32's range: what ``rerun_on_bf16`` replays).
        ``check_topology``: every captured segment is read back from the runtime and compared with the only shape the product's
        forks produce and every measurement was taken on: a chain of single fork / single join diamonds (objcavit_amd/
        graph_topology.py).  Another shape is REPORTED (RuntimeWarning + ``hip_ops.ROUTE_REPORT["graph_topology"]``) when the process
        runs on at most four hardware queues -- there every shape tried replays at full speed (profiles/r05_graph_shapes.txt) -- and
        RAISES, before anything is replayed, on more than four: that is the configuration in which forked graphs replayed 3.5 - 6 ms
        slower per step and one hipGraphLaunch crashed in round 4."""
        if example_image.device.type != "cuda":
            raise RuntimeError("graph capture needs a GPU tensor")
        if pairs not in (None, "bf16"):
            raise ValueError("pairs must be None (the model's own decision) or 'bf16'")
        self.model = model
        self.pairs = pairs
        self.in_flight = max(1, int(in_flight))
        self._ctor = dict(warmup=warmup, object_capacity=object_capacity, object_group=object_group, check_topology=check_topology,
                          in_flight=self.in_flight)
        self._fallback: Optional["GraphedGraphBins"] = None
        # fp16 range guard (hip_ops.RangeGuard): the word every fp16-pair producer of THIS graph's launches ORs into; taken behind
        # every replay into ``last_flag`` (device), read by ``tripped`` / ``checked`` where the caller reads results
        self.range_guard = hip_ops.RangeGuard(example_image.device)
        self.last_flag: Optional[torch.Tensor] = None
        self.trips = 0
        self.static_image = example_image.clone()
        self.object_group = object_group
        self.objects: Optional[PaddedObjects] = None
        if object_capacity is not None:
            B, dev = example_image.shape[0], example_image.device
            cap, fdim = int(object_capacity), int(model.objcavit.obj_feature_dim)
            if cap < 1:
                raise ValueError("object_capacity must be >= 1")
            self.objects = PaddedObjects(torch.zeros(B, cap, fdim, device=dev), torch.full((B, cap, 4), -1.0, device=dev),
                                         torch.ones(B, dtype=torch.int32, device=dev))
            self.load_objects(None, None)          # the provider's objects for the example image: warm-up on real values
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        # the graph OWNS its scratch: every workspace requested during warm-up, capture and replay comes from this
        # store, so no eager call or later capture at other shapes can free a buffer whose address is baked in here
        self.scratch = hip_ops.WorkspaceStore()
        with self._route(), hip_ops.batches_in_flight(self.in_flight), hip_ops.workspace_scope(self.scratch), \
                torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):                      # sizes every workspace / weight cache before capture
                model(self.static_image, self.objects, None, None, self.object_group)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()

        self.segments: List[Union[torch.cuda.CUDAGraph, Tuple[str, Callable]]] = []
        pool = torch.cuda.graph_pool_handle()
        state = {"g": None}

        def begin():
            # keep_graph: the captured hipGraph_t stays alive behind capture_end, so its node count can be read from the
            # runtime before it is instantiated
            g = torch.cuda.CUDAGraph(keep_graph=True)
            # thread_local: calls made by OTHER threads while we capture (the RCCL watchdog of a data-parallel job
            # polls events) must not invalidate the capture
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            state["g"] = g

        def end():
            # a segment without a single node (two adjacent islands, or nothing behind the last one) is DROPPED: replaying an
            # empty hipGraph on every step is pure overhead.  Emptiness is read from the capture itself (hipGraphGetNodes);
            # torch's "graph is empty" warning -- process-global warning state, wording of one torch version -- is only the
            # fallback when the runtime cannot be asked.
            g = state["g"]
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                g.capture_end()
            for w in seen:
                if "empty" not in str(w.message).lower():
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            n = _graph_node_count(g)
            empty = n == 0 if n is not None else any("empty" in str(w.message).lower() for w in seen)
            if empty:
                self.empty_segments_dropped += 1
            else:
                topo = graph_topology.read(g.raw_cuda_graph(), kernel_names=False) if check_topology else None
                if topo is not None:
                    bad = graph_topology.check(topo)
                    self.segment_topology.append(topo.summary())
                    if bad:
                        what = ("captured forward is not a chain of single fork / single join branches (" + "; ".join(bad) + "): no "
                                "measurement of this project covers that shape (profiles/r05_graph_shapes.txt)")
                        if hip_ops.hw_queues_allow_forks():
                            # <= 4 hardware queues: every fork shape tried replays at full speed there -- report, do not refuse
                            hip_ops.ROUTE_REPORT["graph_topology"] = what
                            warnings.warn(what, RuntimeWarning, stacklevel=3)
                        else:
                            state["g"] = None
                            raise RuntimeError(what + "; on more than four hardware queues (GPU_MAX_HW_QUEUES) forked graphs replayed "
                                               "3.5 - 6 ms slower per step and one replay crashed -- fork a side stream off the main "
                                               "chain once and join it once, or run on <= 4 queues")
                g.instantiate()
                self.segments.append(g)
            self.segment_nodes.append(n)
            state["g"] = None

        def on_break(name, call):
            end()
            with hip_ops.timed(name):
                call()                                   # the capture run executes the island once, eagerly
            self.segments.append((name, call))
            begin()

        self.empty_segments_dropped = 0
        self.segment_topology: List[dict] = []            # graph_topology.Topology.summary() per kept segment (check_topology)
        self.segment_nodes: List[Optional[int]] = []      # nodes per captured segment (dropped ones included), None = not readable
        # the hook is an object handed to hip_ops for the duration of THIS capture on THIS thread (thread-local scope)
        with _CAPTURE_LOCK, self._route(), hip_ops.batches_in_flight(self.in_flight), self.range_guard.armed(), \
                hip_ops.island_scope(hip_ops.IslandHook(eager_ops, on_break)), \
                hip_ops.workspace_scope(self.scratch), torch.cuda.stream(self.stream), torch.no_grad():
            begin()
            parts = model.forward_until_head(self.static_image, self.objects, None, None, self.object_group)
            end()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.range_guard.flag.zero_()                    # (the capture run's islands executed once: start clean)
        self.scratch.freeze()
        self.feat, self.queries, self.centers, self.bin_edges, self.detections = parts
        self.ReturnType = model.ReturnType
        self.islands = [s[0] for s in self.segments if isinstance(s, tuple)]

    @torch.no_grad()
    def load_objects(self, object_features, object_xywh_list=None, image: Optional[torch.Tensor] = None) -> None:
        """Copy one batch's objects into the graph's static buffers (current stream): a ``PaddedObjects`` of at most the graph's
        capacity, the reference's two lists (``object_xywh_list`` None = no boxes: every image carries the <UNK> box), or None = ask
        the model's provider (for ``image``, default the static image).  Device counts are not read back here: the kernels take
        them as min(max(count, 1), capacity) (csrc/objects_pad.hip), so a stray 0 cannot mask every key of an image."""
        if self.objects is None:
            raise RuntimeError("this graph was captured with its objects baked in (no object_capacity): it takes only the image")
        st = self.objects
        B, cap = st.features.shape[:2]
        if object_features is None:
            img = self.static_image if image is None else image
            padded = getattr(self.model.object_provider, "padded", None)
            if padded is not None:
                object_features = padded(img)
            else:
                object_features, object_xywh_list, _ = self.model.object_provider(img)
        if not isinstance(object_features, PaddedObjects):
            lens = [int(f.shape[0]) for f in object_features]
            if len(lens) != B or max(lens) > cap:
                raise ValueError(f"captured for {B} images of at most {cap} objects, got {len(lens)} lists, the longest of {max(lens)}")
            object_features = PaddedObjects.from_lists([f.float() for f in object_features], object_xywh_list, st.features.device)
        po = object_features
        n, k = po.features.shape[1], min(po.xywh.shape[2], 4)
        if po.features.shape[0] != B or n > cap or po.features.shape[2] != st.features.shape[2]:
            raise ValueError(f"captured for objects [{B}, <= {cap}, {st.features.shape[2]}], got {tuple(po.features.shape)}")
        st.features[:, :n].copy_(po.features)
        st.xywh[:, :n, :k].copy_(po.xywh[:, :, :k])
        st.counts.copy_(po.counts)

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, object_features=None, object_xywh_list=None):
        if image.shape != self.static_image.shape:
            raise ValueError(f"captured for {tuple(self.static_image.shape)}, got {tuple(image.shape)}")
        if self.objects is None and object_features is not None:
            raise RuntimeError("this graph was captured with its objects baked in (no object_capacity): it takes only the image")
        if image.data_ptr() != self.static_image.data_ptr():
            self.static_image.copy_(image)
        if self.objects is not None:
            self.load_objects(object_features, object_xywh_list, image)
        with self._route(), hip_ops.workspace_scope(self.scratch), self.range_guard.armed():
            for seg in self.segments:
                if isinstance(seg, tuple):
                    with hip_ops.timed(seg[0]):
                        seg[1]()
                else:
                    seg.replay()
            depth = self.model.head(self.feat, self.queries, self.centers)
        if self.pairs is None:
            self.last_flag = self.range_guard.take()     # one-thread launch: flag -> last_flag, flag = 0 (nobody waits for it here)
        return self.ReturnType(depth_pred=depth, bin_edges=self.bin_edges, detections=self.detections)

    def _route(self):
        return hip_ops.bf16_pairs() if self.pairs == "bf16" else contextlib.nullcontext()

    def tripped(self, taken: Optional[torch.Tensor] = None) -> bool:
        """Whether the last replay (or the replay whose ``last_flag`` the caller kept) converted a value beyond the fp16 pairs'
        guarded range.  A host read: call it where the result is read anyway."""
        taken = self.last_flag if taken is None else taken
        return taken is not None and hip_ops.RangeGuard.tripped(taken)

    @torch.no_grad()
    def rerun_on_bf16(self, image: torch.Tensor, object_features=None, object_xywh_list=None):
        """The same batch through the bf16-pair capture of the same forward (fp32's range, 2^-17 products instead of 2^-22: rounds
        1 - 3's arithmetic, parity-tested on its own) -- captured lazily, once, on the first trip; recorded in ROUTE_REPORT."""
        if self._fallback is None:
            cur = torch.cuda.current_stream(image.device)
            cur.synchronize()                            # nothing of this graph is in flight while its sibling is captured
            self._fallback = GraphedGraphBins(self.model, self.static_image, pairs="bf16", **self._ctor)
        self.trips += 1
        hip_ops.ROUTE_REPORT["range_guard"] = (f"{self.trips} batch(es) exceeded the fp16 pairs' guarded range (|x| > 65504 / 16 in a decoder "
                                               "/ heads activation) under a captured graph and were re-run on the bf16-pair capture")
        return self._fallback(image, object_features, object_xywh_list)

    @torch.no_grad()
    def checked(self, image: torch.Tensor, object_features=None, object_xywh_list=None):
        """``__call__`` + the range guard's verdict (a host synchronisation) + the handled fallback: what a caller that reads the
        result next should use (ValidationStep does); pipelined callers keep ``last_flag`` per step and decide when they collect."""
        out = self(image, object_features, object_xywh_list)
        if self.tripped():
            out = self.rerun_on_bf16(image, object_features, object_xywh_list)
        return out
