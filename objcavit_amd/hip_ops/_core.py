"""hip_ops core: activation codes, operand checks, per-entry-point timing, eager islands, forks / side streams, the workspace
store, and the fp16-pair control state (element type of the split pipeline, range guard, route report, first-batch range check).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import EncoderLayerParams, check


ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3, 4


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32, contiguous: bool = True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise _lib.HipLibraryError(f"{name} is on {t.device}: the ObjCAViT hot path runs only on a ROCm GPU "
                                   "(there is no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()



# ---------------------------------------------------------------------------
# optional per-entry-point timing with HIP events on the launch stream
# (bench.py turns this on to measure kernel durations inside the timed region)
# ---------------------------------------------------------------------------
class _Timing:
    enabled = False
    events: Dict[str, list] = {}


def enable_timing(on: bool = True) -> None:
    _Timing.enabled = on
    _Timing.events = {}


def pause_timing(paused: bool = True) -> None:
    """Stop (or resume) recording event pairs WITHOUT dropping the ones already recorded."""
    _Timing.enabled = not paused


def timing_results() -> Dict[str, Tuple[int, float]]:
    """name -> (launch count, mean milliseconds); synchronises the device."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in _Timing.events.items() if v}


class IslandHook:
    """Installed by objcavit_amd.graph around ONE capture (``with island_scope(hook)``): launches whose timing name is in
    ``names`` are kept OUT of the hipGraph (capture is ended in front of them and re-opened behind them) so that they
    run eagerly between two graph segments on every step and can be bracketed by HIP events.  The hook lives in
    thread-local state: two captures on two threads do not see each other's islands."""

    def __init__(self, names, on_break):
        self.names = tuple(names)
        self.on_break = on_break             # callable(name, closure)


class _Tls(threading.local):
    def __init__(self):
        self.island_hook = None
        self.islands_off = 0                 # > 0: inside islands_suspended()
        self.single_chain = 0                # > 0: inside single_chain() -- no further forks
        self.in_flight = 1                   # batches the caller keeps in flight on this GPU (batches_in_flight)
        self.fork_override = {}              # fork name -> forced on / off (forks())
        self.bf16_pairs = 0                  # > 0: inside bf16_pairs() -- the split pipeline on bf16 pairs
        self.range_flag = None               # the armed RangeGuard word of this thread (a tensor), or None
        self.ws_stack = None                 # workspace stores of this thread (bottom = the module-level store)


_TLS = _Tls()


class island_scope:
    def __init__(self, hook: IslandHook):
        self.hook = hook

    def __enter__(self):
        if _TLS.island_hook is not None:
            raise RuntimeError("island_scope: a capture with eager islands is already open on this thread")
        _TLS.island_hook = self.hook
        return self.hook

    def __exit__(self, *exc):
        _TLS.island_hook = None
        return False


class islands_suspended:
    """``with islands_suspended():`` launches issued inside stay IN the capture even when their name is an island's: for a launch
    that runs beside a forked stream (a capture cannot end while a fork is open)."""

    def __enter__(self):
        _TLS.islands_off += 1
        return self

    def __exit__(self, *exc):
        _TLS.islands_off -= 1
        return False


def launch(name: str, call) -> None:
    """Issue one C-ABI launch (``call`` enqueues it on the current stream) under the timing hook -- or hand it to the
    graph capturer of this thread as an eager island."""
    hook = _TLS.island_hook
    if hook is not None and name in hook.names and not _TLS.islands_off:
        hook.on_break(name, call)
        return
    with timed(name):
        call()


class timed:
    """Brackets a C-ABI call with a pair of events on the current stream when timing is enabled."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if _Timing.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _Timing.enabled:
            self.b.record()
            _Timing.events.setdefault(self.name, []).append((self.a, self.b))
        return False


# ---------------------------------------------------------------------------
# a second stream for work that is independent of the main chain (object tokens beside image tokens)
# ---------------------------------------------------------------------------
_SIDE: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def hw_queues_allow_forks() -> bool:
    """A captured forward may fork side streams only while the process runs on at most FOUR hardware queues (GPU_MAX_HW_QUEUES
    unset = the runtime's default of 4, or <= 4).  Round 5 (profiles/r05_graph_shapes.txt): the SAME captured graph with the
    product's single fork replays in 2.63 ms with 2 or 4 hardware queues and in 8.5 ms with 6, 8 or 16 -- and the same forward
    without a fork in 2.81 ms whatever the count.  A hipGraph's parallel branches are replayed on hardware queues of their own;
    beyond four, queues share a pipe of the command processor, which time-slices them: a branch parked on its cross-queue barrier
    packet holds the pipe until its quantum ends while the queue it waits for sits on the same pipe.  The 3.5 - 6 ms "shape"
    pathologies of round 4 (profiles/r04_skip_overlap.txt) were this: which shape lost depended on which queues its streams
    happened to be dealt, not on the shape (tools/graph_shapes.py: every rejected shape replays at full speed on 4 queues)."""
    raw = os.environ.get("GPU_MAX_HW_QUEUES")
    if raw is None:
        return True
    try:
        return int(raw) <= 4
    except ValueError:
        return True


def _side_switch(name: str) -> bool:
    """The four forks of a forward -- "obj" (object branch beside the encoder), "skip" (the decoder's skip-part convolutions beside
    the encoder's late stages, sharing that fork), "token" (object branch beside the image tokens), "head" (token chain beside the
    heads' convolution) -- under ONE switch, OCV_FORKS = 'auto' (default) | '0' | '1': auto = on for a lone batch on at most four
    hardware queues (``hw_queues_allow_forks``), off when the caller keeps several batches in flight on this GPU
    (``batches_in_flight``: bench.py's slots, PipelinedValidation).  A fork inside a captured forward makes the replay use further
    hardware queues; with three slots replaying at once those collide with the other slots' and the slots serialise each other.
    One box, alternating runs, bs 16 (profiles/r04_head_overlap.txt, block 5), three in flight / one at a time:  no fork 1037 / 953
    img/s;  obj 1015 / 961;  obj + head 976 / 972;  token alone 968 / 964.  Single forks can be forced on or off for tests and A/B
    tools with ``with hip_ops.forks(obj=..., token=..., head=..., skip=...)`` (thread-local), which wins over the environment."""
    forced = _TLS.fork_override.get(name)
    if forced is not None:
        return forced
    mode = os.environ.get("OCV_FORKS", "auto")
    if mode not in ("0", "1", "auto"):
        raise ValueError(f"OCV_FORKS={mode!r}: expected 'auto' (default), '1' or '0'")
    return (_TLS.in_flight <= 1 and hw_queues_allow_forks()) if mode == "auto" else mode == "1"


class forks:
    """``with forks(obj=False, head=True):`` forces single forks of the forward on or off on this thread (see ``_side_switch``)."""
    NAMES = ("obj", "token", "head", "skip")

    def __init__(self, **kw):
        bad = set(kw) - set(self.NAMES)
        if bad:
            raise ValueError(f"forks: unknown fork name(s) {sorted(bad)}; expected some of {self.NAMES}")
        self.kw = {k: bool(v) for k, v in kw.items() if v is not None}

    def __enter__(self):
        self.prev = dict(_TLS.fork_override)
        _TLS.fork_override.update(self.kw)
        return self

    def __exit__(self, *exc):
        _TLS.fork_override = self.prev
        return False


def token_overlap_enabled() -> bool:
    """Fork "token" (``_side_switch``): the object branch of the SA/CA stack (embedding, positional MLP, object self-attention:
    ~20 launches of a few workgroups each) on a side stream beside the image branch (patch embedding + image self-attention: equally
    latency-bound, small grids), joined in front of the cross-attention -- where the branch could not already be issued beside the
    encoder (``object_prepass_enabled``).  Lone batch: +3 % at bs 1 - 2, +0.9 % at bs 16."""
    return _side_switch("token") and not _TLS.single_chain


class single_chain:
    """``with single_chain():`` the code inside already runs beside another branch of the forward (the token chain beside the heads'
    convolution): it forks no further side stream (``token_overlap_enabled`` is False inside).  Two parallel branches are all a
    captured forward ever has -- a third one replays pathologically slowly or crashes hipStreamEndCapture on this ROCm
    (``head_overlap_enabled``)."""

    def __enter__(self):
        _TLS.single_chain += 1
        return self

    def __exit__(self, *exc):
        _TLS.single_chain -= 1
        return False


def object_prepass_enabled() -> bool:
    """Fork "obj" (``_side_switch``): where the object branch does not read the image features (the MLP positional strategies) it
    is issued at the top of the forward, on a side stream beside the encoder (GraphBins.forward_until_head), instead of behind the
    decoder.  On its own worth little (lone batch 953 -> 961 img/s); it leaves ONE side chain behind the decoder, which is what
    ``head_overlap_enabled`` needs."""
    return _side_switch("obj")


def skip_overlap_enabled() -> bool:
    """Fork "skip" (``_side_switch``): the skip-part convolutions of the decoder's last three stages (short-K GEMMs over encoder
    activations of stages 2 - 4, ~0.9 ms at bs 16) are issued on side stream 0 behind the encoder's fourth stage, beside its late
    stages, together with the object branch -- one fork, one join (modules/DenseFeatureExtractor.py ``SkipPrepass``).  Lone batch:
    +1.6 % at bs 16, +4.4 % at bs 1."""
    return _side_switch("skip")


class batches_in_flight:
    """``with batches_in_flight(n):`` forwards issued or CAPTURED inside belong to a caller that keeps ``n`` batches in flight on this
    GPU (bench.py's slots, PipelinedValidation; ``GraphedGraphBins(in_flight=n)`` wraps its own warm-up and capture in it).  Read by
    the side-stream switches (``_side_switch``): forks inside a forward pay for a lone batch only.  Thread-local, like a capture: two
    owners in one process never see each other's value (round 4 kept it in a module global that the last writer won)."""

    def __init__(self, n: int):
        self.n = max(1, int(n))

    def __enter__(self):
        self.prev = _TLS.in_flight
        _TLS.in_flight = self.n
        return self

    def __exit__(self, *exc):
        _TLS.in_flight = self.prev
        return False


def head_overlap_enabled() -> bool:
    """Fork "head" (``_side_switch``): the heads' 3x3 convolution over the decoder's map (4800 workgroups, ~0.93 ms at bs 16) is
    issued on the main stream while the image-token chain -- patch embedding, self-attention stack, cross-attention, bin regressor:
    ~25 launches of 2 - 300 workgroups, ~0.6 ms of mostly idle chip -- runs on a second side stream; joined in front of the bin head,
    the first consumer of both.  Lone batch at bs 16: 961 -> 972 img/s.
    Only with ONE side chain behind the decoder (the object branch already issued beside the encoder, or a model without one): a
    captured forward with the object chain, the token chain and the convolution as three parallel branches replays 6 ms SLOWER
    per step on this ROCm (23.2 vs 16.6 ms at bs 16, 8.7 vs 3.5 ms at bs 1), and a nested third branch crashed
    hipStreamEndCapture, so that shape is never built.  The token kernels hold 52 KB of LDS per workgroup and cannot share a CU
    with the convolution's 144 KB: beside it they run ~2x slower and the convolution 1.18 instead of 0.93 ms -- which is why the
    gain is a third of the chain's length."""
    return _side_switch("head")


def side_stream(device: torch.device, which: int = 0) -> "torch.cuda.Stream":
    """The process's side streams of ``device`` (created on first use; scratch is keyed by stream like everyone's): 0 = the object
    branch, 1 = the token chain beside the heads' convolution."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get((idx, which))
    if st is None:
        st = _SIDE[(idx, which)] = torch.cuda.Stream(device=idx)
    return st


_PROBE: Dict[int, "torch.Tensor"] = {}


def streams_share_a_queue(a: "torch.cuda.Stream", b: "torch.cuda.Stream", sleep_cycles: int = 400_000) -> bool:
    """Whether work on stream ``b`` waits for work on stream ``a`` although nothing orders them: a long, empty kernel on ``a``, then a
    tiny one on ``b``; if the tiny one ends only with the long one the two streams sit on ONE hardware queue.  ROCm deals its
    GPU_MAX_HW_QUEUES hardware queues to streams by a rule of its own, at a stream's first use (tools/exp_stream_queues.py) -- the only
    way to know is to look.  The tiny kernel writes a scratch word allocated beforehand (an allocation inside the probe would itself
    wait for the device).  ~1 ms; synchronises the device."""
    dev = a.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    scratch = _PROBE.get(idx)
    if scratch is None:
        scratch = _PROBE[idx] = torch.zeros(64, device=dev)
    for st in (a, b):                                     # first use of a stream = its hardware queue is dealt: not inside the timed part
        with torch.cuda.stream(st):
            scratch.zero_()
    torch.cuda.synchronize(dev)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record()
        torch.cuda._sleep(sleep_cycles)
        e1.record()
    with torch.cuda.stream(b):
        scratch.zero_()
        e2.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e2) > 0.5 * e0.elapsed_time(e1)


def independent_streams(n: int, device: torch.device, candidates: int = 16) -> list:
    """``n`` new streams of ``device`` no two of which share a hardware queue (checked pairwise, both directions:
    ``streams_share_a_queue``), for callers that keep several captured forwards in flight -- two slots on one queue run one after the
    other (round 4: 781 instead of 840 img/s at bs 16; round 6: 500 instead of 739 at bs 1, profiles/r06_stream_queues.txt).  Up to
    ``candidates`` streams are created; rejected ones are dropped.  If the runtime has fewer than ``n`` independent queues to give
    (GPU_MAX_HW_QUEUES < n) the best set found is returned, completed with colliding streams, and the shortfall is recorded in
    ``ROUTE_REPORT["independent_streams"]``.  Not capturable; call it before any capture.  ~1 ms per pair tested."""
    if device.type != "cuda":
        raise _lib.HipLibraryError("independent_streams needs a GPU device")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if not hasattr(torch.cuda, "_sleep"):                  # (the probe's long kernel: a private torch helper -- without it, streams as dealt)
        ROUTE_REPORT["independent_streams"] = "torch.cuda._sleep is missing: slot streams were not checked for shared hardware queues"
        return [torch.cuda.Stream(device=idx) for _ in range(n)]
    chosen, spare = [], []
    for _ in range(max(n, candidates)):
        if len(chosen) == n:
            break
        st = torch.cuda.Stream(device=idx)
        if all(not streams_share_a_queue(c, st) and not streams_share_a_queue(st, c) for c in chosen):
            chosen.append(st)
        else:
            spare.append(st)
    if len(chosen) < n:
        ROUTE_REPORT["independent_streams"] = (f"only {len(chosen)} of {n} requested streams got a hardware queue of their own "
                                               f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}): the others share one and serialise")
        chosen += spare[:n - len(chosen)]
        while len(chosen) < n:
            chosen.append(torch.cuda.Stream(device=idx))
    return chosen


# ---------------------------------------------------------------------------
# workspace: one growing byte buffer per (device, stream, tag), held in a STORE
# ---------------------------------------------------------------------------
class WorkspaceStore(dict):
    """(device index, stream handle, tag) -> uint8 buffer.  The module-level store serves eager calls.  A captured
    hipGraph bakes the buffers' addresses into its nodes, so a graph owns a store of its own
    (``with workspace_scope(store)`` around its warm-up, capture and replays): nothing outside can grow -- i.e. free --
    scratch that the graph still writes on every replay, and ``freeze()`` turns a later growth request inside the
    scope into an error instead of a silent re-allocation.  Buffers are keyed by stream as well as tag: two streams of
    one forward (its forks: ``_side_switch``) never share scratch."""

    def __init__(self):
        super().__init__()
        self.frozen = False

    def freeze(self):
        self.frozen = True


_WS = WorkspaceStore()


def _ws_stack() -> list:
    if _TLS.ws_stack is None:
        _TLS.ws_stack = [_WS]
    return _TLS.ws_stack


class workspace_scope:
    """``with workspace_scope(store)``: workspace requests of THIS thread come from ``store`` (a graph's own scratch)."""

    def __init__(self, store: WorkspaceStore):
        self.store = store

    def __enter__(self):
        _ws_stack().append(self.store)
        return self.store

    def __exit__(self, *exc):
        _ws_stack().pop()
        return False


def workspace(nbytes: int, device: torch.device, tag: str = "default", zero: bool = False) -> torch.Tensor:
    """``zero``: the buffer is zero-filled when it is (re)allocated -- for words a kernel finds zero and leaves zero (the
    arrival counters of the in-launch squeeze-excite tail), which must never come out of recycled, dirty memory."""
    store = _ws_stack()[-1]
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, tag)
    buf = store.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("workspace would grow during graph capture: run one eager warm-up call first")
        if store.frozen and buf is not None:
            raise RuntimeError(f"workspace {key} of a captured graph would have to grow from {buf.numel()} to {nbytes} "
                               "bytes: the graph's nodes hold the old address -- capture a new graph for the new shapes")
        buf = (torch.zeros if zero else torch.empty)(max(nbytes, 1), dtype=torch.uint8, device=device)
        store[key] = buf
    return buf



def conv_split_f16() -> bool:
    """Element type of the two-term split of the decoder's / heads' convolutions (the "hl32" activations between them and their
    weights): OCV_CONV_SPLIT = 'f16' (default, round 4: fp16 pairs, products good to 2^-22 on v_mfma_*_f16; weights scaled per
    output channel out of fp16's subnormals, activations range-checked -- ``range_check`` / ``fp16_range_report``) or 'bf16'
    (rounds 1 - 3: bf16 pairs, 2^-17, fp32's range: the A/B route, and what a model falls back to -- reported in
    ``ROUTE_REPORT`` -- when its weights do not fit fp16 pairs)."""
    mode = os.environ.get("OCV_CONV_SPLIT", "f16")
    if mode not in ("f16", "bf16"):
        raise ValueError(f"OCV_CONV_SPLIT={mode!r}: expected 'f16' (default) or 'bf16'")
    return mode == "f16" and not _TLS.bf16_pairs


class bf16_pairs:
    """``with bf16_pairs():`` forwards issued (or captured) inside run on the forms with FP32'S RANGE (thread-local): the decoder's /
    heads' split pipeline on bf16 pairs whatever OCV_CONV_SPLIT says, the token stacks' two-term fp16 layers as three-term bf16
    (``token_mode``), the few-key cross-attention likewise, the attention cores on exact fp32 (``ocv_attention_set_fp32_range``),
    the bin head as three-term bf16 (``bin_head``) -- rounds 1 - 3's arithmetic, parity-tested on its own.  It is the handled
    fallback of a batch that tripped the fp16 range guard (``RangeGuard``), and how that fallback's hipGraph is captured."""

    def __enter__(self):
        _TLS.bf16_pairs += 1
        if _TLS.bf16_pairs == 1 and torch.cuda.is_available():
            check(_lib.load().ocv_attention_set_fp32_range(1), "ocv_attention_set_fp32_range")
        return self

    def __exit__(self, *exc):
        _TLS.bf16_pairs -= 1
        if _TLS.bf16_pairs == 0 and torch.cuda.is_available():
            check(_lib.load().ocv_attention_set_fp32_range(0), "ocv_attention_set_fp32_range")
        return False


class RangeGuard:
    """One device word that every launch writing fp16 pairs ORs 1 into when a value it converts exceeds 65504 / 16 = 4094 in magnitude (the first-batch calibration's own limit)
    (include/objcavit_hip.h ``ocv_range_flag_set``; csrc/common.hpp ``ocv_range_note``).  The reference computes these layers in
    fp32 for any input (modules/DenseFeatureExtractor.py:37-47,104-118); the fp16-pair pipeline is calibrated on a model's FIRST
    batch only, and a captured graph cannot change its mind -- so the owner of a forward (GraphBins / AdaBins eagerly,
    GraphedGraphBins per replay) arms this word around its launches, takes it behind them (``take``: a one-thread launch on the
    stream, capturable) and reads the taken copy where it reads results (``tripped``: a host read); a tripped batch is re-run on
    bf16 pairs and recorded in ``ROUTE_REPORT``.  Arming is per thread, like a capture."""

    def __init__(self, device: torch.device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)

    def armed(self) -> "_Armed":
        return _Armed(self.flag)

    def take(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Enqueue ``out = flag; flag = 0`` on the current stream; -> ``out`` (int32 [1] on the device, a new tensor by default)."""
        if out is None:
            out = torch.empty(1, dtype=torch.int32, device=self.flag.device)
        check(_lib.load().ocv_range_flag_take_fwd(self.flag.data_ptr(), out.data_ptr(), _stream()), "ocv_range_flag_take_fwd")
        return out

    @staticmethod
    def tripped(taken: torch.Tensor) -> bool:
        """Host read of a taken word (synchronises with the stream that took it)."""
        return bool(int(taken.item()) != 0)


def guarded_forward(owner, decoder, device: torch.device, run):
    """The eager inference forward of a model under its range guard: ``run()`` issues the forward; when the decoder runs on fp16
    pairs and a launch tripped the guard, the SAME batch is issued again on bf16 pairs (``bf16_pairs``) and the route is reported.
    Costs one one-thread launch and one 4-byte host read per forward (a synchronisation: the eager path is the slow path anyway --
    a captured forward keeps the word on the device, objcavit_amd/graph.py); ``owner.range_guard_sync = False`` switches it off.
    Inside a graph capture (warm-up of another owner's graph excepted: that is eager) nothing is read: the capturing owner arms
    its own word."""
    if (device.type != "cuda" or torch.is_grad_enabled() or owner.training or torch.cuda.is_current_stream_capturing()
            or not getattr(owner, "range_guard_sync", True) or _TLS.bf16_pairs or decoder.settled_f16() is False):
        return run()
    guard = owner.__dict__.get("_range_guard")
    if guard is None or guard.flag.device != device:
        guard = owner.__dict__["_range_guard"] = RangeGuard(device)
    with guard.armed():
        out = run()
    if RangeGuard.tripped(guard.take()):
        n = owner.__dict__["_range_trips"] = owner.__dict__.get("_range_trips", 0) + 1
        ROUTE_REPORT["range_guard"] = (f"{n} batch(es) exceeded the fp16 pairs' guarded range (|x| > 65504 / 16 in a decoder / heads "
                                       "activation) and were re-run on bf16 pairs")
        with bf16_pairs():
            out = run()
    return out


class _Armed:
    def __init__(self, flag: torch.Tensor):
        self.flag = flag

    def __enter__(self):
        self.prev = _TLS.range_flag
        _TLS.range_flag = self.flag
        check(_lib.load().ocv_range_flag_set(self.flag.data_ptr()), "ocv_range_flag_set")
        return self

    def __exit__(self, *exc):
        _TLS.range_flag = self.prev
        check(_lib.load().ocv_range_flag_set(None if self.prev is None else self.prev.data_ptr()), "ocv_range_flag_set")
        return False


ROUTE_REPORT: Dict[str, str] = {}        # layer / model name -> why it left the default route (never silent: bench.py prints it)



# ---------------------------------------------------------------------------
# fp16 range check of the split activations (diagnostic: host synchronisation per tensor)
# ---------------------------------------------------------------------------
class _Range:
    enabled = False
    seen: Dict[str, Tuple[float, float]] = {}


def range_check(on: bool = True) -> None:
    """Start (and reset) / stop recording the largest and the smallest-nonzero-block magnitude of every fp16 hl32 tensor the
    path produces.  Diagnostic: every record is a host synchronisation -- run ONE eager forward under it (bench.py does, before
    its timed region), never a captured or timed one."""
    _Range.enabled = on
    if on:
        _Range.seen = {}


def _note_range(name: str, ys: Optional["SplitAct"]) -> None:
    if not _Range.enabled or ys is None or not ys.f16 or torch.cuda.is_current_stream_capturing():
        return
    hi = ys.hi.float().abs()
    amax = float(hi.amax()) if hi.numel() else 0.0
    finite = bool(torch.isfinite(hi).all())
    old = _Range.seen.get(name)
    _Range.seen[name] = (max(amax, old[0]) if old else amax, (old[1] if old else True) and finite)


def fp16_range_report() -> dict:
    """What ``range_check`` saw: per fp16 hl32 tensor its largest magnitude; ``ok`` = every tensor finite, its largest entry below
    fp16's 65504 with a factor 16 to spare and above 2^-6 (a tensor whose LARGEST entry is below that has every low term in fp16's
    subnormals: its pairs are then good to ~2^-17 instead of 2^-22, still the bf16 pairs' precision)."""
    t = {k: v[0] for k, v in _Range.seen.items()}
    bad = {k: v[0] for k, v in _Range.seen.items() if not v[1] or v[0] > 65504.0 / 16 or (0.0 < v[0] < 2.0 ** -6)}
    return {"tensors": len(t), "max_amax": max(t.values()) if t else None, "min_amax": min(t.values()) if t else None,
            "ok": not bad, "out_of_range": bad}


__all__ = [_n for _n in dir() if not _n.startswith("__")]        # (private helpers included: the facade re-exports every name)
