"""Tensor-level wrappers over the C ABI (include/objcavit_hip.h).

PyTorch is used here for device memory, the current HIP stream and nothing
else: every function validates its operands on the host (shape, dtype, device,
contiguity -- a wrong shape must never reach a hand-written kernel), takes raw
``data_ptr()`` values and enqueues our kernels on ``torch.cuda.current_stream()``.
All of them raise if the tensors are not on a GPU or the library is missing.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import EncoderLayerParams, check

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


# ---------------------------------------------------------------------------
# linear / layernorm
# ---------------------------------------------------------------------------
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x @ weight.T + bias); x [..., K] contiguous, weight [N, K]."""
    lib = _lib.load()
    _req(x, "x"); _req(weight, "weight")
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.dim() != 2 or weight.shape[1] != K:
        raise ValueError(f"linear: weight {tuple(weight.shape)} does not match x[..., {K}]")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != N:
            raise ValueError("linear: bias size mismatch")
    M = x.numel() // K
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    else:
        _req(out, "out")
        if out.numel() != M * N:
            raise ValueError("linear: out size mismatch")
    check(lib.ocv_linear_fwd(x.data_ptr(), K, 0, weight.data_ptr(), K, 0, 0, _ptr(bias), out.data_ptr(), N, 0, 1, M, N, K,
                             act, _stream()), "ocv_linear_fwd")
    return out


def linear_residual_layernorm(a: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor,
                              gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                              zero_row_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm(residual + a @ weight.T + bias) over the last dim (== 128)."""
    lib = _lib.load()
    for n, t in (("a", a), ("weight", weight), ("bias", bias), ("residual", residual), ("gamma", gamma), ("beta", beta)):
        _req(t, n)
    K, N = a.shape[-1], weight.shape[0]
    M = a.numel() // K
    if weight.shape != (N, K) or residual.shape[-1] != N or residual.numel() != M * N or gamma.numel() != N or beta.numel() != N:
        raise ValueError("linear_residual_layernorm: shape mismatch")
    if zero_row_mask is not None:
        _req(zero_row_mask, "zero_row_mask", torch.uint8)
        if zero_row_mask.numel() != M:
            raise ValueError("zero_row_mask: one byte per row expected")
    out = torch.empty_like(residual)
    check(lib.ocv_linear_residual_layernorm_fwd(a.data_ptr(), K, weight.data_ptr(), K, bias.data_ptr(), residual.data_ptr(),
                                                N, gamma.data_ptr(), beta.data_ptr(), eps, _ptr(zero_row_mask),
                                                out.data_ptr(), N, M, N, K, _stream()),
          "ocv_linear_residual_layernorm_fwd")
    return out


def ffn_residual_layernorm(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
                           gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                           zero_row_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm(x + W2 relu(W1 x + b1) + b2), fused (hidden activations stay on chip); x [..., 128]."""
    lib = _lib.load()
    for n, t in (("x", x), ("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("gamma", gamma), ("beta", beta)):
        _req(t, n)
    E = x.shape[-1]
    FF = w1.shape[0]
    M = x.numel() // E
    if w1.shape != (FF, E) or w2.shape != (E, FF) or b1.numel() != FF or b2.numel() != E or gamma.numel() != E or beta.numel() != E:
        raise ValueError("ffn_residual_layernorm: shape mismatch")
    if zero_row_mask is not None:
        _req(zero_row_mask, "zero_row_mask", torch.uint8)
        if zero_row_mask.numel() != M:
            raise ValueError("zero_row_mask: one byte per row expected")
    out = torch.empty_like(x)
    check(lib.ocv_ffn_residual_layernorm_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                             gamma.data_ptr(), beta.data_ptr(), eps, _ptr(zero_row_mask), out.data_ptr(), M, E,
                                             FF, _stream()), "ocv_ffn_residual_layernorm_fwd")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _req(x, "x"); _req(gamma, "gamma"); _req(beta, "beta")
    E = x.shape[-1]
    if gamma.numel() != E or beta.numel() != E:
        raise ValueError("layernorm: parameter size mismatch")
    if residual is not None:
        _req(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("layernorm: residual shape mismatch")
    out = torch.empty_like(x)
    check(lib.ocv_layernorm_residual_fwd(x.data_ptr(), _ptr(residual), gamma.data_ptr(), beta.data_ptr(), eps,
                                         out.data_ptr(), x.numel() // E, E, _stream()), "ocv_layernorm_residual_fwd")
    return out


# ---------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------
def _mask_u8(mask: Optional[torch.Tensor], B: int, Sk: int) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if mask.device.type != "cuda":
        raise _lib.HipLibraryError("key_padding_mask must be on the GPU")
    if mask.shape != (B, Sk):
        raise ValueError(f"key_padding_mask: expected {(B, Sk)}, got {tuple(mask.shape)}")
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8) if mask.is_contiguous() else mask.contiguous().view(torch.uint8)
    elif mask.dtype != torch.uint8:
        raise TypeError("key_padding_mask must be bool or uint8")
    return mask.contiguous()


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_padding_mask: Optional[torch.Tensor],
                   n_heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(32) + mask) v per head; q [B,Sq,E], k / v [B,Sk,E] (last-dim-contiguous views allowed)."""
    lib = _lib.load()
    for n, t in (("q", q), ("k", k), ("v", v)):
        _req(t, n, contiguous=False)
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError(f"{n}: expected [B, S, E] with unit stride on E")
    B, Sq, E = q.shape
    Sk = k.shape[1]
    if k.shape != (B, Sk, E) or v.shape != (B, Sk, E) or E != n_heads * 32:
        raise ValueError("attention_core: shape mismatch (head dim must be 32)")
    m = _mask_u8(key_padding_mask, B, Sk)
    ctx = torch.empty(B, Sq, E, dtype=torch.float32, device=q.device)
    check(lib.ocv_attention_fwd(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                v.data_ptr(), v.stride(0), v.stride(1), _ptr(m), ctx.data_ptr(), Sq * E, E, B, n_heads,
                                Sq, Sk, 1.0 / math.sqrt(32.0), _stream()), "ocv_attention_fwd")
    return ctx


def mha(q_src: torch.Tensor, k_src: torch.Tensor, v_src: torch.Tensor, in_proj_w: torch.Tensor, in_proj_b: torch.Tensor,
        out_w: torch.Tensor, out_b: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None,
        n_heads: int = 4, kv_limit: int = 0, packed: Optional[dict] = None) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True, need_weights=False) forward.
    kv_limit > 0: the caller guarantees every key j >= kv_limit is masked in every row, so those keys are skipped.
    ``packed``: the caller's cache of SplitWeight3 objects for this module (filled / refreshed here, keyed on the weights'
    identity and version) -> with at most 32 live keys every contraction runs as a two-term fp16 split with K / V projected once
    per image (ocv_mha_few_keys_h2_fwd; OCV_TOKENS=split3 | fp32 select the older forms), otherwise the projections run as three-term bf16
    splits (ocv_mha_split3_fwd); None, or OCV_TOKENS=fp32 -> the exact-fp32 kernels (ocv_mha_fwd)."""
    lib = _lib.load()
    for n, t in (("q_src", q_src), ("k_src", k_src), ("v_src", v_src), ("in_proj_weight", in_proj_w),
                 ("in_proj_bias", in_proj_b), ("out_proj.weight", out_w), ("out_proj.bias", out_b)):
        _req(t, n)
    B, Sq, E = q_src.shape
    Sk = k_src.shape[1]
    if k_src.shape != (B, Sk, E) or v_src.shape != (B, Sk, E):
        raise ValueError("mha: key / value shape mismatch")
    if in_proj_w.shape != (3 * E, E) or in_proj_b.numel() != 3 * E or out_w.shape != (E, E) or out_b.numel() != E:
        raise ValueError("mha: parameter shape mismatch")
    m = _mask_u8(key_padding_mask, B, Sk)
    nb = lib.ocv_mha_workspace_bytes(B, Sq, Sk, E)
    ws = workspace(nb, q_src.device)
    out = torch.empty(B, Sq, E, dtype=torch.float32, device=q_src.device)
    name = "mha_self" if q_src.data_ptr() == k_src.data_ptr() else ("mha_cross" if kv_limit else "mha_cross_full")
    # few live keys (the image <- object cross-attention), by OCV_TOKENS (``token_mode``): h2 (default: every contraction a two-term
    # fp16 split, csrc/xattn_h2.hip) | split3 (three-term bf16 projections + exact-fp32 scores, fp32's range) | fp32 (round 2's
    # single exact-fp32 launch); profiles/r03_cross_attention_roofline.txt has the three side by side at bs 16 ... 2048
    form = token_mode()
    few = 0 < (kv_limit if 0 < kv_limit < Sk else Sk) <= 32 and (kv_limit == 0 or m is not None)
    small = few and form == "fp32"
    if packed is not None and token_split3_enabled() and E == 128 and n_heads == 4 and few and form == "h2" and B <= 65535:
        h2 = []
        for field, w in (("in_proj_h2", in_proj_w), ("out_proj_h2", out_w)):
            ver = (w.data_ptr(), w._version)
            hit = packed.get(field)
            if hit is None or hit[0] != ver:
                hit = packed[field] = (ver, SplitWeightH2(w))
            h2.append(hit[1].packed)
        with timed(name):                              # the K / V record (32 KB per image) fits the MHA workspace sized above
            check(lib.ocv_mha_few_keys_h2_fwd(q_src.data_ptr(), k_src.data_ptr(), v_src.data_ptr(), _ptr(m), h2[0].data_ptr(),
                                              in_proj_b.data_ptr(), h2[1].data_ptr(), out_b.data_ptr(), out.data_ptr(), B, Sq, Sk,
                                              int(kv_limit), E, n_heads, ws.data_ptr(), ws.numel(), _stream()), "ocv_mha_few_keys_h2_fwd")
        return out
    if packed is not None and token_split3_enabled() and E == 128 and n_heads == 4 and not small:
        p3 = []
        for field, w in (("in_proj_p3", in_proj_w), ("out_proj_p3", out_w)):
            ver = (w.data_ptr(), w._version)
            hit = packed.get(field)
            if hit is None or hit[0] != ver:
                hit = packed[field] = (ver, SplitWeight3(w))
            p3.append(hit[1].packed)
        with timed(name):
            check(lib.ocv_mha_split3_fwd(q_src.data_ptr(), k_src.data_ptr(), v_src.data_ptr(), _ptr(m), p3[0].data_ptr(),
                                         in_proj_b.data_ptr(), p3[1].data_ptr(), out_b.data_ptr(), out.data_ptr(), B, Sq, Sk,
                                         int(kv_limit), E, n_heads, ws.data_ptr(), ws.numel(), _stream()), "ocv_mha_split3_fwd")
        return out
    with timed(name):
        check(lib.ocv_mha_fwd(q_src.data_ptr(), k_src.data_ptr(), v_src.data_ptr(), _ptr(m), in_proj_w.data_ptr(),
                              in_proj_b.data_ptr(), out_w.data_ptr(), out_b.data_ptr(), out.data_ptr(), B, Sq, Sk, int(kv_limit), E, n_heads,
                              ws.data_ptr(), ws.numel(), _stream()), "ocv_mha_fwd")
    return out


_LAYER_FIELDS = (("in_proj_w", "self_attn.in_proj_weight"), ("in_proj_b", "self_attn.in_proj_bias"),
                 ("out_proj_w", "self_attn.out_proj.weight"), ("out_proj_b", "self_attn.out_proj.bias"),
                 ("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
                 ("linear1_w", "linear1.weight"), ("linear1_b", "linear1.bias"),
                 ("linear2_w", "linear2.weight"), ("linear2_b", "linear2.bias"),
                 ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"))


class SplitWeightH2:
    """A static [N, K] matrix as two fp16 terms, w = hi + 2^-11 lo' (22 significant bits), packed in matrix-core operand order
    by the device (ocv_pack_split_h2_fwd; layout in include/objcavit_hip.h).  Built once per weight version by the callers."""

    def __init__(self, weight: torch.Tensor):
        lib = _lib.load()
        w = _req(weight.detach().reshape(weight.shape[0], -1).contiguous(), "weight")
        self.n, self.k = int(w.shape[0]), int(w.shape[1])
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("weight packing during graph capture: run one eager warm-up call first")
        self.packed = torch.empty(int(lib.ocv_split_h2_packed_elems(self.n, self.k)), dtype=torch.float16, device=w.device)
        check(lib.ocv_pack_split_h2_fwd(w.data_ptr(), self.k, self.n, self.k, self.packed.data_ptr(), _stream()), "ocv_pack_split_h2_fwd")


class SplitWeight3:
    """A static [N, K] matrix split into three bf16 terms (24 significant bits) and packed in matrix-core B-operand
    order by the device (ocv_pack_split3_fwd; layout in include/objcavit_hip.h).  Built once per weight version by the
    callers (cached next to the parameter)."""

    def __init__(self, weight: torch.Tensor):
        lib = _lib.load()
        w = _req(weight.detach().reshape(weight.shape[0], -1).contiguous(), "weight")
        self.n, self.k = int(w.shape[0]), int(w.shape[1])
        if self.k % 8 != 0:
            raise ValueError("SplitWeight3: K must be a multiple of 8")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("weight packing during graph capture: run one eager warm-up call first")
        self.packed = torch.empty(int(lib.ocv_split3_packed_elems(self.n, self.k)), dtype=torch.bfloat16, device=w.device)
        check(lib.ocv_pack_split3_fwd(w.data_ptr(), self.k, self.n, self.k, self.packed.data_ptr(), _stream()), "ocv_pack_split3_fwd")


def linear_split3(x: torch.Tensor, weight: SplitWeight3, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """act(x @ W.T + bias) with three-term-split operands (fp32-faithful); x [..., K] contiguous."""
    lib = _lib.load()
    _req(x, "x")
    K = x.shape[-1]
    if K != weight.k:
        raise ValueError(f"linear_split3: weight with K={weight.k} does not match x[..., {K}]")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != weight.n:
            raise ValueError("linear_split3: bias size mismatch")
    M = x.numel() // K
    out = torch.empty(*x.shape[:-1], weight.n, dtype=torch.float32, device=x.device)
    check(lib.ocv_linear_split3_fwd(x.data_ptr(), K, weight.packed.data_ptr(), _ptr(bias), out.data_ptr(), weight.n, M, weight.n, K,
                                    act, _stream()), "ocv_linear_split3_fwd")
    return out


def ffn_residual_layernorm_split3(x: torch.Tensor, w1: SplitWeight3, b1: torch.Tensor, w2: SplitWeight3, b2: torch.Tensor,
                                  gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                                  zero_row_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm(x + W2 relu(W1 x + b1) + b2) with three-term-split operands; x [..., 128]."""
    lib = _lib.load()
    for n, t in (("x", x), ("b1", b1), ("b2", b2), ("gamma", gamma), ("beta", beta)):
        _req(t, n)
    E = x.shape[-1]
    FF = w1.n
    M = x.numel() // E
    if w1.k != E or w2.n != E or w2.k != FF or b1.numel() != FF or b2.numel() != E:
        raise ValueError("ffn_residual_layernorm_split3: shape mismatch")
    if zero_row_mask is not None:
        _req(zero_row_mask, "zero_row_mask", torch.uint8)
    nb = int(lib.ocv_ffn_split3_workspace_bytes(M, FF))
    ws = workspace(nb, x.device, "ffn3") if nb else None
    out = torch.empty_like(x)
    check(lib.ocv_ffn_residual_layernorm_split3_fwd(x.data_ptr(), w1.packed.data_ptr(), b1.data_ptr(), w2.packed.data_ptr(), b2.data_ptr(),
                                                    gamma.data_ptr(), beta.data_ptr(), eps, _ptr(zero_row_mask), out.data_ptr(), M, E, FF,
                                                    _ptr(ws), nb, _stream()), "ocv_ffn_residual_layernorm_split3_fwd")
    return out


_P3_FIELDS = (("in_proj_p3", "self_attn.in_proj_weight"), ("out_proj_p3", "self_attn.out_proj.weight"),
              ("linear1_p3", "linear1.weight"), ("linear2_p3", "linear2.weight"))


def token_mode() -> str:
    """OCV_TOKENS: 'h2' (default: the layer tails -- output projection, LayerNorms, feed-forward block, next projection -- as
    two-term fp16 splits, csrc/token_h2.hip; the remaining token linears as three-term bf16 splits), 'split3' (three-term bf16
    everywhere: round 2's route, fp32's range) or 'fp32' (the exact-fp32 MFMA kernels: the A/B numerics route)."""
    mode = os.environ.get("OCV_TOKENS", "h2")
    if mode not in ("h2", "split3", "fp32"):
        raise ValueError(f"OCV_TOKENS={mode!r}: expected 'h2' (default), 'split3' or 'fp32'")
    return "split3" if (mode == "h2" and _TLS.bf16_pairs) else mode          # (inside bf16_pairs(): the forms with fp32's range)


def token_split3_enabled() -> bool:
    """Whether the transformer layers' projections / feed-forward blocks run on packed split weights (OCV_TOKENS = h2 or split3)
    rather than on the exact-fp32 MFMA kernels (OCV_TOKENS=fp32)."""
    return token_mode() != "fp32"


def layer_params(layer: torch.nn.Module, packed: Optional[dict] = None) -> Tuple[EncoderLayerParams, list]:
    """Pointer table for one nn.TransformerEncoderLayer-shaped parameter holder.  ``packed``: the caller's cache of
    SplitWeight3 objects for this layer (filled / refreshed here, keyed on the parameters' identity and version);
    None = exact-fp32 kernels.  Returns (struct, keep-alive list of tensors)."""
    sd = dict(layer.named_parameters())
    st = EncoderLayerParams()
    keep = []
    for field, key in _LAYER_FIELDS:
        t = _req(sd[key].detach(), key)
        keep.append(t)
        setattr(st, field, t.data_ptr())
    if packed is not None:
        for field, key in _P3_FIELDS:
            w = sd[key]
            ver = (w.data_ptr(), w._version)
            hit = packed.get(field)
            if hit is None or hit[0] != ver:
                hit = packed[field] = (ver, SplitWeight3(w))
            keep.append(hit[1].packed)
            setattr(st, field, hit[1].packed.data_ptr())
        if token_mode() == "h2":                     # + the two-term fp16 copies: the layer tails run on them
            for field, key in _P3_FIELDS:
                f2 = field.replace("_p3", "_h2")
                w = sd[key]
                ver = (w.data_ptr(), w._version)
                hit = packed.get(f2)
                if hit is None or hit[0] != ver:
                    hit = packed[f2] = (ver, SplitWeightH2(w))
                keep.append(hit[1].packed)
                setattr(st, f2, hit[1].packed.data_ptr())
    return st, keep


def encoder_layer(x: torch.Tensor, params: EncoderLayerParams, key_padding_mask: Optional[torch.Tensor] = None,
                  zero_padded_rows: bool = False, n_heads: int = 4, dim_ff: int = 1024, eps: float = 1e-5,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 3:
        raise ValueError("encoder_layer: x must be [B, S, E]")
    B, S, E = x.shape
    m = _mask_u8(key_padding_mask, B, S)
    nb = lib.ocv_encoder_layer_workspace_bytes(B, S, E, dim_ff)
    ws = workspace(nb, x.device)
    if out is None:
        out = torch.empty_like(x)
    with timed("encoder_layer" if S > 64 else "encoder_layer_obj"):
      check(lib.ocv_encoder_layer_fwd(x.data_ptr(), C.byref(params), _ptr(m), int(zero_padded_rows), out.data_ptr(), B, S, E,
                                      n_heads, dim_ff, eps, ws.data_ptr(), ws.numel(), _stream()), "ocv_encoder_layer_fwd")
    return out


def encoder_stack(x: torch.Tensor, params: Sequence[EncoderLayerParams], key_padding_mask: Optional[torch.Tensor] = None,
                  zero_padded_rows: bool = False, n_heads: int = 4, dim_ff: int = 1024, eps: float = 1e-5) -> torch.Tensor:
    """A whole nn.TransformerEncoder in 1 + 2 L launches (ocv_encoder_stack_fwd); every layer's params must carry the
    packed split3 weights (layer_params(layer, packed_cache))."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 3:
        raise ValueError("encoder_stack: x must be [B, S, E]")
    B, S, E = x.shape
    m = _mask_u8(key_padding_mask, B, S)
    arr = (EncoderLayerParams * len(params))(*params)
    nb = lib.ocv_encoder_stack_workspace_bytes(B, S, E)
    ws = workspace(nb, x.device, "encoder_stack")
    out = torch.empty_like(x)
    with timed("encoder_stack" if S > 64 else "encoder_stack_obj"):
        check(lib.ocv_encoder_stack_fwd(x.data_ptr(), arr, len(params), _ptr(m), int(zero_padded_rows), out.data_ptr(), B, S, E,
                                        n_heads, dim_ff, eps, ws.data_ptr(), ws.numel(), _stream()), "ocv_encoder_stack_fwd")
    return out


# ---------------------------------------------------------------------------
# patch embedding / pixel-wise dot / bin head
# ---------------------------------------------------------------------------
def _map4(t: torch.Tensor, name: str) -> Tuple[torch.Tensor, int]:
    """[B, C, h, w] feature map that is dense either as NCHW or as NHWC (torch channels_last):
    -> (tensor, channels_last flag).  Anything else is made NCHW-contiguous."""
    _req(t, name, contiguous=False)
    if t.dim() != 4:
        raise ValueError(f"{name}: expected [B, C, h, w]")
    if t.is_contiguous():
        return t, 0
    if t.is_contiguous(memory_format=torch.channels_last) and t.shape[1] % 64 == 0:
        return t, 1
    return t.contiguous(), 0


class ChannelsLastWeight:
    """Per-owner cache of a conv weight in channels_last storage order [E, kh, kw, C].  Owned by the module that owns
    the parameter (so the key (data_ptr, version) cannot alias another, already freed tensor)."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, w: torch.Tensor) -> torch.Tensor:
        if w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous():
            return w
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if key != self._key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight re-layout during graph capture: run one eager warm-up call first")
            self._val = w.detach().contiguous(memory_format=torch.channels_last)
            self._key = key
        return self._val


def patch_embed(fmap: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                pos: Optional[torch.Tensor], cl_cache: Optional[ChannelsLastWeight] = None) -> torch.Tensor:
    """tokens [B, S, E] = conv16x16/16(fmap) flattened + bias + pos; pos is [S, E] or [B, S, E].
    fmap may be NCHW-contiguous or channels_last (then the weight is consumed in channels_last order; pass the
    owner's ``cl_cache`` to avoid re-laying it out on every call)."""
    lib = _lib.load()
    fmap, cl = _map4(fmap, "fmap")
    _req(weight, "weight", contiguous=False)
    B, Cc, h, w = fmap.shape
    E = weight.shape[0]
    if weight.shape != (E, Cc, 16, 16):
        raise ValueError(f"patch_embed: weight {tuple(weight.shape)} does not match fmap channels {Cc} / 16x16 patches")
    if cl:
        weight = cl_cache.get(weight) if cl_cache is not None else weight.contiguous(memory_format=torch.channels_last)
    else:
        weight = weight.contiguous()
    gh, gw = h // 16, w // 16
    S = gh * gw
    if S < 1:
        raise ValueError("patch_embed: feature map smaller than one patch")
    pos_bs = 0
    if pos is not None:
        _req(pos, "pos")
        if pos.shape == (S, E):
            pos_bs = 0
        elif pos.shape == (B, S, E):
            pos_bs = S * E
        else:
            raise ValueError(f"patch_embed: pos must be {(S, E)} or {(B, S, E)}, got {tuple(pos.shape)}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != E:
            raise ValueError("patch_embed: bias size mismatch")
    nb = lib.ocv_patch_embed_workspace_bytes(B, Cc, h, w, E)
    if nb == 0:
        raise ValueError(f"patch_embed: unsupported configuration B={B} C={Cc} h={h} w={w} E={E}")
    ws = workspace(nb, fmap.device)
    out = torch.empty(B, S, E, dtype=torch.float32, device=fmap.device)
    with timed("patch_embed"):
        check(lib.ocv_patch_embed_fwd(fmap.data_ptr(), cl, weight.data_ptr(), _ptr(bias), _ptr(pos), pos_bs, out.data_ptr(),
                                      B, Cc, h, w, E, ws.data_ptr(), ws.numel(), _stream()), "ocv_patch_embed_fwd")
    return out


class PatchEmbedSplitWeight:
    """Per-owner cache of a 16x16 patch-embedding weight in the operand order of ocv_patch_embed_split_fwd (bf16 hi / lo,
    [16, E, 16 C]), keyed on (data_ptr, version) like every other weight cache here."""

    def __init__(self):
        self._key = None
        self._vals = {}                     # f16 -> prepared weight: BOTH element types stay alive side by side (a captured graph
                                            # of the fp16 route and its bf16 fallback graph hold their addresses)

    def get(self, w: torch.Tensor, f16: bool = False):
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if key != self._key:
            self._vals = {}
            self._key = key
        f16 = bool(f16)
        if f16 not in self._vals:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
            # (None: the weight does not fit fp16 pairs -- judged once per weight version, a host synchronisation)
            self._vals[f16] = prep_patch_embed_weight(w, f16) if not f16 or fp16_weight_safe(w.detach().flatten(1)) else None
        return self._vals[f16]


def patch_embed_auto(fmap: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], pos: Optional[torch.Tensor],
                     cl_cache: Optional[ChannelsLastWeight], split_cache: Optional[PatchEmbedSplitWeight]) -> torch.Tensor:
    """The patch embedding of a feature map: when the map carries its split copy (``fmap._ocv_split``, left there by the
    decoder's last convolution) the split-bf16 form reads that (0.39 -> 0.2 ms at bs = 16); else, or with
    OCV_PATCH_EMBED=exact in the environment, the exact-fp32 kernel reads the fp32 map."""
    pre = getattr(fmap, "_ocv_split", None)
    mode = os.environ.get("OCV_PATCH_EMBED", "split")
    if mode not in ("split", "exact"):
        raise ValueError(f"OCV_PATCH_EMBED={mode!r}: expected 'split' (default) or 'exact'")
    if (pre is not None and mode == "split" and split_cache is not None and tuple(pre.shape) == tuple(fmap.shape)
            and patch_embed_split_supported(fmap.shape[0], fmap.shape[1], fmap.shape[2], fmap.shape[3], weight.shape[0])):
        if not pre.f16:
            hi, lo = split_cache.get(weight, False)
            return patch_embed_split(pre, hi, lo, bias, pos)
        prep = split_cache.get(weight, True)
        if prep is not None:
            return patch_embed_split(pre, prep[0], prep[1], bias, pos, oscale=prep[2])
        ROUTE_REPORT["patch_embed"] = "weights do not fit fp16 pairs (column spread > 2^17): exact-fp32 kernel on the fp32 map"
    return patch_embed(fp32_map(fmap), weight, bias, pos, cl_cache=cl_cache)


def prep_patch_embed_weight(weight: torch.Tensor, f16: bool = False):
    """[E, C, 16, 16] fp32 -> the two-term split [16 (ky), E, 16 C] with column kx * C + c: the operand order of
    ocv_patch_embed_split_fwd.  f16 = False: (w_hi, w_lo) bf16; f16 = True: (w_hi, w_lo, oscale) fp16 pairs of W * 2^k[e] and
    oscale [E] = 2^-k, as ``prep_conv_weight``.  Done once per weight version by the callers (cached there)."""
    E, Cc, kh, kw = weight.shape
    if (kh, kw) != (16, 16) or Cc % 32 != 0:
        raise ValueError("prep_patch_embed_weight: needs a 16x16 kernel and a multiple of 32 input channels")
    w = weight.detach().float().permute(2, 0, 3, 1).reshape(16, E, 16 * Cc)
    if not f16:
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        return hi.contiguous(), lo.contiguous()
    amax = w.abs().amax(dim=(0, 2))
    k = torch.where(amax > 0, torch.round(8.0 - torch.log2(amax.clamp_min(1e-30))), torch.zeros_like(amax)).clamp(-100.0, 100.0)
    w = w * torch.exp2(k)[None, :, None]
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous(), torch.exp2(-k).float().contiguous()


def patch_embed_split_supported(B: int, Cc: int, h: int, w: int, E: int) -> bool:
    return (h % 16 == 0 or B == 1) and int(_lib.load().ocv_patch_embed_split_workspace_bytes(B, Cc, h, w, E)) > 0


def patch_embed_split(fmap: "SplitAct", w_hi: torch.Tensor, w_lo: torch.Tensor, bias: Optional[torch.Tensor],
                      pos: Optional[torch.Tensor], oscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokens [B, S, E] = conv16x16/16(fmap) flattened + bias + pos on a feature map held in the hl32 split layout
    (ocv_patch_embed_split_fwd: 16 two-term-split GEMMs in one launch of the convolution kernel + a fixed-order sum).
    w_hi / w_lo (/ oscale) from ``prep_patch_embed_weight`` in the map's element type; pos is [S, E] or [B, S, E]."""
    lib = _lib.load()
    dt = fmap.hl.dtype
    _req(fmap.hl, "fmap.hl", dt)
    B, Cc, h, w = fmap.shape
    for n, t in (("w_hi", w_hi), ("w_lo", w_lo)):
        _req(t, n, dt)
    if w_hi.dim() != 3 or w_hi.shape[0] != 16 or w_hi.shape[2] != 16 * Cc or w_lo.shape != w_hi.shape:
        raise ValueError(f"patch_embed_split: weights {tuple(w_hi.shape)} do not match {Cc} channels / 16x16 patches")
    E = w_hi.shape[1]
    if oscale is not None:
        _req(oscale, "oscale")
        if oscale.numel() != E:
            raise ValueError("patch_embed_split: oscale size mismatch")
    gh, gw = h // 16, w // 16
    S = gh * gw
    nb = int(lib.ocv_patch_embed_split_workspace_bytes(B, Cc, h, w, E))
    if nb == 0 or not (h % 16 == 0 or B == 1):
        raise ValueError(f"patch_embed_split: unsupported configuration B={B} C={Cc} h={h} w={w} E={E}")
    pos_bs = 0
    if pos is not None:
        _req(pos, "pos")
        if pos.shape == (S, E):
            pos_bs = 0
        elif pos.shape == (B, S, E):
            pos_bs = S * E
        else:
            raise ValueError(f"patch_embed_split: pos must be {(S, E)} or {(B, S, E)}, got {tuple(pos.shape)}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != E:
            raise ValueError("patch_embed_split: bias size mismatch")
    ws = workspace(nb, fmap.hl.device, "patch_embed_split")
    out = torch.empty(B, S, E, dtype=torch.float32, device=fmap.hl.device)
    with timed("patch_embed"):
        check(lib.ocv_patch_embed_split_fwd(fmap.hl.data_ptr(), Cc, w_hi.data_ptr(), w_lo.data_ptr(), _ptr(oscale), int(fmap.f16),
                                            _ptr(bias), _ptr(pos), pos_bs, out.data_ptr(), B, h, w, E, ws.data_ptr(), ws.numel(),
                                            _stream()), "ocv_patch_embed_split_fwd")
    return out


def _check_queries(queries: torch.Tensor, B: int, Cc: int) -> None:
    _req(queries, "queries", contiguous=False)
    if queries.dim() != 3 or queries.shape[0] != B or queries.shape[2] != Cc or queries.stride(2) != 1:
        raise ValueError("queries: expected [B, Q, C] with unit stride on C")


def pixel_dot(feat: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """PixelWiseDotProduct: [B,C,h,w] (NCHW or channels_last) x [B,Q,C] -> [B,Q,h,w] (NCHW-contiguous)."""
    lib = _lib.load()
    feat, cl = _map4(feat, "feat")
    B, Cc, h, w = feat.shape
    _check_queries(queries, B, Cc)
    Q = queries.shape[1]
    ram = torch.empty(B, Q, h, w, dtype=torch.float32, device=feat.device)
    with timed("pixel_dot"):
        check(lib.ocv_pixel_dot_fwd(feat.data_ptr(), cl, queries.data_ptr(), queries.stride(0), queries.stride(1),
                                    ram.data_ptr(), B, Cc, Q, h * w, _stream()), "ocv_pixel_dot_fwd")
    return ram


def bin_head(feat: torch.Tensor, queries: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor,
             centers: torch.Tensor, exact: bool = False) -> torch.Tensor:
    """depth [B,1,h,w] = sum_k softmax_k(conv1x1(pixel_dot(feat, queries)))_k * centers_k, fused.
    feat NCHW-contiguous: exact fp32 MFMA.  feat channels_last: logits as a TWO-term fp16 split with a scaled low term (22-bit
    products at the error of an fp32 FMA chain, three MFMAs per block, all 256 bins per workgroup: OCV_BINHEAD=h2, the default),
    as a THREE-term bf16 split (OCV_BINHEAD=split3: six MFMAs, two bin halves + a merge launch; fp32's RANGE -- also what a
    forward inside ``bf16_pairs()``, the range guard's fallback, takes), or on the exact fp32 MFMA kernel with ``exact=True`` /
    OCV_BINHEAD=exact."""
    lib = _lib.load()
    mode = os.environ.get("OCV_BINHEAD", "h2")              # read per call
    if mode not in ("h2", "split3", "exact"):
        raise ValueError(f"OCV_BINHEAD={mode!r}: expected 'h2' (default), 'split3' or 'exact'")
    if mode == "h2" and _TLS.bf16_pairs:
        mode = "split3"                                     # a batch beyond the fp16 pairs' range: the head with fp32's range
    exact = exact or mode == "exact"
    feat, cl = _map4(feat, "feat")
    _req(b_out, "b_out"); _req(centers, "centers")
    B, Cc, h, w = feat.shape
    _check_queries(queries, B, Cc)
    Q = queries.shape[1]
    w2 = _req(w_out.reshape(w_out.shape[0], -1), "w_out")
    nbins = w2.shape[0]
    if w2.shape != (nbins, Q) or b_out.numel() != nbins or centers.shape != (B, nbins):
        raise ValueError("bin_head: parameter shape mismatch")
    nb = lib.ocv_bin_head_workspace_bytes(B, nbins, Cc)
    if nb == 0:
        raise ValueError(f"bin_head: unsupported configuration C={Cc} Q={Q} n_bins={nbins}")
    ws = workspace(nb, feat.device, "bin_head")
    depth = torch.empty(B, 1, h, w, dtype=torch.float32, device=feat.device)
    wf = ws.view(torch.float32)
    check(lib.ocv_bin_head_fold_fwd(queries.data_ptr(), queries.stride(0), queries.stride(1), w2.data_ptr(), wf.data_ptr(), B,
                                    Cc, Q, nbins, _stream()), "ocv_bin_head_fold_fwd")
    route = 0 if not cl else (1 if exact else (3 if mode == "h2" else 2))          # include/objcavit_hip.h: ocv_bin_head_folded_fwd
    npart = int(lib.ocv_bin_head_partials_bytes(B, h * w)) if route == 2 else 0
    part = workspace(npart, feat.device, "bin_head_partials") if npart else None
    with timed("bin_head"):          # the logit / softmax / depth launch(es): one, or the split-3 halves + merge
        check(lib.ocv_bin_head_folded_ws_fwd(feat.data_ptr(), route, wf.data_ptr(), b_out.data_ptr(),
                                             centers.data_ptr(), depth.data_ptr(), B, Cc, nbins, h * w, _ptr(part), npart,
                                             _stream()), "ocv_bin_head_folded_ws_fwd")
    return depth


# ---------------------------------------------------------------------------
# bin widths -> edges -> centres (csrc/bin_edges.hip)
# ---------------------------------------------------------------------------
BINNORM = {"linear": 0, "sigmoid": 1, "none": 2}


def bin_edges(raw: torch.Tensor, norm: str, min_depth: float, max_depth: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """raw [B, n_bins] (the regressor's output; ``norm`` = 'linear' | 'sigmoid', or 'none' for rows that are normalised already) ->
    (bin_widths_normed [B, n_bins], bin_edges [B, n_bins + 1], centers [B, n_bins]) in one launch."""
    lib = _lib.load()
    _req(raw, "raw")
    if raw.dim() != 2 or norm not in BINNORM:
        raise ValueError("bin_edges: raw must be [B, n_bins] and norm one of " + ", ".join(BINNORM))
    B, n = raw.shape
    w = torch.empty_like(raw)
    e = torch.empty(B, n + 1, dtype=torch.float32, device=raw.device)
    c = torch.empty_like(raw)
    check(lib.ocv_bin_edges_fwd(raw.data_ptr(), BINNORM[norm], float(min_depth), float(max_depth), w.data_ptr(), e.data_ptr(),
                                c.data_ptr(), B, n, _stream()), "ocv_bin_edges_fwd")
    return w, e, c


# ---------------------------------------------------------------------------
# ragged object lists with device-resident counts (csrc/objects_pad.hip)
# ---------------------------------------------------------------------------
def _counts_i32(counts: torch.Tensor, B: int) -> torch.Tensor:
    _req(counts, "counts", torch.int32)
    if counts.shape != (B,):
        raise ValueError(f"counts: expected int32 [{B}], got {tuple(counts.shape)}")
    return counts


def object_tokens_pad(tokens: torch.Tensor, counts: torch.Tensor, pad_value: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """tokens [B, cap, E] (rows >= counts[b] arbitrary) -> (tokens with those rows set to ``pad_value``, uint8 mask [B, cap] with
    1 = padding): pad_sequence(..., padding_value) + the key-padding mask of modules/ObjCAViT.py:180-183, counts on the device."""
    lib = _lib.load()
    _req(tokens, "tokens")
    if tokens.dim() != 3:
        raise ValueError("object_tokens_pad: tokens must be [B, capacity, E]")
    B, cap, E = tokens.shape
    _counts_i32(counts, B)
    out = torch.empty_like(tokens)
    mask = torch.empty(B, cap, dtype=torch.uint8, device=tokens.device)
    check(lib.ocv_object_tokens_pad_fwd(tokens.data_ptr(), counts.data_ptr(), float(pad_value), out.data_ptr(), mask.data_ptr(),
                                        B, cap, E, _stream()), "ocv_object_tokens_pad_fwd")
    return out, mask


def object_front_pad(objects: torch.Tensor, counts: torch.Tensor, S: int, pad_value: float, group: Optional[int] = None,
                     nmax: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """objects [B, cap, E] -> (keys [B, S, E] with the rows padded at the FRONT to S, uint8 mask [B, S] = (j >= counts[b])):
    modules/ObjCAViT.py:192-194 with Nmax = the longest list of the image's group (``group`` consecutive images = one call of
    the reference; None = the whole batch) or ``nmax`` when given (> 0)."""
    lib = _lib.load()
    _req(objects, "objects")
    if objects.dim() != 3:
        raise ValueError("object_front_pad: objects must be [B, capacity, E]")
    B, cap, E = objects.shape
    _counts_i32(counts, B)
    if cap > S:
        raise ValueError(f"more objects per image ({cap}) than image tokens ({S})")
    if nmax < 0 or nmax > cap:
        raise ValueError(f"object_front_pad: nmax = {nmax} outside [0, capacity = {cap}]")
    out = torch.empty(B, S, E, dtype=torch.float32, device=objects.device)
    kpm = torch.empty(B, S, dtype=torch.uint8, device=objects.device)
    check(lib.ocv_object_front_pad_fwd(objects.data_ptr(), counts.data_ptr(), int(group or B), int(nmax), float(pad_value),
                                       out.data_ptr(), kpm.data_ptr(), B, cap, int(S), E, _stream()), "ocv_object_front_pad_fwd")
    return out, kpm


# ---------------------------------------------------------------------------
# positional-embedding samplers (GridRandomPositionalEmbeddings)
# ---------------------------------------------------------------------------
POS_CENTRE_OBJ, POS_CENTRE_IMG, POS_ROI = 0, 1, 2


def pos_grid_sample(table: torch.Tensor, gh: int, gw: int, coords: torch.Tensor, mode: int, p0: float, p1: float = 0.0,
                    rows_per_image: int = 1, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[n, E] samples of the gh x gw grid held in the first gh*gw rows of ``table`` [L, E] at ``coords`` [n, >=2|4]
    (include/objcavit_hip.h, ocv_pos_grid_sample_fwd); ``addend`` [n, E] is added to the samples when given.
    No host synchronisation, no data-dependent launch shape."""
    lib = _lib.load()
    _req(table, "table"); _req(coords, "coords", contiguous=False)
    if table.dim() != 2 or coords.dim() != 2:
        raise ValueError("pos_grid_sample: table must be [L, E], coords [n, k]")
    L, E = table.shape
    n, k = coords.shape
    if coords.stride(1) != 1 or (n > 1 and coords.stride(0) < k):
        coords = coords.contiguous()
    ld = coords.stride(0) if n > 1 else k           # rows may be a column slice of a wider matrix (xywh[:, 0:2])
    if gh < 1 or gw < 1 or gh * gw > L:
        raise ValueError(f"pos_grid_sample: a {gh} x {gw} grid needs {gh * gw} table rows, the table has {L}")
    if k < (4 if mode == POS_ROI else 2):
        raise ValueError(f"pos_grid_sample: coords with {k} columns are too narrow for mode {mode}")
    if addend is not None:
        _req(addend, "addend")
        if addend.shape != (n, E):
            raise ValueError(f"pos_grid_sample: addend must be {(n, E)}, got {tuple(addend.shape)}")
    out = torch.empty(n, E, dtype=torch.float32, device=table.device)
    if n == 0:
        return out
    with timed("pos_grid_sample"):
        check(lib.ocv_pos_grid_sample_fwd(table.data_ptr(), gh, gw, E, coords.data_ptr(), ld, n, mode, float(p0), float(p1),
                                          int(rows_per_image), _ptr(addend), out.data_ptr(), _stream()),
              "ocv_pos_grid_sample_fwd")
    return out


# ---------------------------------------------------------------------------
# depthwise convolution (EfficientNet MBConv)
# ---------------------------------------------------------------------------
def depthwise_conv_same(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
                        act: int = ACT_NONE) -> torch.Tensor:
    """Depthwise k x k conv with TensorFlow 'SAME' padding, + bias (folded BN) + optional SiLU.
    x [B,C,H,W] NCHW-contiguous, weight [C,1,k,k]."""
    lib = _lib.load()
    _req(x, "x"); _req(weight, "weight")
    B, Cc, H, W = x.shape
    k = weight.shape[-1]
    if weight.shape != (Cc, 1, k, k):
        raise ValueError(f"depthwise_conv_same: weight {tuple(weight.shape)} does not match {Cc} channels")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cc:
            raise ValueError("depthwise_conv_same: bias size mismatch")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    out = torch.empty(B, Cc, Ho, Wo, dtype=torch.float32, device=x.device)
    with timed("depthwise"):
        check(lib.ocv_depthwise_conv_fwd(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), B, Cc, H, W, k, stride,
                                         ph // 2, pw // 2, Ho, Wo, act, _stream()), "ocv_depthwise_conv_fwd")
    return out


# ---------------------------------------------------------------------------
# split-bf16 implicit-GEMM convolution on NHWC activations
# ---------------------------------------------------------------------------
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


def fp16_weight_safe(w2d: torch.Tensor) -> bool:
    """Whether a weight matrix [N, K] keeps at least bf16-pair precision (16 bits) on EVERY entry as fp16 pairs after its rows have
    been scaled to a largest entry near 2^8: an entry more than 2^17 below its row's largest has fewer than five low-term bits
    left above fp16's subnormal step.  Judged per input column (a column that is small in every row = an input channel whose
    weights are tiny next to the others', which matters exactly when its activations are huge); all-zero columns are fine."""
    w = w2d.detach().abs().double()
    rmax = w.amax(dim=1, keepdim=True).clamp_min(1e-300)
    col = (w / rmax).amax(dim=0)
    col = col[col > 0]
    return bool(col.numel() == 0 or float(col.min()) >= 2.0 ** -17)


def prep_conv_weight(weight: torch.Tensor, f16: bool = False):
    """[Cout, Cin, k, k] fp32 -> the two-term split in the kernels' order [k*k, Cout, Cp], Cp = Cin rounded up to 32 (zero padded).
    f16 = False: (w_hi, w_lo) bf16, w_hi = bf16(W), w_lo = bf16(W - w_hi).
    f16 = True:  (w_hi, w_lo, oscale): fp16 pairs of W * 2^k[n], k[n] the power of two that puts output channel n's largest entry
      in [2^7.5, 2^8.5) (out of fp16's subnormals: BN-folded weights of ~0.02 would otherwise have low terms of 1e-5, below the
      6e-5 where fp16 stops being normal), and oscale [Cout] fp32 = 2^-k for the kernel's epilogue (exact).
    Done once per weight version by the callers (cached there)."""
    Cout, Cin, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("prep_conv_weight: kernel must be 1x1 or 3x3")
    w = weight.detach().float().permute(2, 3, 0, 1).reshape(kh * kw, Cout, Cin)
    Cp = (Cin + 31) // 32 * 32
    if not f16:
        if Cp != Cin:
            w = torch.nn.functional.pad(w, (0, Cp - Cin))
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        return hi.contiguous(), lo.contiguous()
    amax = w.abs().amax(dim=(0, 2))                                            # [Cout]
    k = torch.where(amax > 0, torch.round(8.0 - torch.log2(amax.clamp_min(1e-30))), torch.zeros_like(amax)).clamp(-100.0, 100.0)
    w = w * torch.exp2(k)[None, :, None]
    if Cp != Cin:
        w = torch.nn.functional.pad(w, (0, Cp - Cin))
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous(), torch.exp2(-k).float().contiguous()


def conv_nhwc_exact(x1: torch.Tensor, x2: Optional[torch.Tensor], w_tap_major: torch.Tensor, bias: Optional[torch.Tensor],
                    ksize: int, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(conv_kxk(cat([x1, x2], 1)) + bias) (+ residual) in exact fp32 (ocv_conv_nhwc_exact_fwd); logical shapes
    [B, C, H, W], storage channels_last; w_tap_major fp32 [k*k, Cout, C1+C2] = weight.permute(2, 3, 0, 1)."""
    lib = _lib.load()
    x1 = _nhwc(x1, "x1")
    B, C1, H, W = x1.shape
    C2 = 0
    if x2 is not None:
        x2 = _nhwc(x2, "x2")
        if x2.shape[0] != B or x2.shape[2:] != x1.shape[2:]:
            raise ValueError("conv_nhwc_exact: x2 must match x1 in batch and spatial size")
        C2 = x2.shape[1]
    _req(w_tap_major, "w_tap_major")
    if w_tap_major.dim() != 3 or w_tap_major.shape[0] != ksize * ksize or w_tap_major.shape[2] != C1 + C2:
        raise ValueError(f"conv_nhwc_exact: weights {tuple(w_tap_major.shape)} do not match {C1}+{C2} input channels, k={ksize}")
    Cout = w_tap_major.shape[1]
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("conv_nhwc_exact: bias size mismatch")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x1.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = _nhwc(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("conv_nhwc_exact: residual shape mismatch")
    with timed(f"conv{ksize}x{ksize}x|{B},{H},{W},{C1 + C2},{Cout}"):
        check(lib.ocv_conv_nhwc_exact_fwd(x1.data_ptr(), C1, _ptr(x2), C2, w_tap_major.data_ptr(), _ptr(bias), _ptr(residual),
                                          y.data_ptr(), B, H, W, Cout, ksize, act, _stream()), "ocv_conv_nhwc_exact_fwd")
    return y


def tap_interp_supported(h: int, w: int, H: int, W: int, Cout: int) -> bool:
    return bool(_lib.load().ocv_tap_interp_supported(int(h), int(w), int(H), int(W), int(Cout)))


def tap_interp_combine(z: torch.Tensor, s: Optional[torch.Tensor], bias: Optional[torch.Tensor], size: Tuple[int, int],
                       act: int = ACT_NONE, out_fp32: bool = True, out_split: bool = False,
                       border: Optional[torch.Tensor] = None, split_f16: bool = False):
    """act(bias + s + sum over the 9 taps of the bilinear (align_corners) interpolation of z's tap products at the tap
    position): ocv_tap_interp_combine_fwd.  z [B, 9 Cout, h, w] channels_last (tap-major columns), s [B, Cout, H, W]
    channels_last or None.  ``border`` [9 Cout]: z is the interior of an (h+2) x (w+2) grid whose border ring holds this
    vector (Decoder.conv2's padding).  ``split_f16``: element type of the split output.
    Returns fp32 tensor, SplitAct, or (fp32, SplitAct)."""
    lib = _lib.load()
    if not (out_fp32 or out_split):
        raise ValueError("tap_interp_combine: nothing to output")
    z = _nhwc(z, "z")
    B, C9, h, w = z.shape
    if C9 % 9 != 0:
        raise ValueError("tap_interp_combine: z must have 9 * Cout channels")
    Cout = C9 // 9
    H, W = int(size[0]), int(size[1])
    zpad = 0
    if border is not None:
        _req(border, "border")
        if border.numel() != C9:
            raise ValueError("tap_interp_combine: border must hold 9 * Cout values")
        zpad, h, w = 1, h + 2, w + 2
    if s is not None:
        s = _nhwc(s, "s")
        if tuple(s.shape) != (B, Cout, H, W):
            raise ValueError(f"tap_interp_combine: s must be {(B, Cout, H, W)}, got {tuple(s.shape)}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("tap_interp_combine: bias size mismatch")
    if not tap_interp_supported(h, w, H, W, Cout):
        raise ValueError(f"tap_interp_combine: unsupported resize {h}x{w} -> {H}x{W} / channel count {Cout}")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=z.device, memory_format=torch.channels_last) if out_fp32 else None
    ys = SplitAct.empty(B, Cout, H, W, z.device, f16=split_f16) if out_split else None
    with timed(f"tap_interp|{B},{H},{W},{Cout}"):
        check(lib.ocv_tap_interp_combine_x_fwd(z.data_ptr(), h, w, zpad, _ptr(border), _ptr(s), _ptr(bias), _ptr(y),
                                               ys.hl.data_ptr() if out_split else None, int(bool(split_f16)), B, H, W, Cout, act,
                                               _stream()), "ocv_tap_interp_combine_fwd")
    _note_range(f"tap_interp|{B},{H},{W},{Cout}", ys)
    if out_fp32 and out_split:
        return y, ys
    return y if out_fp32 else ys


_NAN: Dict["torch.device", torch.Tensor] = {}


def split_only_enabled() -> bool:
    """Whether the decoder may hand the heads its output in split form ONLY (``map_placeholder``): the default; off with
    OCV_PATCH_EMBED=exact (that A/B route reads the fp32 map) or OCV_DECODER_FP32=1."""
    return os.environ.get("OCV_PATCH_EMBED", "split") == "split" and os.environ.get("OCV_DECODER_FP32", "0") != "1"


def map_placeholder(split: "SplitAct") -> torch.Tensor:
    """The decoder's output when both heads' consumers -- the 16x16 patch embedding and the 3x3 convolution -- take its split copy
    (``_ocv_split``): a [B, C, H, W] tensor of the right shape and device WITHOUT storage of its own (one NaN, stride 0), so the
    convolution that produces the map writes 4 bytes per value instead of 8 (629 MB less per step at bs 16).  Anything that does read
    the fp32 values (the reported fallbacks: weights that do not fit fp16 pairs) goes through ``fp32_map`` first; a read that
    forgets to is NaN, not silently wrong."""
    B, Cc, H, W = split.shape
    dev = split.hl.device
    nan = _NAN.get(dev)
    if nan is None:                                        # (one scalar per device, made once: no fill launch per forward / replay)
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("map_placeholder: run one eager warm-up call before capture")
        nan = _NAN[dev] = torch.full((1,), float("nan"), dtype=torch.float32, device=dev)
    t = nan.expand(B, Cc, H, W)
    t._ocv_split = split
    t._ocv_fp32_missing = True
    return t


def fp32_map(fmap: torch.Tensor) -> torch.Tensor:
    """``fmap`` itself, or -- for a ``map_placeholder`` -- the fp32 map rebuilt from its split copy (hi + lo: 22 bits from fp16
    pairs, 16 from bf16 pairs), channels_last, carrying the same split copy."""
    if not getattr(fmap, "_ocv_fp32_missing", False):
        return fmap
    sp = fmap._ocv_split
    B, Cc, H, W = sp.shape
    v = sp.hl.view(B, H, W, -1, 2, 32).float()
    out = (v[..., 0, :] + v[..., 1, :]).reshape(B, H, W, -1)[..., :Cc].permute(0, 3, 1, 2)      # NHWC storage = channels_last
    out = out.contiguous(memory_format=torch.channels_last)
    out._ocv_split = sp
    return out


def split_act(x: torch.Tensor, f16: bool = False) -> "SplitAct":
    """fp32 channels_last activation -> hl32 split (the resize kernel at scale 1); ``f16``: fp16 pairs instead of bf16 pairs."""
    return upsample_concat_split(x, None, tuple(x.shape[-2:]), f16=f16)


_WINO43_G = ((1.0, 0.0, 0.0), (-1 / 3, -1 / 3, -1 / 3), (1 / 3, -1 / 3, 1 / 3), (1 / 15, 2 / 15, 4 / 15), (-16 / 15, 8 / 15, -4 / 15), (0.0, 0.0, 1.0))


def prep_winograd43_weight(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """[Cout, Cin, 3, 3] fp32 -> (u_hi, u_lo fp16 [36, Cout, Cp], fscale fp32 [36], cscale fp32 [Cp]): the Winograd F(4x4, 3x3)
    filter transform U[6 i + j] = (G g G^T)[i][j] in fp64, scaled by two sets of powers of two before it is split:
    per POSITION 2^k (the transform's entries go down to 1/576 of the filter's: unscaled, their low terms fall into fp16's
    subnormals and the result is 100x less accurate) and per INPUT CHANNEL 2^-a (a channel whose weights are tiny because its
    activations are huge -- or the reverse -- would otherwise have one of the two operands at the edge of fp16's range; the
    input transform multiplies the channel's activations by cscale = 2^a, so products are unchanged).  Both are chosen so that the
    largest entry of every position and of every channel sits near 2^8; hi = fp16(U'), lo = fp16(U' - hi); fscale = 2^-k is
    applied to the raw GEMM results by the output transform.  Cp = Cin rounded up to 32 (cscale 1 on the pad channels).
    Once per weight version (cached by the callers)."""
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError("prep_winograd43_weight: kernel must be 3x3")
    G = torch.tensor(_WINO43_G, dtype=torch.float64, device=weight.device)
    u = torch.einsum("ia,ocab,jb->ijoc", G, weight.detach().double(), G).reshape(36, Cout, Cin)
    # channel equalisation first (on the position-normalised magnitudes), then the position scale on what is left
    pmax = u.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-300)
    cmax = (u.abs() / pmax).amax(dim=(0, 1))                                   # [Cin], <= 1
    a = torch.where(cmax > 0, torch.round(torch.log2(cmax.clamp_min(1e-300))), torch.zeros_like(cmax)).clamp(-60.0, 60.0)
    a = a - a.max()                                                            # the largest channel keeps its scale
    u = u * torch.exp2(-a)[None, None, :]
    amax = u.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    k = torch.round(8.0 - torch.log2(amax))
    u = (u * torch.exp2(k)[:, None, None]).float()
    Cp = (Cin + 31) // 32 * 32
    cscale = torch.exp2(a).float()
    if Cp != Cin:
        u = torch.nn.functional.pad(u, (0, Cp - Cin))
        cscale = torch.nn.functional.pad(cscale, (0, Cp - Cin), value=1.0)
    hi = u.to(torch.float16)
    lo = (u - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous(), torch.exp2(-k).float().contiguous(), cscale.contiguous()


def conv3x3_winograd43_split(x: "SplitAct", u_hi: torch.Tensor, u_lo: torch.Tensor, fscale: torch.Tensor, bias: Optional[torch.Tensor],
                             act: int = ACT_NONE, out_fp32: bool = True, out_split: bool = False, cscale: Optional[torch.Tensor] = None):
    """3x3 convolution (stride 1, padding 1) of a pre-split activation in Winograd F(4x4, 3x3) form on two-term fp16 splits
    (ocv_conv3x3_winograd43_split_fwd).  Returns fp32 tensor, SplitAct, or (fp32, SplitAct) like conv_nhwc_split."""
    lib = _lib.load()
    if not (out_fp32 or out_split):
        raise ValueError("conv3x3_winograd43_split: nothing to output")
    _req(x.hl, "x.hl", x.hl.dtype)
    B, Cin, H, W = x.shape
    Cp = (Cin + 31) // 32 * 32
    if x.hl.dim() != 4 or x.hl.shape[3] != 2 * Cp:
        raise ValueError("conv3x3_winograd43_split: x.hl must be [B, H, W, 2 * ceil32(C)]")
    for n, t in (("u_hi", u_hi), ("u_lo", u_lo)):
        _req(t, n, torch.float16)
    _req(fscale, "fscale")
    if cscale is not None:
        _req(cscale, "cscale")
        if cscale.numel() != Cp:
            raise ValueError(f"conv3x3_winograd43_split: cscale must hold {Cp} values (Cin rounded up to 32)")
    if u_hi.dim() != 3 or u_hi.shape[0] != 36 or u_hi.shape[2] != Cp or u_lo.shape != u_hi.shape or fscale.numel() != 36:
        raise ValueError(f"conv3x3_winograd43_split: transformed weights {tuple(u_hi.shape)} do not match {Cin} input channels")
    Cout = u_hi.shape[1]
    if Cout % 8 != 0:
        raise ValueError("conv3x3_winograd43_split: Cout must be a multiple of 8")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("conv3x3_winograd43_split: bias size mismatch")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.hl.device, memory_format=torch.channels_last) if out_fp32 else None
    ys = SplitAct.empty(B, Cout, H, W, x.hl.device, f16=x.f16) if out_split else None
    nws = int(lib.ocv_conv3x3_winograd43_workspace_bytes(B, H, W, Cin, Cout))
    ws = workspace(nws, x.hl.device, "conv_winograd")
    with timed(f"conv3x3w4|{B},{H},{W},{Cin},{Cout}"):
        check(lib.ocv_conv3x3_winograd43_split_fwd(x.hl.data_ptr(), Cin, u_hi.data_ptr(), u_lo.data_ptr(), fscale.data_ptr(), _ptr(cscale), _ptr(bias),
                                                   _ptr(y), ys.hl.data_ptr() if out_split else None, B, H, W, Cout, act, int(x.f16),
                                                   ws.data_ptr(), ws.numel(), _stream()), "ocv_conv3x3_winograd43_split_fwd")
    _note_range(f"conv3x3w4|{B},{H},{W},{Cin},{Cout}", ys)
    if out_fp32 and out_split:
        return y, ys
    return y if out_fp32 else ys


def winograd_pays(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    """Where the Winograd F(4x4, 3x3) form of a 3x3 convolution (two-term fp16 splits inside, 36 GEMMs, 4x fewer matrix operations)
    beats the direct kernel: the transformed input and the raw result go through HBM, so the arithmetic must dominate -- the
    decoder's 30 x 40 and 60 x 80 second convolutions (1024 -> 1024: 413 us against 1000 direct; 512 -> 512: 557 against 882;
    256 -> 256 at 120 x 160 is a tie and stays direct -- tools/run_wino43.py, profiles/r03_winograd43.txt)."""
    return Cout % 8 == 0 and Cin >= 512 and Cout >= 512 and B * H * W <= 131072


def _nhwc(t: torch.Tensor, name: str) -> torch.Tensor:
    _req(t, name, contiguous=False)
    if t.dim() != 4:
        raise ValueError(f"{name}: expected a 4-D [B, C, H, W] tensor")
    if not t.is_contiguous(memory_format=torch.channels_last):
        t = t.contiguous(memory_format=torch.channels_last)
    return t


def conv_nhwc(x1: torch.Tensor, x2: Optional[torch.Tensor], w_hi: torch.Tensor, w_lo: torch.Tensor,
              bias: Optional[torch.Tensor], ksize: int, act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(conv_kxk(cat([x1, x2], 1)) + bias) (+ residual); logical shapes [B, C, H, W], storage channels_last."""
    lib = _lib.load()
    x1 = _nhwc(x1, "x1")
    B, C1, H, W = x1.shape
    C2 = 0
    if x2 is not None:
        x2 = _nhwc(x2, "x2")
        if x2.shape[0] != B or x2.shape[2:] != x1.shape[2:]:
            raise ValueError("conv_nhwc: x2 must match x1 in batch and spatial size")
        C2 = x2.shape[1]
    for n, t in (("w_hi", w_hi), ("w_lo", w_lo)):
        _req(t, n, torch.bfloat16)
    taps, Cout, Cp = w_hi.shape
    if w_lo.shape != w_hi.shape or taps != ksize * ksize or Cp != (C1 + C2 + 31) // 32 * 32:
        raise ValueError(f"conv_nhwc: weights {tuple(w_hi.shape)} do not match {C1}+{C2} input channels, k={ksize}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("conv_nhwc: bias size mismatch")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x1.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = _nhwc(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("conv_nhwc: residual shape mismatch")
    with timed(f"conv{ksize}x{ksize}|{B},{H},{W},{C1 + C2},{Cout}"):
        check(lib.ocv_conv_nhwc_fwd(x1.data_ptr(), C1, _ptr(x2), C2, w_hi.data_ptr(), w_lo.data_ptr(), _ptr(bias),
                                    _ptr(residual), y.data_ptr(), B, H, W, Cout, ksize, act, _stream()), "ocv_conv_nhwc_fwd")
    return y


# ---------------------------------------------------------------------------
# NHWC encoder blocks (pointwise conv with fused gate / bias / act / residual, depthwise, squeeze)
# ---------------------------------------------------------------------------
class SplitWeight:
    """A static [Cout, Cin] matrix pre-split for the bf16x3 kernels (hi = bf16(W), lo = bf16(W - hi)) and packed in
    matrix-core B-operand order (include/objcavit_hip.h, ocv_pointwise_conv_nhwc_split_fwd): one contiguous 1 KB
    fragment per (32-channel tile, 16-wide K step, hi|lo).  Built once per weight version by the callers (cached
    next to their BN-folded weights)."""

    def __init__(self, weight: torch.Tensor):
        w = weight.detach().float().reshape(weight.shape[0], -1)
        self.cout, self.cin = w.shape
        self.kp = (self.cin + 15) // 16 * 16
        npad = (self.cout + 31) // 32 * 32
        w = torch.nn.functional.pad(w, (0, self.kp - self.cin, 0, npad - self.cout))
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        nt, ns = npad // 32, self.kp // 16
        parts = torch.stack([hi, lo], 0).reshape(2, nt, 32, ns, 2, 8)          # [part, jt, l31, s, hh, e]
        self.packed = parts.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1)   # [jt, s, part, hh, l31, e]
        assert self.packed.numel() == nt * ns * 2 * 512


def pointwise_weight(weight: torch.Tensor):
    """What the encoder hands to pointwise_nhwc: the split form (default) or, with OCV_PW=fp32 in the environment, the
    fp32 matrix itself (exact v_mfma_f32_32x32x2_f32 path, ~3x slower from stage 4 on)."""
    if os.environ.get("OCV_PW", "split") == "fp32":
        return weight.detach().reshape(weight.shape[0], -1).contiguous()
    return SplitWeight(weight)


def pointwise_nhwc(x: torch.Tensor, weight, bias: Optional[torch.Tensor], act: int = ACT_NONE,
                   gate: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, out_split: bool = False):
    """1x1 convolution on a channels_last [B, C, H, W] tensor (or a plain [M, C] matrix): act(x*gate @ W^T + b) + res.
    weight: fp32 [Cout, Cin] (or [Cout, Cin, 1, 1]) -> exact fp32 kernel; a SplitWeight -> split-bf16 kernel.
    gate [B, Cin].  ``out_split`` (SplitWeight, 4-D x, Cout % 8 == 0): also return the hl32 split copy -> (y, SplitAct)."""
    lib = _lib.load()
    four = x.dim() == 4
    if four:
        x = _nhwc(x, "x")
        B, Cin, H, Wd = x.shape
        M, rpi = B * H * Wd, H * Wd
    else:
        _req(x, "x")
        M, Cin = x.shape
        B, rpi = M, 1
    split = isinstance(weight, SplitWeight)
    if split:
        _req(weight.packed, "weight.packed", torch.bfloat16)
        Cout, wcin = weight.cout, weight.cin
    else:
        w2 = _req(weight.reshape(weight.shape[0], -1), "weight")
        Cout, wcin = w2.shape
    if wcin != Cin:
        raise ValueError(f"pointwise_nhwc: weight with {wcin} input channels does not match {Cin} input channels")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("pointwise_nhwc: bias size mismatch")
    if gate is not None:
        _req(gate, "gate")
        if gate.shape != (B, Cin):
            raise ValueError(f"pointwise_nhwc: gate must be {(B, Cin)}, got {tuple(gate.shape)}")
    if four:
        y = torch.empty(B, Cout, H, Wd, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    else:
        y = torch.empty(M, Cout, dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _nhwc(residual, "residual") if four else _req(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("pointwise_nhwc: residual shape mismatch")
    ys = None
    if out_split:
        if not (split and four and Cout % 8 == 0):
            raise ValueError("pointwise_nhwc: out_split needs a SplitWeight, a 4-D input and Cout % 8 == 0")
        ys = SplitAct.empty(B, Cout, H, Wd, x.device)
    nws = int(lib.ocv_pointwise_split_workspace_bytes(M, Cin, Cout)) if (split and ys is None) else 0   # split-K slices (tiny batches only)
    ws = workspace(nws, x.device, "pw_splitk") if nws else None
    with timed(f"pointwise|{M},{Cin},{Cout}"):
        if split:
            check(lib.ocv_pointwise_conv_nhwc_split_ws_fwd(x.data_ptr(), _ptr(gate), rpi, weight.packed.data_ptr(),
                                                           _ptr(bias), _ptr(residual), y.data_ptr(),
                                                           ys.hl.data_ptr() if ys is not None else None, M, Cin, Cout, act,
                                                           _ptr(ws), nws, _stream()),
                  "ocv_pointwise_conv_nhwc_split_ws_fwd")
        else:
            check(lib.ocv_pointwise_conv_nhwc_fwd(x.data_ptr(), _ptr(gate), rpi, w2.data_ptr(), _ptr(bias), _ptr(residual),
                                                  y.data_ptr(), M, Cin, Cout, act, _stream()), "ocv_pointwise_conv_nhwc_fwd")
    return (y, ys) if out_split else y


class PerImageSplitWeight:
    """B packed split-bf16 matrices [Cout, Cin] (hip_ops.SplitWeight order), one per image, ``img_elems`` bf16 elements
    apart: the project weight with the image's squeeze-excite gate folded in (depthwise_se_gate_weights)."""
    __slots__ = ("packed", "cout", "cin", "img_elems", "images")

    def __init__(self, packed: torch.Tensor, cout: int, cin: int, img_elems: int, images: int):
        self.packed, self.cout, self.cin, self.img_elems, self.images = packed, int(cout), int(cin), int(img_elems), int(images)


def pointwise_hl_project_pays(B: int, rows_per_image: int, cin: int, cout: int) -> bool:
    """Whether an MBConv project convolution takes the pre-split route with the squeeze-excite gate folded into PER-IMAGE
    weights.  Measured at bs = 16 (profiles/r03_pointwise_hl_sweep.txt): the project GEMM itself gains on every late layer with
    K >= 1056 (1056 -> 176: 57 -> 46 us, 1824 -> 304: 42 -> 30, 3072 -> 512: 86 -> 71), but writing and re-reading
    B x Cout x Cin x 4 bytes of gated weights costs 6 us at 1200 rows per image (stage 5: 12 MB against 81 MB of rows) and
    10 - 18 us at 300 rows (stages 6, 7: as many bytes as the rows themselves), which eats the gain there; at K = 768 (stage 4) the
    GEMM does not gain.  So: long K and weights well under the rows' own traffic -- stage 5's six 1056 -> 176 blocks.  End to end
    (bench.py --inflight 1, same box, two rounds): 919.7 / 918.0 img/s with this route against 915.7 / 914.2 without; with the
    stage 6 - 7 layers as well 889.5 / 888.5; the EXPAND layers on the pre-split route (hl32 copies written by the project in front)
    lost end to end in every combination (896 - 915) and left the product in round 5.  A batch of 1 - 3: the fp32-row kernel with
    its K slabs shared out over workgroups and the plain gate launch win (round 4)."""
    if cin % 32 != 0 or cout % 4 != 0 or B < 4:
        return False
    return B * rows_per_image <= 32768 and cin >= 1024 and 4 * cout <= rows_per_image


def pointwise_hl(x: "SplitAct", weight, bias: Optional[torch.Tensor], act: int = ACT_NONE,
                 residual: Optional[torch.Tensor] = None, out_fp32: bool = True, out_split: bool = False):
    """1x1 convolution of a PRE-SPLIT activation (hl32, read by LDS-DMA; csrc/pointwise_hl.hip): act(x @ W^T + b) + res.
    weight: a SplitWeight (one matrix) or a PerImageSplitWeight (gate folded in per image; no tile spans two images).
    Returns the fp32 channels_last tensor, the SplitAct, or (fp32, SplitAct)."""
    lib = _lib.load()
    if not (out_fp32 or out_split):
        raise ValueError("pointwise_hl: nothing to output")
    _req(x.hl, "x.hl", torch.bfloat16)
    B, Cin, H, Wd = x.shape
    if x.hl.dim() != 4 or x.hl.shape[3] != 2 * ((Cin + 31) // 32 * 32):
        raise ValueError("pointwise_hl: x.hl must be [B, H, W, 2 * ceil32(C)] bf16")
    per_image = isinstance(weight, PerImageSplitWeight)
    if not per_image and not isinstance(weight, SplitWeight):
        raise TypeError("pointwise_hl: weight must be a SplitWeight or a PerImageSplitWeight")
    _req(weight.packed, "weight.packed", torch.bfloat16)
    if weight.cin != Cin:
        raise ValueError(f"pointwise_hl: weight with {weight.cin} input channels does not match {Cin} input channels")
    Cout = weight.cout
    if per_image and (weight.images != B or weight.packed.numel() < B * weight.img_elems):
        raise ValueError("pointwise_hl: per-image weights do not match the batch")
    if Cout % 4 != 0 or (out_split and Cout % 8 != 0):
        raise ValueError(f"pointwise_hl: unsupported channel count {Cout}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("pointwise_hl: bias size mismatch")
    M = B * H * Wd
    y = torch.empty(B, Cout, H, Wd, dtype=torch.float32, device=x.hl.device, memory_format=torch.channels_last) if out_fp32 else None
    ys = SplitAct.empty(B, Cout, H, Wd, x.hl.device) if out_split else None
    if residual is not None:
        residual = _nhwc(residual, "residual")
        if tuple(residual.shape) != (B, Cout, H, Wd):
            raise ValueError("pointwise_hl: residual shape mismatch")
    with timed(f"pointwise_hl|{M},{Cin},{Cout}"):
        check(lib.ocv_pointwise_hl_fwd(x.hl.data_ptr(), Cin, weight.packed.data_ptr(), weight.img_elems if per_image else 0,
                                       H * Wd, _ptr(bias), _ptr(residual), _ptr(y), ys.hl.data_ptr() if ys is not None else None,
                                       M, Cout, act, _stream()), "ocv_pointwise_hl_fwd")
    if out_fp32 and out_split:
        return y, ys
    return y if out_fp32 else ys


def depth_metrics(pred: torch.Tensor, gt: torch.Tensor, min_depth: float, max_depth: float,
                  crop: Optional[Tuple[int, int, int, int]] = None, pred_mirror: Optional[torch.Tensor] = None,
                  first_image_id: int = 0) -> torch.Tensor:
    """Per-image metric records [B, 10] (dp.RECORD_FIELDS) of a prediction [B,1,h,w] against ground truth [B,1,H,W]:
    clamp (+ flip-TTA average with ``pred_mirror``, the un-flipped output for the mirrored image), bilinear
    align_corners resize, nan/inf fix, validity mask and crop box (y0, y1, x0, x1), eight metrics -- one pass."""
    lib = _lib.load()
    _req(pred, "pred"); _req(gt, "gt")
    if pred.dim() != 4 or gt.dim() != 4 or pred.shape[1] != 1 or gt.shape[1] != 1 or pred.shape[0] != gt.shape[0]:
        raise ValueError("depth_metrics: expected pred [B,1,h,w] and gt [B,1,H,W]")
    if pred_mirror is not None:
        _req(pred_mirror, "pred_mirror")
        if pred_mirror.shape != pred.shape:
            raise ValueError("depth_metrics: pred_mirror must have pred's shape")
    B, _, h, w = pred.shape
    H, W = gt.shape[2:]
    y0, y1, x0, x1 = crop if crop is not None else (0, H, 0, W)
    nb = lib.ocv_depth_metrics_workspace_bytes(B, H, W)
    ws = workspace(nb, pred.device, "metrics")
    rec = torch.empty(B, 10, dtype=torch.float32, device=pred.device)
    with timed("depth_metrics"):
        check(lib.ocv_depth_metrics_fwd(pred.data_ptr(), _ptr(pred_mirror), h, w, gt.data_ptr(), H, W, float(min_depth),
                                        float(max_depth), int(y0), int(y1), int(x0), int(x1), int(first_image_id),
                                        rec.data_ptr(), B, ws.data_ptr(), ws.numel(), _stream()), "ocv_depth_metrics_fwd")
    return rec


def stem_conv_same(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
                   act: int = ACT_NONE) -> torch.Tensor:
    """Dense 3x3 convolution with TF 'SAME' padding of an NCHW image, + bias + act; returns a channels_last tensor.
    weight [Cout, Cin, 3, 3] with Cin * 9 <= 32, Cout <= 64."""
    lib = _lib.load()
    _req(x, "x")
    _req(weight, "weight")
    if x.dim() != 4 or weight.dim() != 4 or weight.shape[1] != x.shape[1] or weight.shape[2] != weight.shape[3]:
        raise ValueError("stem_conv_same: expected x [B, Cin, H, W] and weight [Cout, Cin, k, k]")
    B, Cin, H, W = x.shape
    Cout, k = weight.shape[0], weight.shape[2]
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("stem_conv_same: bias size mismatch")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    out = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with timed("stem_conv"):
        check(lib.ocv_stem_conv_fwd(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), B, Cin, H, W, Cout, k,
                                    stride, ph // 2, pw // 2, Ho, Wo, act, _stream()), "ocv_stem_conv_fwd")
    return out


def conv3x3_few_channels(x: torch.Tensor, w_taps: torch.Tensor) -> torch.Tensor:
    """3x3 / stride 1 / zero padding 1 convolution of an image with at most four channels (any dense layout: read through its
    strides) on exact fp32 FMAs; ``w_taps`` [9, C, Cout] fp32 (tap-major: weight.permute(2, 3, 1, 0)).  Raw result (no bias),
    channels_last [B, Cout, H, W]."""
    lib = _lib.load()
    _req(x, "x", contiguous=False)
    _req(w_taps, "w_taps")
    if x.dim() != 4 or w_taps.dim() != 3 or w_taps.shape[0] != 9 or w_taps.shape[1] != x.shape[1] or not 1 <= x.shape[1] <= 4:
        raise ValueError("conv3x3_few_channels: expected x [B, C <= 4, H, W] and w_taps [9, C, Cout]")
    B, C, H, W = x.shape
    Cout = int(w_taps.shape[2])
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with timed("conv_few"):
        check(lib.ocv_conv3x3_few_channels_fwd(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), w_taps.data_ptr(),
                                               out.data_ptr(), B, C, H, W, Cout, _stream()), "ocv_conv3x3_few_channels_fwd")
    return out


def depthwise_nhwc_same(x: torch.Tensor, weight_kkc: torch.Tensor, bias: Optional[torch.Tensor], k: int, stride: int,
                        act: int = ACT_NONE) -> torch.Tensor:
    """Depthwise k x k conv, TF 'SAME' padding, channels_last in / out.  weight_kkc: [k*k, C] (tap-major)."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    _req(weight_kkc, "weight")
    B, Cc, H, W = x.shape
    if weight_kkc.shape != (k * k, Cc):
        raise ValueError(f"depthwise_nhwc_same: weight {tuple(weight_kkc.shape)} does not match k={k}, C={Cc}")
    if bias is not None:
        _req(bias, "bias")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    out = torch.empty(B, Cc, Ho, Wo, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with timed(f"depthwise|{B},{H},{W},{Cc},k{k}s{stride}"):
        check(lib.ocv_depthwise_conv_nhwc_fwd(x.data_ptr(), weight_kkc.data_ptr(), _ptr(bias), out.data_ptr(), B, Cc, H, W,
                                              k, stride, ph // 2, pw // 2, Ho, Wo, act, _stream()),
              "ocv_depthwise_conv_nhwc_fwd")
    return out


def depthwise_se_gate(x: torch.Tensor, weight_kkc: torch.Tensor, bias: Optional[torch.Tensor], k: int, stride: int,
                      w1: torch.Tensor, b1: torch.Tensor, w2t: torch.Tensor, b2: torch.Tensor):
    """silu(depthwise k x k (TF 'SAME') + bias) of a channels_last tensor AND the squeeze-excite gate of that output: three
    launches (depthwise, hidden layer, gate; one for the last two where the squeeze-excite weights are small).
    Returns (y [B, C, Ho, Wo] channels_last, gate [B, C])."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    _req(weight_kkc, "weight")
    B, Cc, H, W = x.shape
    if weight_kkc.shape != (k * k, Cc):
        raise ValueError(f"depthwise_se_gate: weight {tuple(weight_kkc.shape)} does not match k={k}, C={Cc}")
    for n, t in (("bias", bias), ("w1", w1), ("b1", b1), ("w2t", w2t), ("b2", b2)):
        if t is not None:
            _req(t, n)
    R = w1.shape[0]
    if w1.shape != (R, Cc) or w2t.shape != (R, Cc) or b1.numel() != R or b2.numel() != Cc:
        raise ValueError("depthwise_se_gate: squeeze-excite parameter shape mismatch")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    tiles = lib.ocv_depthwise_sum_tiles(B, Cc, Ho, Wo, k, stride)
    if tiles <= 0:
        raise ValueError("depthwise_se_gate: unsupported shape")
    out = torch.empty(B, Cc, Ho, Wo, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    part = workspace(B * tiles * Cc * 4, x.device, "dw_part")
    gate = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    hid = workspace(B * R * 4, x.device, "se_hidden")
    with timed(f"depthwise|{B},{H},{W},{Cc},k{k}s{stride}"):
        check(lib.ocv_depthwise_conv_nhwc_sum_fwd(x.data_ptr(), weight_kkc.data_ptr(), _ptr(bias), out.data_ptr(),
                                                  part.data_ptr(), B, Cc, H, W, k, stride, ph // 2, pw // 2, Ho, Wo,
                                                  _stream()), "ocv_depthwise_conv_nhwc_sum_fwd")
    with timed("se_gate"):
        check(lib.ocv_se_gate_partials_fwd(part.data_ptr(), tiles, Ho * Wo, w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(),
                                           b2.data_ptr(), gate.data_ptr(), hid.data_ptr(), B, Cc, R, _stream()),
              "ocv_se_gate_partials_fwd")
    return out, gate


def depthwise_se_gate_weights(x: torch.Tensor, weight_kkc: torch.Tensor, bias: Optional[torch.Tensor], k: int, stride: int,
                              w1: torch.Tensor, b1: torch.Tensor, w2t: torch.Tensor, b2: torch.Tensor, w_proj: torch.Tensor,
                              want_gate: bool = False):
    """silu(depthwise k x k (TF 'SAME') + bias) of a channels_last tensor written ONCE, in the hl32 split layout, and the
    squeeze-excite gate of that output FOLDED INTO the project weight per image: returns (y SplitAct [B, C, Ho, Wo],
    PerImageSplitWeight of w_proj [N, C] * diag(gate[b])) (+ the gate [B, C] with ``want_gate``).  Three launches."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    _req(weight_kkc, "weight")
    B, Cc, H, W = x.shape
    if weight_kkc.shape != (k * k, Cc) or Cc % 32 != 0:
        raise ValueError(f"depthwise_se_gate_weights: weight {tuple(weight_kkc.shape)} does not match k={k}, C={Cc} (C % 32 == 0)")
    for n, t in (("bias", bias), ("w1", w1), ("b1", b1), ("w2t", w2t), ("b2", b2), ("w_proj", w_proj)):
        if t is not None:
            _req(t, n)
    R = w1.shape[0]
    N = w_proj.shape[0]
    if w1.shape != (R, Cc) or w2t.shape != (R, Cc) or b1.numel() != R or b2.numel() != Cc or tuple(w_proj.shape) != (N, Cc):
        raise ValueError("depthwise_se_gate_weights: parameter shape mismatch")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    tiles = lib.ocv_depthwise_sum_tiles(B, Cc, Ho, Wo, k, stride)
    if tiles <= 0:
        raise ValueError("depthwise_se_gate_weights: unsupported shape")
    ys = SplitAct.empty(B, Cc, Ho, Wo, x.device)
    part = workspace(B * tiles * Cc * 4, x.device, "dw_part")
    hid = workspace(B * R * 4, x.device, "se_hidden")
    img_elems = int(lib.ocv_pointwise_packed_weight_elems(Cc, N))
    wpk = torch.empty(B * img_elems, dtype=torch.bfloat16, device=x.device)
    gate = torch.empty(B, Cc, dtype=torch.float32, device=x.device) if want_gate else None
    with timed(f"depthwise|{B},{H},{W},{Cc},k{k}s{stride}"):
        check(lib.ocv_depthwise_conv_nhwc_sum_hl_fwd(x.data_ptr(), weight_kkc.data_ptr(), _ptr(bias), None, ys.hl.data_ptr(),
                                                     part.data_ptr(), B, Cc, H, W, k, stride, ph // 2, pw // 2, Ho, Wo,
                                                     _stream()), "ocv_depthwise_conv_nhwc_sum_hl_fwd")
    with timed("se_gate_weights"):
        check(lib.ocv_se_gate_weights_fwd(part.data_ptr(), tiles, Ho * Wo, w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(),
                                          b2.data_ptr(), w_proj.data_ptr(), wpk.data_ptr(), img_elems, _ptr(gate), hid.data_ptr(),
                                          B, Cc, R, N, _stream()), "ocv_se_gate_weights_fwd")
    wg = PerImageSplitWeight(wpk, N, Cc, img_elems, B)
    return (ys, wg, gate) if want_gate else (ys, wg)


def expand_depthwise_fusable(cin: int, weight, k: int = 3) -> bool:
    """Whether ``expand_depthwise_se_gate`` is the faster plan for an MBConv block: packed split-bf16 expand weight,
    24 <= Cin <= 64 and a 3 x 3 depthwise kernel (measured at B = 16: 40 -> 240 at 120 x 160 181 us fused against 123 + 150
    as two launches, 24 -> 144 stride 2 at 240 x 320 239 against 202 + 203; the 5 x 5 blocks -- 25 FMAs per output and
    1.7x halo recompute of the expand SiLU -- are VALU-bound fused and stay on the two-launch path: 64 -> 384 at 60 x 80
    219 us fused against 42 + 71)."""
    return isinstance(weight, SplitWeight) and 24 <= cin <= 64 and cin % 8 == 0 and k == 3


def expand_depthwise_se_gate(x: torch.Tensor, w_expand: "SplitWeight", b_expand: Optional[torch.Tensor], weight_kkc: torch.Tensor,
                             bias: Optional[torch.Tensor], k: int, stride: int, w1: torch.Tensor, b1: torch.Tensor,
                             w2t: torch.Tensor, b2: torch.Tensor):
    """silu(depthwise(silu(x @ We^T + be)) + bd) of a channels_last tensor without materialising the expanded tensor, AND
    the squeeze-excite gate of that output: returns (y [B, mid, Ho, Wo] channels_last, gate [B, mid])."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    B, Cin, H, W = x.shape
    if not isinstance(w_expand, SplitWeight) or w_expand.cin != Cin:
        raise ValueError("expand_depthwise_se_gate: expand weight must be a SplitWeight matching x's channels")
    _req(w_expand.packed, "w_expand.packed", torch.bfloat16)
    mid = w_expand.cout
    _req(weight_kkc, "weight")
    if weight_kkc.shape != (k * k, mid):
        raise ValueError(f"expand_depthwise_se_gate: depthwise weight {tuple(weight_kkc.shape)} does not match k={k}, C={mid}")
    for n, t in (("b_expand", b_expand), ("bias", bias), ("w1", w1), ("b1", b1), ("w2t", w2t), ("b2", b2)):
        if t is not None:
            _req(t, n)
    R = w1.shape[0]
    if w1.shape != (R, mid) or w2t.shape != (R, mid) or b1.numel() != R or b2.numel() != mid:
        raise ValueError("expand_depthwise_se_gate: squeeze-excite parameter shape mismatch")
    if (b_expand is not None and b_expand.numel() != mid) or (bias is not None and bias.numel() != mid):
        raise ValueError("expand_depthwise_se_gate: bias size mismatch")
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + k - H, 0)
    pw = max((Wo - 1) * stride + k - W, 0)
    tiles = lib.ocv_mbconv_expand_dw_tiles(Ho, Wo, k, stride)
    if tiles <= 0:
        raise ValueError("expand_depthwise_se_gate: unsupported shape")
    out = torch.empty(B, mid, Ho, Wo, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    part = workspace(B * tiles * mid * 4, x.device, "dw_part")
    gate = torch.empty(B, mid, dtype=torch.float32, device=x.device)
    hid = workspace(B * R * 4, x.device, "se_hidden")
    with timed(f"expand_dw|{B},{H},{W},{Cin},{mid},k{k}s{stride}"):
        check(lib.ocv_mbconv_expand_dw_fwd(x.data_ptr(), w_expand.packed.data_ptr(), _ptr(b_expand), weight_kkc.data_ptr(),
                                           _ptr(bias), out.data_ptr(), part.data_ptr(), B, H, W, Cin, mid, k, stride,
                                           ph // 2, pw // 2, Ho, Wo, _stream()), "ocv_mbconv_expand_dw_fwd")
    with timed("se_gate"):
        check(lib.ocv_se_gate_partials_fwd(part.data_ptr(), tiles, Ho * Wo, w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(),
                                           b2.data_ptr(), gate.data_ptr(), hid.data_ptr(), B, mid, R, _stream()),
              "ocv_se_gate_partials_fwd")
    return out, gate


def channel_mean_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B, C] = mean over H, W of a channels_last [B, C, H, W] tensor."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    B, Cc, H, W = x.shape
    nb = lib.ocv_channel_mean_workspace_bytes(B, Cc, H * W)
    if nb == 0:
        raise ValueError("channel_mean_nhwc: unsupported shape")
    ws = workspace(nb, x.device, "mean")
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    check(lib.ocv_channel_mean_nhwc_fwd(x.data_ptr(), out.data_ptr(), B, Cc, H * W, ws.data_ptr(), ws.numel(), _stream()),
          "ocv_channel_mean_nhwc_fwd")
    return out


def se_gate(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2t: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """Squeeze-excite gate [B, C] of a channels_last activation: sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2).
    w1 [R, C]; w2t [R, C] = W2 transposed (coalesced over channels)."""
    lib = _lib.load()
    m = channel_mean_nhwc(x)
    B, Cc = m.shape
    for n, t in (("w1", w1), ("b1", b1), ("w2t", w2t), ("b2", b2)):
        _req(t, n)
    R = w1.shape[0]
    if w1.shape != (R, Cc) or w2t.shape != (R, Cc) or b1.numel() != R or b2.numel() != Cc:
        raise ValueError("se_gate: parameter shape mismatch")
    gate = torch.empty_like(m)
    hid = workspace(B * R * 4, x.device, "se_hidden")
    check(lib.ocv_se_gate_fwd(m.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), gate.data_ptr(),
                              hid.data_ptr(), B, Cc, R, _stream()), "ocv_se_gate_fwd")
    return gate


# ---------------------------------------------------------------------------
# split-bf16 activations between our convolutions
# ---------------------------------------------------------------------------
class SplitAct:
    """An activation of logical shape [B, C, H, W] held in the "hl32" two-term split layout of include/objcavit_hip.h: one
    2-byte buffer ``hl`` [B, H, W, 2 * Cp] (Cp = C rounded up to 32; dtype bfloat16 or float16 = the element type of the pairs)
    with, per pixel and per 32-channel block, the 32 hi = t(v) values followed by the 32 lo = t(v - hi) values; pad channels are zero.  Produced by
    ``upsample_concat_split`` / ``conv_nhwc_split(..., out_split=True)``, consumed by ``conv_nhwc_split`` with no
    per-tap conversion work."""
    __slots__ = ("hl", "C")

    def __init__(self, hl: torch.Tensor, C: int):
        self.hl, self.C = hl, int(C)

    @staticmethod
    def empty(B: int, C: int, H: int, W: int, device, f16: bool = False) -> "SplitAct":
        Cp = (C + 31) // 32 * 32
        n = int(_lib.load().ocv_split_act_elems(B, H, W, C))
        if n != B * H * W * 2 * Cp:
            raise ValueError(f"SplitAct: bad sizes {(B, C, H, W)}")
        return SplitAct(torch.empty(B, H, W, 2 * Cp, dtype=torch.float16 if f16 else torch.bfloat16, device=device), C)

    @property
    def f16(self) -> bool:
        """Whether the pairs are fp16 (2^-22 products, +-65504) rather than bf16 (2^-17, fp32's range)."""
        return self.hl.dtype == torch.float16

    @property
    def shape(self):
        B, H, W, _ = self.hl.shape
        return torch.Size((B, self.C, H, W))

    def parts(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(hi, lo) as [B, C, H, W] bf16 views of the buffer (pad channels dropped)."""
        B, H, W, c2 = self.hl.shape
        v = self.hl.view(B, H, W, c2 // 64, 2, 32)
        hi = v[..., 0, :].reshape(B, H, W, c2 // 2)[..., :self.C].permute(0, 3, 1, 2)
        lo = v[..., 1, :].reshape(B, H, W, c2 // 2)[..., :self.C].permute(0, 3, 1, 2)
        return hi, lo

    @property
    def hi(self) -> torch.Tensor:
        return self.parts()[0]

    @property
    def lo(self) -> torch.Tensor:
        return self.parts()[1]

    def float(self) -> torch.Tensor:
        hi, lo = self.parts()
        return hi.float() + lo.float()


def upsample_concat_split(x: torch.Tensor, skip: Optional[torch.Tensor], size: Tuple[int, int], f16: bool = False) -> SplitAct:
    """split(cat([bilinear_resize(x, size, align_corners=True), skip], dim=1)); x / skip channels_last fp32; the split as bf16
    pairs, or fp16 pairs with ``f16``."""
    lib = _lib.load()
    x = _nhwc(x, "x")
    B, C1, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    C2 = 0
    if skip is not None:
        skip = _nhwc(skip, "skip")
        if skip.shape[0] != B or tuple(skip.shape[2:]) != (H, W):
            raise ValueError("upsample_concat_split: skip must be [B, C2, H, W] at the target size")
        C2 = skip.shape[1]
    out = SplitAct.empty(B, C1 + C2, H, W, x.device, f16=f16)
    with timed("upsample_concat_split"):
        check(lib.ocv_upsample_concat_split_x_fwd(x.data_ptr(), h, w, C1, _ptr(skip), C2, out.hl.data_ptr(), int(bool(f16)), B, H, W,
                                                  _stream()), "ocv_upsample_concat_split_fwd")
    _note_range(f"split|{B},{H},{W},{C1 + C2}", out)
    return out


def conv_nhwc_split(x: SplitAct, w_hi: torch.Tensor, w_lo: torch.Tensor, bias: Optional[torch.Tensor], ksize: int,
                    act: int = ACT_NONE, out_fp32: bool = True, out_split: bool = False, oscale: Optional[torch.Tensor] = None):
    """Two-term-split convolution on a pre-split input; the weights' element type must be the input's (bf16 pairs, or fp16 pairs
    with their per-output-channel ``oscale``: prep_conv_weight(f16=True)), the split output has it too.
    Returns fp32 tensor, SplitAct, or (fp32, SplitAct)."""
    lib = _lib.load()
    if not (out_fp32 or out_split):
        raise ValueError("conv_nhwc_split: nothing to output")
    dt = x.hl.dtype
    _req(x.hl, "x.hl", dt)
    B, Cin, H, W = x.shape
    if x.hl.dim() != 4 or x.hl.shape[3] != 2 * ((Cin + 31) // 32 * 32):
        raise ValueError("conv_nhwc_split: x.hl must be [B, H, W, 2 * ceil32(C)]")
    for n, t in (("w_hi", w_hi), ("w_lo", w_lo)):
        _req(t, n, dt)
    taps, Cout, Cp = w_hi.shape
    if w_lo.shape != w_hi.shape or taps != ksize * ksize or Cp != (Cin + 31) // 32 * 32:
        raise ValueError(f"conv_nhwc_split: weights {tuple(w_hi.shape)} do not match {Cin} input channels, k={ksize}")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("conv_nhwc_split: bias size mismatch")
    if oscale is not None:
        _req(oscale, "oscale")
        if oscale.numel() != Cout:
            raise ValueError("conv_nhwc_split: oscale size mismatch")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.hl.device, memory_format=torch.channels_last) if out_fp32 else None
    ys = SplitAct.empty(B, Cout, H, W, x.hl.device, f16=x.f16) if out_split else None
    nws = int(lib.ocv_conv_nhwc_split_workspace_bytes(B, H, W, Cin, Cout, ksize))      # split-K partial sums (most shapes: 0)
    ws = workspace(nws, x.hl.device, "conv_splitk") if nws else None
    ptrs = (x.hl.data_ptr(), Cin, w_hi.data_ptr(), w_lo.data_ptr(), _ptr(oscale), int(x.f16), _ptr(bias), None, _ptr(y),
            ys.hl.data_ptr() if out_split else None, B, H, W, Cout, ksize, act, _ptr(ws), nws)
    keep = (x, w_hi, w_lo, oscale, bias, y, ys, ws)      # an eager island re-issues this launch on every replay
    launch(f"conv{ksize}x{ksize}|{B},{H},{W},{Cin},{Cout}",
           lambda: (keep, check(lib.ocv_conv_nhwc_split_x_fwd(*ptrs, _stream()), "ocv_conv_nhwc_split_x_fwd"))[1])
    _note_range(f"conv{ksize}x{ksize}|{B},{H},{W},{Cin},{Cout}", ys)
    if out_fp32 and out_split:
        return y, ys
    return y if out_fp32 else ys


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
