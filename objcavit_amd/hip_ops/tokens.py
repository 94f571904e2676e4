"""hip_ops: linear / LayerNorm / attention / multi-head attention / transformer encoder layers (csrc/linear.hip, attention.hip,
token_split3.hip, token_h2.hip, xattn_h2.hip).
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
from ._core import *            # noqa: F401,F403


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


_ATTN_FORMS = {"h2": 0, "fp32": 1}
_attn_form_set = [None]


def attention_form_sync() -> None:
    """OCV_ATTN_FORM = h2 (default: the self-attention core on two-term fp16 products) | fp32 (exact-fp32 MFMA, the A/B numerics route)
    handed to the library (ocv_attention_set_dispatch) whenever it changed: the C side reads no environment.  Called by every
    function below that reaches the attention core."""
    form = os.environ.get("OCV_ATTN_FORM", "h2")
    if form != _attn_form_set[0]:
        if form not in _ATTN_FORMS:
            raise ValueError(f"OCV_ATTN_FORM={form!r}: expected 'h2' (default) or 'fp32'")
        check(_lib.load().ocv_attention_set_dispatch(_ATTN_FORMS[form]), "ocv_attention_set_dispatch")
        _attn_form_set[0] = form


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_padding_mask: Optional[torch.Tensor],
                   n_heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(32) + mask) v per head; q [B,Sq,E], k / v [B,Sk,E] (last-dim-contiguous views allowed)."""
    lib = _lib.load()
    attention_form_sync()
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
    attention_form_sync()
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
    attention_form_sync()
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
    attention_form_sync()
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]        # (private helpers included: the facade re-exports every name)
