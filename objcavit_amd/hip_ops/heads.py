"""hip_ops: patch embedding, pixel-wise dot product, fused bin head, bin edges, ragged object lists, positional-embedding samplers
(csrc/patch_embed.hip, bin_head.hip, bin_edges.hip, objects_pad.hip, pos_sample.hip).
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
from .conv import *             # noqa: F401,F403  (SplitAct, fp32_map)


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
    products at the error of an fp32 FMA chain, three MFMAs per block, all 256 bins per workgroup) formed on TWO LEVELS -- every bin
    coarsely (one MFMA per block), the full logits and the softmax arithmetic only for the 32-bin tiles that hold a bin within
    e^-24 of a pixel's largest (csrc/bin_head.hip: OCV_BINHEAD=h2, the default; h2dense = every tile in full, round 4's kernel),
    as a THREE-term bf16 split (OCV_BINHEAD=split3: six MFMAs, two bin halves + a merge launch; fp32's RANGE -- also what a
    forward inside ``bf16_pairs()``, the range guard's fallback, takes), or on the exact fp32 MFMA kernel with ``exact=True`` /
    OCV_BINHEAD=exact."""
    lib = _lib.load()
    mode = os.environ.get("OCV_BINHEAD", "h2")              # read per call
    if mode not in ("h2", "h2dense", "split3", "exact"):
        raise ValueError(f"OCV_BINHEAD={mode!r}: expected 'h2' (default), 'h2dense', 'split3' or 'exact'")
    if mode in ("h2", "h2dense") and _TLS.bf16_pairs:
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
    route = 0 if not cl else (1 if exact else ({"h2": 4, "h2dense": 3}.get(mode, 2)))          # include/objcavit_hip.h: ocv_bin_head_folded_fwd
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


def regressor_bins(head: torch.Tensor, w1, b1, w2, b2, w3, b3, norm: str, min_depth: float, max_depth: float,
                   leaky_slope: float = 0.01) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The bin regressor (Linear + LeakyReLU, Linear + LeakyReLU, Linear) on one row per image + ``bin_edges`` in ONE launch.
    head [B, E]: rows may be strided (``tokens[:, 0, :]``), each row contiguous.  -> (bin_widths_normed, bin_edges, centers)."""
    lib = _lib.load()
    if norm not in BINNORM:
        raise ValueError("regressor_bins: norm must be one of " + ", ".join(BINNORM))
    _req(head, "head", contiguous=False)
    if head.dim() != 2 or head.stride(1) != 1:
        raise ValueError("regressor_bins: head must be [B, E] with contiguous rows")
    for nme, t in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("w3", w3), ("b3", b3)):
        _req(t, nme)
    B, E = head.shape
    H1, H2, n = w1.shape[0], w2.shape[0], w3.shape[0]
    if w1.shape != (H1, E) or w2.shape != (H2, H1) or w3.shape != (n, H2) or b1.numel() != H1 or b2.numel() != H2 or b3.numel() != n:
        raise ValueError("regressor_bins: parameter shape mismatch")
    w = torch.empty(B, n, dtype=torch.float32, device=head.device)
    e = torch.empty(B, n + 1, dtype=torch.float32, device=head.device)
    c = torch.empty_like(w)
    with timed("regressor_bins"):
        check(lib.ocv_regressor_bins_fwd(head.data_ptr(), head.stride(0), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                         w3.data_ptr(), b3.data_ptr(), E, H1, H2, n, float(leaky_slope), BINNORM[norm], float(min_depth),
                                         float(max_depth), w.data_ptr(), e.data_ptr(), c.data_ptr(), B, _stream()), "ocv_regressor_bins_fwd")
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]        # (private helpers included: the facade re-exports every name)
