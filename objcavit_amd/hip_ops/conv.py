"""hip_ops: the decoder's / heads' convolutions -- split activations (hl32), two-term split implicit GEMM, Winograd F(4x4, 3x3), the
low-resolution tap form, resize + concat + split, exact-fp32 route (csrc/conv_igemm.hip, tap_interp.hip, upsample.hip, conv_exact.hip).
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
# split-bf16 implicit-GEMM convolution on NHWC activations
# ---------------------------------------------------------------------------
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


def packed_taps_pay(Cin: int) -> bool:
    """Whether a 3x3 convolution over ``Cin`` channels runs as PACKED TAPS (ocv_conv3x3_split_packed_taps_fwd: the nine taps' real
    8-channel granules laid end to end along K) rather than tap-major with every tap padded to a multiple of 32 channels: where
    that saves at least a sixth of the K steps -- 24 channels (7 steps instead of 9), 40 (12 instead of 18), 8 / 16 (3 / 5 instead
    of 9); 64, 128, 176 ... do not (the decoder's other skip parts, every full-width convolution)."""
    if Cin < 8 or Cin % 8 != 0:
        return False
    packed = (9 * (Cin // 8) + 3) // 4
    dense = 9 * ((Cin + 31) // 32)
    return 6 * packed <= 5 * dense


def prep_conv_weight_packed_taps(weight: torch.Tensor, f16: bool = False):
    """[Cout, Cin, 3, 3] fp32 (Cin % 8 == 0) -> the two-term split of ONE [1, Cout, Kp] matrix in packed-tap order (include/
    objcavit_hip.h, ocv_conv3x3_split_packed_taps_fwd): column 8 ((Cin / 8) t + g) + e = weight[:, 8 g + e] of tap t = 3 ky + kx,
    zeros behind the ninth tap up to Kp = ocv_conv3x3_packed_taps_k(Cin).  Pairs and per-row scales as ``prep_conv_weight``
    (f16: (hi, lo, oscale); else (hi, lo))."""
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or Cin % 8 != 0:
        raise ValueError("prep_conv_weight_packed_taps: needs a 3x3 kernel over a multiple of 8 input channels")
    Kp = int(_lib.load().ocv_conv3x3_packed_taps_k(Cin))
    w = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)            # [co][tap][c]: tap-major granules of 8
    w = torch.nn.functional.pad(w, (0, Kp - 9 * Cin)).reshape(1, Cout, Kp)
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


def conv3x3_split_packed_taps(x: SplitAct, w_hi: torch.Tensor, w_lo: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE,
                              out_fp32: bool = True, out_split: bool = False, oscale: Optional[torch.Tensor] = None):
    """3x3 convolution (stride 1, padding 1) of a pre-split activation on packed-tap weights (``prep_conv_weight_packed_taps``):
    ocv_conv3x3_split_packed_taps_fwd.  Returns fp32 tensor, SplitAct, or (fp32, SplitAct) like ``conv_nhwc_split``."""
    lib = _lib.load()
    if not (out_fp32 or out_split):
        raise ValueError("conv3x3_split_packed_taps: nothing to output")
    dt = x.hl.dtype
    _req(x.hl, "x.hl", dt)
    B, Cin, H, W = x.shape
    if x.hl.dim() != 4 or x.hl.shape[3] != 2 * ((Cin + 31) // 32 * 32):
        raise ValueError("conv3x3_split_packed_taps: x.hl must be [B, H, W, 2 * ceil32(C)]")
    for n, t in (("w_hi", w_hi), ("w_lo", w_lo)):
        _req(t, n, dt)
    Kp = int(lib.ocv_conv3x3_packed_taps_k(Cin))
    if Cin % 8 != 0 or Kp == 0 or w_hi.dim() != 3 or w_hi.shape[0] != 1 or w_hi.shape[2] != Kp or w_lo.shape != w_hi.shape:
        raise ValueError(f"conv3x3_split_packed_taps: weights {tuple(w_hi.shape)} do not match {Cin} input channels (Kp = {Kp})")
    Cout = w_hi.shape[1]
    for n, t in (("bias", bias), ("oscale", oscale)):
        if t is not None:
            _req(t, n)
            if t.numel() != Cout:
                raise ValueError(f"conv3x3_split_packed_taps: {n} size mismatch")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.hl.device, memory_format=torch.channels_last) if out_fp32 else None
    ys = SplitAct.empty(B, Cout, H, W, x.hl.device, f16=x.f16) if out_split else None
    ptrs = (x.hl.data_ptr(), Cin, w_hi.data_ptr(), w_lo.data_ptr(), _ptr(oscale), int(x.f16), _ptr(bias), None, _ptr(y),
            ys.hl.data_ptr() if out_split else None, B, H, W, Cout, act)
    keep = (x, w_hi, w_lo, oscale, bias, y, ys)          # an eager island re-issues this launch on every replay
    launch(f"conv3x3p|{B},{H},{W},{Cin},{Cout}",
           lambda: (keep, check(lib.ocv_conv3x3_split_packed_taps_fwd(*ptrs, _stream()), "ocv_conv3x3_split_packed_taps_fwd"))[1])
    _note_range(f"conv3x3p|{B},{H},{W},{Cin},{Cout}", ys)
    if out_fp32 and out_split:
        return y, ys
    return y if out_fp32 else ys


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


__all__ = [_n for _n in dir() if not _n.startswith("__")]        # (private helpers included: the facade re-exports every name)
