"""hip_ops: the EfficientNet encoder's NHWC blocks -- stem, 1x1 (split / exact / pre-split), depthwise (+ squeeze-excite pooling and
gate), fused expand + depthwise -- and the validation metrics (csrc/stem.hip, pointwise_split.hip, pointwise_hl.hip, depthwise_se.hip,
mbconv_fused.hip, encoder_nhwc.hip, metrics.hip).
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
from .conv import *             # noqa: F401,F403  (SplitAct, _nhwc)


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


__all__ = [_n for _n in dir() if not _n.startswith("__")]        # (private helpers included: the facade re-exports every name)
