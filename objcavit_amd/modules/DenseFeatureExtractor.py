"""Dense feature extractor (EfficientNet encoder + UNet decoder), drop-in for
the reference's ``modules/DenseFeatureExtractor.py`` (same class names,
constructor arguments, attribute names and state_dict keys).

GPU inference plan (eval + no_grad), all on
libobjcavit_hip.so: every conv+BatchNorm pair of the decoder is folded once
into a single biased convolution (cached per parameter version); resize +
concat + fp32 -> split-bf16 is one kernel (ocv_upsample_concat_split_fwd) and
the 3x3 convolutions run on pre-split activations (ocv_conv_nhwc_split_ws_fwd);
the encoder keeps only the five activations the decoder reads instead of all
sixteen.  OCV_CONV=exact routes the 3x3 convolutions through the hand-written
exact-fp32 implicit GEMM instead (csrc/conv_exact.hip; A/B numerics), which
also takes the channel counts the split kernel does not.  PyTorch convolutions
run only in training / on the CPU, where the plain module graph is what the
golden generator wraps.
"""
from __future__ import annotations

import logging
import sys
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops
from .efficientnet import tf_efficientnet_b5_ap


def split_bf16_convs_enabled() -> bool:
    """Dense convolutions of the decoder / heads on the GPU inference path: the hand-written split-bf16 implicit GEMM
    (default) or, with OCV_CONV=exact, the hand-written EXACT-fp32 implicit GEMM (csrc/conv_exact.hip, 5x slower: the
    A/B numerics route -- round 1 used MIOpen for it).  Either way nothing but libobjcavit_hip.so computes them."""
    import os
    mode = os.environ.get("OCV_CONV", "split_bf16")
    if mode not in ("split_bf16", "exact"):
        raise ValueError(f"OCV_CONV={mode!r}: expected 'split_bf16' (default) or 'exact'")
    return mode != "exact"


class Fp16Unsafe(Exception):
    """A convolution's (possibly composed) weight does not fit fp16 pairs (hip_ops.fp16_weight_safe): the decoder then runs its
    whole split pipeline on bf16 pairs instead and says so in hip_ops.ROUTE_REPORT."""


class SplitConv3x3:
    """Inference-time plan for one 3x3 (or 1x1) convolution [+ folded BatchNorm]: weights split into two 2-byte terms once per
    parameter version -- bf16 pairs, or fp16 pairs scaled per output channel (round 4), whichever the pre-split input it is
    handed holds -- then ``ocv_conv_nhwc_split_x_fwd`` / the Winograd forms."""

    def __init__(self, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None):
        self.conv, self.bn = conv, bn
        self._key = None
        self._prep = None
        self._wino = None
        self._w_folded = None
        self._w_exact = None
        self._w_up = None
        self._w_up_all = {}              # every arrangement built for this weight version: the fp16-pair one and its bf16 fallback
        self._prep16 = None              # stay alive side by side (two captured graphs may hold their addresses)

    def prep_for(self, f16: bool):
        """(w_hi, w_lo, bias, oscale) in the element type of the input: bf16 pairs (oscale None) or fp16 pairs."""
        self._ensure_prepared()
        if not f16:
            return self._prep[0], self._prep[1], self._prep[2], None
        if self._prep16 is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
            if not hip_ops.fp16_weight_safe(self._w_folded.flatten(1)):
                raise Fp16Unsafe(f"{self.conv}: column spread beyond 2^17")
            self._prep16 = hip_ops.prep_conv_weight(self._w_folded, f16=True)
        return self._prep16[0], self._prep16[1], self._prep[2], self._prep16[2]

    def usable(self, c1: int, c2: int = 0) -> bool:
        k = self.conv.kernel_size
        return (split_bf16_convs_enabled() and k[0] == k[1] and k[0] in (1, 3) and self.conv.stride == (1, 1)
                and self.conv.padding == (k[0] // 2, k[0] // 2) and self.conv.groups == 1 and self.conv.dilation == (1, 1)
                and c1 % 4 == 0 and c2 % 4 == 0 and (c2 == 0 or c1 % 32 == 0))

    def __call__(self, x1, x2=None, act=hip_ops.ACT_NONE):
        self._ensure_prepared()
        hi, lo, b = self._prep
        return hip_ops.conv_nhwc(x1, x2, hi, lo, b, self.conv.kernel_size[0], act)

    def exact(self, x1, x2=None, act=hip_ops.ACT_NONE):
        """The same convolution (any odd kernel size <= 7, "same" padding, stride 1, any channel counts) on the exact-fp32
        kernel: OCV_CONV=exact, and the shapes the split-bf16 kernels do not take."""
        c = self.conv
        k = c.kernel_size[0]
        if not (c.kernel_size == (k, k) and k % 2 == 1 and k <= 7 and c.stride == (1, 1) and c.padding == (k // 2, k // 2)
                and c.groups == 1 and c.dilation == (1, 1)):
            from .._lib import HipLibraryError
            raise HipLibraryError(f"no hand-written kernel for this convolution: {c}")
        self._ensure_prepared()
        if self._w_exact is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
            w = self._w_folded
            self._w_exact = w.float().permute(2, 3, 0, 1).reshape(k * k, w.shape[0], w.shape[1]).contiguous()
        return hip_ops.conv_nhwc_exact(x1, x2, self._w_exact, self._prep[2], k, act)

    def _ensure_prepared(self):
        ps = [self.conv.weight] + ([self.conv.bias] if self.conv.bias is not None else [])
        if self.bn is not None:
            ps += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
            with torch.no_grad():
                if self.bn is not None:
                    w, b = _fold_conv_bn(self.conv, self.bn)
                else:
                    w, b = self.conv.weight, self.conv.bias
                k = w.shape[-1]
                hi, lo = hip_ops.prep_conv_weight(w) if k in (1, 3) else (None, None)
                self._prep = (hi, lo, None if b is None else b.detach().float().contiguous())
                self._wino = None                                      # transformed / tap-major weights: built on first use
                self._w_exact = None
                self._w_up = None
                self._w_up_all = {}
                self._prep16 = None
                self._w_folded = w.detach()
            self._key = key

    def upconv_weights(self, c1: int, compose=None, f16: bool = False):
        """The weight of a convolution over cat([up(x), skip]) re-arranged for the low-resolution form
        (csrc/tap_interp.hip): the up-sampled half as ONE 1x1 weight with the nine taps stacked tap-major
        ([9 Cout, C1]: row t Cout + co, t = 3 ky + kx), the skip half as an ordinary 3x3 weight (None when C2 = 0).
        ``compose`` = (key, fn): x itself is a bias-only-affine image of an earlier tensor, x = P x0 + pb with
        (P [C1, K], pb [C1] or None) = fn() in float64 -- the tap-stacked weight is then composed with P in float64
        ([9 Cout, K], applied to x0 directly) and ``border`` is the tap-stacked image of pb ([9 Cout] fp32, what z holds
        where x = pb), None without ``compose``.  ``f16``: the two-term splits as fp16 pairs with per-row scales (a_osc / s_osc)
        instead of bf16 pairs.  -> dict(a_hi, a_lo, a_osc, s_hi, s_lo, s_osc, bias, border)."""
        self._ensure_prepared()
        ckey = None if compose is None else compose[0]
        if (self._w_up is None or self._w_up[0] != (c1, ckey, bool(f16))) and (c1, ckey, bool(f16)) in self._w_up_all:
            self._w_up = self._w_up_all[(c1, ckey, bool(f16))]
        if self._w_up is None or self._w_up[0] != (c1, ckey, bool(f16)):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
            w = self._w_folded
            cout, cin = w.shape[0], w.shape[1]
            wa = w[:, :c1].permute(2, 3, 0, 1).reshape(9 * cout, c1)
            cvec = None
            if compose is not None:
                P, pb = compose[1]()
                wa64 = wa.double()
                cvec = (wa64 @ pb.double()).float().contiguous() if pb is not None else torch.zeros(9 * cout, device=w.device)
                wa = (wa64 @ P.double()).float()
            if f16 and not hip_ops.fp16_weight_safe(wa):
                raise Fp16Unsafe(f"{self.conv} (tap-stacked{', composed' if compose is not None else ''}): column spread beyond 2^17")
            pa = hip_ops.prep_conv_weight(wa.reshape(9 * cout, -1, 1, 1), f16=f16)
            d = dict(a_hi=pa[0], a_lo=pa[1], a_osc=pa[2] if f16 else None, s_hi=None, s_lo=None, s_osc=None, border=cvec)
            if 0 < cin - c1 <= 4:
                # a skip tensor of at most four channels (the IMAGE, do_final_upscale): exact-fp32 direct form, tap-major fp32
                # weight [9, C2, Cout] in the s_hi slot, s_lo None (hip_ops.conv3x3_few_channels)
                d["s_hi"] = w[:, c1:].permute(2, 3, 1, 0).reshape(9, cin - c1, cout).contiguous()
            elif cin > c1:
                ws = w[:, c1:].contiguous()
                if f16 and not hip_ops.fp16_weight_safe(ws.flatten(1)):
                    raise Fp16Unsafe(f"{self.conv} (skip part): column spread beyond 2^17")
                # few skip channels (24, 40: the two high-resolution stages): packed taps -- 7 / 12 K steps instead of 9 / 18
                d["s_packed"] = hip_ops.packed_taps_pay(cin - c1)
                ps = (hip_ops.prep_conv_weight_packed_taps if d["s_packed"] else hip_ops.prep_conv_weight)(ws, f16=f16)
                d.update(s_hi=ps[0], s_lo=ps[1], s_osc=ps[2] if f16 else None)
            self._w_up = self._w_up_all[(c1, ckey, bool(f16))] = ((c1, ckey, bool(f16)), d)
        return dict(self._w_up[1], bias=self._prep[2])

    def run_split(self, x: "hip_ops.SplitAct", act=hip_ops.ACT_NONE, out_fp32=True, out_split=False):
        """Same convolution on a pre-split activation (no per-tap fp32 -> 2-byte work in the kernel), in the element type of
        ``x`` (bf16 or fp16 pairs)."""
        hi, lo, b, osc = self.prep_for(x.f16)
        B, Cin, H, W = x.shape
        if self.conv.kernel_size[0] == 3 and hip_ops.winograd_pays(B, H, W, Cin, self.conv.out_channels):
            # the deep stages: Winograd F(4x4, 3x3), 4x fewer matrix-core operations; fp16 pairs inside whatever the input's
            # element type (every tile scaled by a power of two from its own maximum: fp32's range) -- csrc/conv_igemm.hip
            if self._wino is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("weight preparation during graph capture: run one eager warm-up call first")
                self._wino = (4,) + tuple(hip_ops.prep_winograd43_weight(self._w_folded))
            return hip_ops.conv3x3_winograd43_split(x, self._wino[1], self._wino[2], self._wino[3], b, act, out_fp32=out_fp32,
                                                    out_split=out_split, cscale=self._wino[4])
        return hip_ops.conv_nhwc_split(x, hi, lo, b, self.conv.kernel_size[0], act, out_fp32=out_fp32, out_split=out_split, oscale=osc)

# skip-connection table: encoder-name fragment -> (feature_select, skip channels 3..0)
# (reference modules/DenseFeatureExtractor.py:62-85)
_SKIPS = {
    "efficientnet-b5": ((4, 5, 6, 8, 11), (176, 64, 40, 24)),
    "efficientnet-b1": ((4, 5, 6, 8, 11), (112, 40, 24, 16)),
    "efficientnet-v2-s": ((2, 3, 4, 6, 9), (160, 64, 48, 24)),
    "efficientnet-v2-m": ((2, 3, 4, 6, 9), (176, 80, 48, 24)),
}


class Encoder(nn.Module):
    """Collects the activation after every child of the backbone, expanding the
    children of ``blocks`` / ``features`` (reference :11-27).  ``keep`` limits
    what is retained (indices into that list); None keeps everything."""

    def __init__(self, backend: nn.Module, keep: Optional[tuple] = None):
        super().__init__()
        self.original_model = backend
        self.keep = keep

    def forward(self, x: torch.Tensor, _defer_head: bool = False, _on_feature=None) -> List[Optional[torch.Tensor]]:
        """``_defer_head`` (DenseFeatureExtractor's GPU inference call only): the bias-free 1x1 ``conv_head``, when nothing
        but Identity modules follow it, is handed on un-applied (DeferredConv1x1) for the decoder to compose.
        ``_on_feature(idx, tensor)``: called the moment a KEPT activation exists (SkipPrepass: the decoder's skip-part convolutions
        start beside the rest of the encoder)."""
        feats: List[Optional[torch.Tensor]] = [x]
        cur = x

        def push(t):
            nonlocal cur
            cur = t
            idx = len(feats)
            kept = t is not None and (self.keep is None or idx in self.keep)
            feats.append(t if kept else None)
            if kept and _on_feature is not None and isinstance(t, torch.Tensor):
                _on_feature(idx, t)

        skip = self._fused_stem(x, push)
        names = list(self.original_model._modules)
        for i, (name, child) in enumerate(self.original_model._modules.items()):
            if name in skip:
                continue
            if name in ("blocks", "features"):
                for sub in child._modules.values():
                    push(sub(cur))
            elif (_defer_head and name == "conv_head" and isinstance(child, nn.Conv2d) and child.bias is None
                  and child.kernel_size == (1, 1) and child.stride == (1, 1) and child.padding == (0, 0) and child.groups == 1
                  and isinstance(cur, torch.Tensor) and cur.device.type == "cuda" and not child.training
                  and not torch.is_grad_enabled()
                  and all(isinstance(self.original_model._modules[n], nn.Identity) for n in names[i + 1:])):
                push(DeferredConv1x1(cur, child))
            else:
                push(child(cur))
        return feats


def _encoder_fused_stem(self, x, push):
    """conv_stem + bn1 + act1 as ONE launch (csrc/stem.hip) on the inference fast path, when the caller does not keep
    the two intermediate activations.  Returns the child names that were consumed."""
    m = self.original_model
    names = list(m._modules)[:3]
    if names != ["conv_stem", "bn1", "act1"] or self.keep is None or 1 in self.keep or 2 in self.keep:
        return ()
    conv, bn, act = m.conv_stem, m.bn1, m.act1
    if not (x.device.type == "cuda" and not torch.is_grad_enabled() and not m.training and isinstance(conv, nn.Conv2d)
            and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.SiLU) and conv.kernel_size == (3, 3)
            and conv.groups == 1 and conv.in_channels * 9 <= 32 and conv.out_channels <= 64
            and conv.padding == (0, 0) and conv.stride[0] == conv.stride[1] and x.dtype == torch.float32):
        return ()
    key = (x.device, conv.weight._version, conv.weight.data_ptr(), bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version)
    c = self.__dict__.get("_stem_cache")
    if c is None or c[0] != key:
        with torch.no_grad():
            w, b = _fold_conv_bn(conv, bn)
            c = (key, w.float().contiguous(), b.float().contiguous())
        self.__dict__["_stem_cache"] = c
    y = hip_ops.stem_conv_same(x.contiguous(), c[1], c[2], conv.stride[0], hip_ops.ACT_SILU)
    push(None)
    push(None)
    push(y)
    return ("conv_stem", "bn1", "act1")


Encoder._fused_stem = _encoder_fused_stem


def _fold_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * s.view(-1, 1, 1, 1)
    b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    return w, (b0 - bn.running_mean) * s + bn.bias


class _ShapeOnly:
    """Stand-in for a tensor where only ``.shape`` (and ``.device``) is consulted."""

    def __init__(self, *shape, device=None):
        self.shape = torch.Size(shape)
        self.device = device


class DeferredConv1x1:
    """``conv(x)`` for a bias-free 1x1 convolution, not yet applied: what the Encoder hands the Decoder for the backbone's
    conv_head on the GPU inference path, so the decoder can compose it into its first GEMM (Decoder._up1_affine).
    ``materialize()`` applies it."""

    def __init__(self, x: torch.Tensor, conv: nn.Conv2d):
        self.x, self.conv = x, conv
        self.shape = torch.Size((x.shape[0], conv.out_channels, x.shape[2], x.shape[3]))
        self.device, self.dtype = x.device, x.dtype

    def materialize(self) -> torch.Tensor:
        return self.conv(self.x)


class UpSampleWithSkip(nn.Module):
    """bilinear(align_corners=True) resize to the skip's size, concat, then
    2 x [conv3x3, BN, LeakyReLU(0.01)] (reference :30-47)."""

    def __init__(self, input_features: int, output_features: int):
        super().__init__()
        self._net = nn.Sequential(
            nn.Conv2d(input_features, output_features, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(output_features),
            nn.LeakyReLU(),
            nn.Conv2d(output_features, output_features, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(output_features),
            nn.LeakyReLU())
        self._folded = None
        self._split1 = SplitConv3x3(self._net[0], self._net[1])
        self._split2 = SplitConv3x3(self._net[3], self._net[4])

    def train(self, mode: bool = True):
        self._folded = None
        return super().train(mode)

    def _load_from_state_dict(self, *a, **kw):
        self._folded = None
        return super()._load_from_state_dict(*a, **kw)

    def lowres_ready(self, x, skip_features) -> bool:
        """Whether the first convolution can run in its low-resolution form (csrc/tap_interp.hip): the up-sampled
        channels fill whole 32-blocks, the resize is an up-sampling the kernel's staging covers.  OCV_UPCONV=direct in the
        environment keeps the resize + direct 3x3 convolution (A/B)."""
        import os
        if os.environ.get("OCV_UPCONV", "lowres") == "direct":
            return False
        c1, cout = x.shape[1], self._net[0].out_channels
        return (c1 % 32 == 0 and cout % 8 == 0 and x.shape[2] < skip_features.shape[2] and x.shape[3] < skip_features.shape[3]
                and hip_ops.tap_interp_supported(x.shape[2], x.shape[3], skip_features.shape[2], skip_features.shape[3], cout))

    def forward_split(self, x, skip_features, out_fp32=True, out_split=False, affine_of=None, f16=False, sk_pre=None):
        """GPU inference plan.  ``x``: the stage input, fp32 channels_last or already split (hip_ops.SplitAct).
        ``affine_of`` = (x0, key, fn, (h, w)): the stage input is NOT materialised -- it is the h x w grid whose interior is
        P x0 + pb (per pixel, (P, pb) = fn() in float64) and whose one-pixel border ring is pb (Decoder.conv2's padded
        1x1 convolution, optionally behind the backbone's bias-free conv_head): the tap-stacked weight is composed with P
        once per weight version and the 1x1 GEMM reads x0 (``x`` is then only consulted for its shape).
        ``f16``: the element type of every split tensor of the stage (fp16 pairs / bf16 pairs: hip_ops.conv_split_f16).
        First convolution, low-resolution form (default): conv3x3(cat(up(x), skip)) = sum over the nine taps of the
        bilinear interpolation of (W_tap x) -- formed once per LOW-resolution pixel by one 1x1 GEMM with 9 Cout columns,
        ~4x fewer matrix-core operations for the up-sampled channels -- + conv3x3 over the skip channels, combined,
        biased, activated and split by ocv_tap_interp_combine_fwd.  Otherwise: resize + concat + fp32->split in ONE
        pass, then the direct 3x3 convolution.  The second convolution hands the next stage fp32 and / or the split pair.
        ``sk_pre``: the skip part of the first convolution, already formed beside the encoder (``SkipPrepass``; ``skip_part`` below
        is the one statement of that launch)."""
        H, W = skip_features.shape[-2:]

        def as_split(t):
            if isinstance(t, hip_ops.SplitAct):
                return t if t.f16 == f16 else hip_ops.split_act(t.float().contiguous(memory_format=torch.channels_last), f16=f16)
            ride = getattr(t, "_ocv_hl", None)          # (an encoder block of the late stages may leave its split copy beside it)
            return ride if ride is not None and ride.f16 == f16 else hip_ops.split_act(t, f16=f16)

        if affine_of is not None or self.lowres_ready(x, skip_features):
            if affine_of is not None:
                x0, key, fn, _ = affine_of
                xs = as_split(x0)
                wt = self._split1.upconv_weights(x.shape[1], compose=(key, fn), f16=f16)
                z = hip_ops.conv_nhwc_split(xs, wt["a_hi"], wt["a_lo"], wt["border"], 1, hip_ops.ACT_NONE, out_fp32=True, oscale=wt["a_osc"])
            else:
                xs = as_split(x)
                wt = self._split1.upconv_weights(xs.shape[1], f16=f16)
                z = hip_ops.conv_nhwc_split(xs, wt["a_hi"], wt["a_lo"], None, 1, hip_ops.ACT_NONE, out_fp32=True, oscale=wt["a_osc"])
            sk = None
            if wt["s_hi"] is not None and wt["s_lo"] is None:
                # <= 4 skip channels: 27 - 36 multiply-adds per output on the vector units, the image read in place
                sk = hip_ops.conv3x3_few_channels(skip_features, wt["s_hi"])
            elif wt["s_hi"] is not None:
                c2 = skip_features.shape[1]
                if c2 % 4 and getattr(skip_features, "_ocv_hl", None) is None:      # 3-channel image: one zero channel more (the weight's pad columns are zero too)
                    skip_features = F.pad(skip_features, (0, 0, 0, 0, 0, 4 - c2 % 4))
                    sk = self.skip_part(skip_features, wt, f16)
                else:
                    # what a later forward may issue beside the encoder (SkipPrepass): this skip shape with this weight arrangement
                    self.__dict__["_skip_plan"] = (tuple(skip_features.shape), self._split1._w_up[0])
                    sk = sk_pre if sk_pre is not None else self.skip_part(skip_features, wt, f16)
            f = hip_ops.tap_interp_combine(z, sk, wt["bias"], (H, W), hip_ops.ACT_LEAKY_RELU, out_fp32=False, out_split=True,
                                           border=wt["border"], split_f16=f16)
        else:
            if isinstance(x, hip_ops.SplitAct):
                x = x.float()
            cat = hip_ops.upsample_concat_split(x, skip_features, (H, W), f16=f16)
            f = self._split1.run_split(cat, hip_ops.ACT_LEAKY_RELU, out_fp32=False, out_split=True)
        return self._split2.run_split(f, hip_ops.ACT_LEAKY_RELU, out_fp32=out_fp32, out_split=out_split)

    def skip_part(self, skip_features, wt, f16):
        """conv3x3 over the skip channels with the first convolution's weight columns of those channels: fp32, no bias, no
        activation (the tap interpolation adds it to the up-sampled channels' part)."""
        if isinstance(skip_features, hip_ops.SplitAct):
            xs = skip_features if skip_features.f16 == f16 else \
                hip_ops.split_act(skip_features.float().contiguous(memory_format=torch.channels_last), f16=f16)
        else:
            ride = getattr(skip_features, "_ocv_hl", None)      # (an encoder block of the late stages may leave its split copy beside it)
            xs = ride if ride is not None and ride.f16 == f16 else hip_ops.split_act(skip_features, f16=f16)
        if wt.get("s_packed"):
            return hip_ops.conv3x3_split_packed_taps(xs, wt["s_hi"], wt["s_lo"], None, hip_ops.ACT_NONE, out_fp32=True, oscale=wt["s_osc"])
        return hip_ops.conv_nhwc_split(xs, wt["s_hi"], wt["s_lo"], None, 3, hip_ops.ACT_NONE, out_fp32=True, oscale=wt["s_osc"])

    def split_ready(self, x, skip_features) -> bool:
        c1, c2 = x.shape[1], skip_features.shape[1]
        if not (not self.training and not torch.is_grad_enabled() and x.device.type == "cuda" and split_bf16_convs_enabled()
                and self._net[0].out_channels % 8 == 0):
            return False
        if self._split1.usable(c1, c2) and (c1 + c2) % 8 == 0:
            return True
        # skip channels the concatenating kernels do not take (the final_upscale stage's skip is the 3-channel IMAGE): in the
        # low-resolution form the skip part is a convolution of its own, over the skip tensor zero-padded to 4 channels
        return self._split1.usable(c1, 0) and self.lowres_ready(x, skip_features)

    def forward(self, x, skip_features):
        if self.split_ready(x, skip_features):
            try:
                return self.forward_split(x, skip_features, f16=hip_ops.conv_split_f16())
            except Fp16Unsafe as e:
                hip_ops.ROUTE_REPORT[f"UpSampleWithSkip({self._net[0].in_channels}->{self._net[0].out_channels})"] = f"bf16 pairs: {e}"
                return self.forward_split(x, skip_features, f16=False)
        up = F.interpolate(x, size=skip_features.shape[-2:], mode="bilinear", align_corners=True)
        if not self.training and not torch.is_grad_enabled() and up.device.type == "cuda":
            # hand-written convolutions on fp32 operands: the channel concat is virtual (two A-operand sources), BN is
            # folded, LeakyReLU fused -- split-bf16, or exact fp32 (OCV_CONV=exact / channel counts the split kernel
            # does not take); only the resize above is an ATen element-wise kernel here
            if self._split1.usable(up.shape[1], skip_features.shape[1]):
                f = self._split1(up, skip_features, hip_ops.ACT_LEAKY_RELU)
            else:
                f = self._split1.exact(up, skip_features, hip_ops.ACT_LEAKY_RELU)
            if self._split2.usable(f.shape[1]):
                return self._split2(f, None, hip_ops.ACT_LEAKY_RELU)
            return self._split2.exact(f, None, hip_ops.ACT_LEAKY_RELU)
        f = torch.cat([up, skip_features], dim=1)
        if self.training or torch.is_grad_enabled():
            return self._net(f)
        # CPU, eval: plain PyTorch with folded BatchNorm (what the golden generator wraps; never the GPU path)
        if self._folded is None or self._folded[0].device != f.device:
            with torch.no_grad():
                self._folded = (*_fold_conv_bn(self._net[0], self._net[1]), *_fold_conv_bn(self._net[3], self._net[4]))
        w1, b1, w2, b2 = self._folded
        f = F.leaky_relu(F.conv2d(f, w1, b1, padding=1), 0.01)
        return F.leaky_relu(F.conv2d(f, w2, b2, padding=1), 0.01)


class SkipPrepass:
    """The skip-part convolutions of the decoder's first convolutions (conv3x3 over 64 / 40 / 24 encoder channels at 60x80 ...
    240x320: short-K GEMMs, ~0.9 ms of a bs-16 step at a fifth of the matrix pipe) depend on the ENCODER's activations only: the three
    are issued on the side stream the moment the last of their activations exists (``Encoder.forward(_on_feature=...)``: behind
    stage 4 of 7), beside the encoder's late stages, whose launches leave most of the chip idle, and joined behind the encoder.
    Same launches on the same operands as in ``UpSampleWithSkip.forward_split`` (``skip_part``): bitwise the same result.  Only for
    stages whose previous forward took that route with the same shapes and weight arrangement (``_skip_plan``), and only once the
    decoder's element type is settled (never during the calibrating first call).
    ONE fork and ONE join, on the object branch's side stream (hip_ops.side_stream 0), shared with that branch (``extra``:
    GraphBins hands its object pre-pass over instead of forking it at the top of the forward).  Measured shapes
    (tools/history/ab_skip_overlap.sh, one batch at a time, captured forward, alternating runs on one box):
      * each convolution issued when its own activation appeared and joined where it was read (four forks, four joins between two
        streams): every replay 3.5 ms SLOWER at every batch size (bs 16: 20.6 vs 16.9 ms; bs 1: 6.9 vs 3.4), one replay crashed
        (hipGraphLaunch);
      * one fork behind stage 4 for the three, the object branch still forked at the top (= a side branch with TWO incoming edges
        from the main chain): 21.1 vs 16.6 ms, 8.2 vs 3.4 ms -- on this ROCm a graph branch may depend on the main chain ONCE;
      * this shape: **bs 16 984.6 -> 1000.0 img/s (+1.6 %), bs 1 296.4 -> 309.6 (+4.4 %)**; with two / one of the three
        convolutions (forks behind stage 3 / 2) 992 / 988 and 304 / 303 (tools/history/ab_skip_stages.sh)."""

    def __init__(self, decoder: "Decoder", device: torch.device, extra=None):
        self.extra, self.extra_result = extra, None      # further image-independent work for the same fork (GraphBins: the object branch)
        self.main = torch.cuda.current_stream(device)
        self.side = hip_ops.side_stream(device, 0)
        self.f16 = decoder.settled_f16()
        sel = decoder.feature_select
        stages = [(sel[0], decoder.up4), (sel[1], decoder.up3), (sel[2], decoder.up2)]
        self.stage_of = {} if self.f16 is None else dict(stages)
        self.trigger = stages[-1][0]                     # the last of the activations to appear
        self.pending = []                                # (stage, activation, weight dict)
        self.ready = {}                                  # id(stage) -> skip part
        self.forked = self.joined = False

    def on_feature(self, idx: int, t: torch.Tensor) -> None:
        up = self.stage_of.get(idx)
        if up is not None and t.dim() == 4:
            plan = up.__dict__.get("_skip_plan")
            up._split1._ensure_prepared()
            cache = up._split1._w_up
            t = t.contiguous(memory_format=torch.channels_last)
            if plan is not None and cache is not None and plan == (tuple(t.shape), cache[0]) and cache[0][2] == self.f16 \
                    and cache[1]["s_lo"] is not None:
                self.pending.append((up, t, cache[1]))
        if idx == self.trigger and (self.pending or self.extra is not None):
            self.side.wait_stream(self.main)
            self.forked = True
            with torch.cuda.stream(self.side), hip_ops.islands_suspended():
                for up, t, wt in self.pending:
                    t.record_stream(self.side)
                    self.ready[id(up)] = up.skip_part(t, wt, self.f16)
                if self.extra is not None:
                    self.extra_result = self.extra()
            self.pending = []

    def join(self) -> None:
        """The main stream waits for the side stream (once; GraphBins' own join of the object branch counts: ``joined``)."""
        if self.forked and not self.joined:
            self.main.wait_stream(self.side)
        self.joined = True

    def take(self, up, f16: bool):
        """The stage's skip part -- or None (the stage forms it itself)."""
        sk = self.ready.pop(id(up), None)
        if sk is None or f16 != self.f16:
            return None
        self.join()
        sk.record_stream(self.main)
        return sk


class Decoder(nn.Module):
    """UNet decoder (reference :50-118).  ``conv2`` is a 1x1 convolution with
    padding=1 -- inherited quirk, kept."""

    def __init__(self, num_features=2048, num_classes=1, bottleneck_features=2048, mode="features",
                 encoder_name=None, do_final_upscale=False):
        super().__init__()
        f = int(num_features)
        self.encoder_name = encoder_name
        self.conv2 = nn.Conv2d(bottleneck_features, f, kernel_size=1, stride=1, padding=1)
        for frag, (select, skips) in _SKIPS.items():
            if frag in (encoder_name or ""):
                self.feature_select = list(select)
                break
        else:
            sys.exit("Error: encoder name not recognised when building decoder.")
        self.up1 = UpSampleWithSkip(f // 1 + skips[0], f // 2)
        self.up2 = UpSampleWithSkip(f // 2 + skips[1], f // 4)
        self.up3 = UpSampleWithSkip(f // 4 + skips[2], f // 8)
        self.up4 = UpSampleWithSkip(f // 8 + skips[3], f // 16)
        self.final_upscale = UpSampleWithSkip(f // 16 + 3, f // 16) if do_final_upscale else None
        self.mode = mode if mode is not None else "features"
        self.conv3 = nn.Conv2d(f // 16, num_classes if self.mode == "features" else 1, kernel_size=3, stride=1, padding=1)
        self._split3 = SplitConv3x3(self.conv3)

    def _conv2_padded_1x1(self, b4):
        """conv2 is a 1x1 convolution with padding=1 (reference :57): the output grows by a border that only ever
        sees zero padding, i.e. equals the bias.  On the inference fast path the interior runs as the split-bf16
        pointwise kernel and the border is filled."""
        c = self.conv2
        if not (b4.device.type == "cuda" and not self.training and not torch.is_grad_enabled()
                and c.kernel_size == (1, 1) and c.padding == (1, 1) and c.stride == (1, 1) and c.groups == 1
                and c.in_channels % 8 == 0 and b4.dtype == torch.float32):
            if b4.device.type == "cuda" and not self.training and not torch.is_grad_enabled():
                from .._lib import HipLibraryError
                raise HipLibraryError(f"Decoder.conv2: no hand-written kernel for {c} on {tuple(b4.shape)}")
            return c(b4)
        key = (b4.device, c.weight._version, c.weight.data_ptr(), None if c.bias is None else c.bias._version)
        cache = self.__dict__.get("_conv2_cache")
        if cache is None or cache[0] != key:
            bias = None if c.bias is None else c.bias.detach().float().contiguous()
            cache = (key, hip_ops.pointwise_weight(c.weight), bias)
            self.__dict__["_conv2_cache"] = cache
        _, w, bias = cache
        inner = hip_ops.pointwise_nhwc(b4, w, bias, hip_ops.ACT_NONE)
        B, C, h, wd = inner.shape
        out = torch.empty(B, C, h + 2, wd + 2, dtype=inner.dtype, device=inner.device, memory_format=torch.channels_last)
        if bias is None:
            out.zero_()
        else:
            out.copy_(bias.view(1, C, 1, 1).expand_as(out))
        out[:, :, 1:-1, 1:-1] = inner
        return out

    def _up1_affine(self, b4, b3):
        """The first stage's input as an affine image of an EARLIER tensor, when that pays: conv2 (1x1, padding 1, no
        activation; reference :57,:105) -- and the backbone's bias-free conv_head in front of it, when the encoder deferred
        it -- are per-pixel linear maps directly in front of up1's low-resolution GEMM (also per-pixel linear), so the three
        compose into ONE weight [9 Cout, K] applied to the earlier tensor; conv2's border ring (= its bias) becomes a constant
        vector the interpolation kernel substitutes.  Returns (shape stand-in for conv2's output, affine_of) or None.
        OCV_UPCONV_FOLD=0 in the environment keeps the separate launches (A/B)."""
        import os
        c = self.conv2
        x0 = b4.x if isinstance(b4, DeferredConv1x1) else b4
        if os.environ.get("OCV_UPCONV_FOLD", "1") == "0" or not (
                x0.device.type == "cuda" and not self.training and not torch.is_grad_enabled()
                and c.kernel_size == (1, 1) and c.padding == (1, 1) and c.stride == (1, 1) and c.groups == 1
                and x0.dtype == torch.float32 and x0.shape[1] % 32 == 0):
            return None
        B, _, h, w = b4.shape
        shape = _ShapeOnly(B, c.out_channels, h + 2, w + 2, device=x0.device)
        if not (self.up1.split_ready(shape, b3) and self.up1.lowres_ready(shape, b3)):
            return None
        head = b4.conv if isinstance(b4, DeferredConv1x1) else None
        ps = [c.weight] + ([c.bias] if c.bias is not None else []) + ([head.weight] if head is not None else [])
        key = tuple((p.data_ptr(), p._version) for p in ps)

        def fn():
            P = c.weight.detach().flatten(1).double()
            if head is not None:
                P = P @ head.weight.detach().flatten(1).double()
            return P, None if c.bias is None else c.bias.detach()

        return shape, (x0, key, fn, (h + 2, w + 2))

    def settled_f16(self):
        """The element type of the split pipeline once it is decided for the current weights (True: fp16 pairs, False: bf16 pairs);
        None before the calibrating first call / after a weight update."""
        wkey = self._wkey()
        mode = self.__dict__.get("_f16_modes", {}).get(wkey)
        if mode is None or (mode[1] and len(mode) == 2):
            return None
        return bool(mode[1])

    def _wkey(self):
        """(requested element type, identity + version of every parameter / buffer): what a decision about the split pipeline's
        element type is valid for.  The fp16 route and the bf16 route (hip_ops.bf16_pairs: the range guard's fallback) of ONE weight
        version keep their own entries side by side."""
        return (hip_ops.conv_split_f16(),) + tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers())

    def _set_mode(self, wkey, mode):
        modes = {k: v for k, v in self.__dict__.get("_f16_modes", {}).items() if k[1:] == wkey[1:]}      # (stale weight versions dropped)
        modes[wkey] = mode
        self.__dict__["_f16_modes"] = modes
        return mode

    def forward(self, features, _split_only: bool = False, _skip_pre: Optional[SkipPrepass] = None):
        """``_split_only`` (GraphBins / AdaBins inference calls): the heads read the split copy of the result, so the last convolution
        writes only that and the returned tensor is a ``hip_ops.map_placeholder``.  ``_skip_pre``: skip-part convolutions already
        issued beside the encoder (``SkipPrepass``)."""
        if _skip_pre is not None:
            _skip_pre.join()                       # (before anything of the decoder: one fork, one join -- SkipPrepass)
        return self._forward(features, _split_only, _skip_pre)

    def _forward(self, features, _split_only, _skip_pre):
        b0, b1, b2, b3, b4 = (features[i] for i in self.feature_select)
        if b4.device.type == "cuda" and not self.training and not torch.is_grad_enabled():
            # inference on the GPU: the decoder runs in channels_last (NHWC), the layout of every kernel of the path
            cl = torch.channels_last
            b0, b1, b2, b3 = (t.contiguous(memory_format=cl) for t in (b0, b1, b2, b3))
            if isinstance(b4, DeferredConv1x1):
                b4.x = b4.x.contiguous(memory_format=cl)
            else:
                b4 = b4.contiguous(memory_format=cl)
        affine = self._up1_affine(b4, b3) if self._split3.usable(self.conv3.in_channels) and self.conv3.in_channels % 8 == 0 else None
        if affine is not None:
            x = affine[0]
        else:
            if isinstance(b4, DeferredConv1x1):
                b4 = b4.materialize()
            x = self._conv2_padded_1x1(b4)
        stages = ((self.up1, b3), (self.up2, b2), (self.up3, b1), (self.up4, b0))
        fin = self.final_upscale
        if (all(up.split_ready(x, skip) for up, skip in stages[:1])
                and self._split3.usable(self.conv3.in_channels) and self.conv3.in_channels % 8 == 0
                and (fin is None or fin.split_ready(_ShapeOnly(b0.shape[0], self.up4._net[3].out_channels, b0.shape[2], b0.shape[3], device=b0.device), features[0]))):
            # all-split pipeline: the last stage hands conv3 its input pre-split; conv3 returns the fp32 feature map
            # (patch embedding reads it) AND its split copy, which rides along for the heads' 3x3 convolution.  Element type of
            # every split tensor: fp16 pairs (round 4) unless a weight of the pipeline does not fit them -- then bf16 pairs for
            # the whole pipeline, decided once per weight version and REPORTED (hip_ops.ROUTE_REPORT), never silent.
            wkey = self._wkey()
            mode = self.__dict__.get("_f16_modes", {}).get(wkey)
            if mode is None:
                mode = self._set_mode(wkey, (wkey, hip_ops.conv_split_f16()))

            def pre_of(up, f16):
                return None if _skip_pre is None else _skip_pre.take(up, f16)

            def pipeline(f16):
                x_ = x
                for i, (up, skip) in enumerate(stages[:-1]):
                    # the next stage's low-resolution first convolution reads its input in split form, its resize kernel fp32
                    nxt, nskip = stages[i + 1]
                    want_split = nxt.lowres_ready(_ShapeOnly(x_.shape[0], up._net[3].out_channels, skip.shape[2], skip.shape[3]), nskip)
                    x_ = up.forward_split(x_, skip, out_fp32=not want_split, out_split=want_split,
                                          affine_of=affine[1] if i == 0 and affine is not None else None, f16=f16, sk_pre=pre_of(up, f16))
                xs = self.up4.forward_split(x_, b0, out_fp32=False, out_split=True, f16=f16, sk_pre=pre_of(self.up4, f16))
                if fin is not None:
                    # do_final_upscale (reference :99-101,116-117): a fifth stage against the IMAGE, in the same low-resolution form
                    # (tap GEMM at half resolution, a 3 x 3 convolution over the image's three channels, tap interpolation)
                    xs = fin.forward_split(xs, features[0], out_fp32=False, out_split=True, f16=f16)
                if _split_only and hip_ops.split_only_enabled():
                    sp = self._split3.run_split(xs, hip_ops.ACT_NONE, out_fp32=False, out_split=True)
                    return hip_ops.map_placeholder(sp), sp
                return self._split3.run_split(xs, hip_ops.ACT_NONE, out_fp32=True, out_split=True)

            calibrate = mode[1] and len(mode) == 2 and not torch.cuda.is_current_stream_capturing()
            try:
                if calibrate:
                    # FIRST eager forward of this weight version on fp16 pairs: the activations' range is measured once (host
                    # synchronisations, this call only).  The pairs' error floor is absolute (2^-25 per value) and their ceiling
                    # 65504: a tensor whose largest entry lies outside [2^-6, 4094] sends the pipeline to bf16 pairs -- reported.
                    hip_ops.range_check(True)
                try:
                    out, out_split = pipeline(mode[1])
                    rep = hip_ops.fp16_range_report() if calibrate else None
                finally:
                    if calibrate:
                        hip_ops.range_check(False)          # whatever happened: never leave the synchronising recorder on
                if calibrate:
                    mode = self._set_mode(wkey, (wkey, rep["ok"], rep))
                    if not rep["ok"]:
                        hip_ops.ROUTE_REPORT["Decoder"] = ("split pipeline on bf16 pairs instead of fp16 pairs: activation range "
                                                           f"{rep['out_of_range']} outside [2^-6, 65504 / 16] on the first batch")
                        out, out_split = pipeline(False)
            except Fp16Unsafe as e:
                if torch.cuda.is_current_stream_capturing():
                    raise
                hip_ops.ROUTE_REPORT["Decoder"] = f"split pipeline on bf16 pairs instead of fp16 pairs: {e}"
                self._set_mode(wkey, (wkey, False, None))
                out, out_split = pipeline(False)
            out._ocv_split = out_split
            return out
        for up, skip in stages:
            x = up(x, skip)
        if self.final_upscale is not None:
            x = self.final_upscale(x, features[0])
        if x.device.type == "cuda" and not self.training and not torch.is_grad_enabled():
            return self._split3(x) if self._split3.usable(x.shape[1]) else self._split3.exact(x)
        return self.conv3(x)


class DenseFeatureExtractor(nn.Module):
    """``args`` is the reference's config tree (see objcavit_amd.config).  Only
    the EfficientNet-B5 backbone is available locally; the reference's other
    encoder choices need torchvision / hub downloads (reference :141-168)."""

    def __init__(self, args, backbone: Optional[nn.Module] = None):
        super().__init__()
        self.args = args
        self.logger = logging.getLogger(__name__)
        block = self.args[self.args.model.name]
        self.n_bins = block.n_bins
        self.num_decoded_channels = 128
        self._encoder_params_module_list = []
        self._non_encoder_params_module_list = []

        name = block.encoder_name
        if backbone is None:
            if "efficientnet-b5" not in name:
                sys.exit(f"Error: encoder '{name}' is not available in this build (efficientnet-b5 only).")
            backbone = tf_efficientnet_b5_ap(pretrained=False)
        # remove unused final layers, as the reference does (:152-156)
        backbone.bn2 = nn.Identity()
        backbone.act2 = nn.Identity()
        backbone.global_pool = nn.Identity()
        backbone.classifier = nn.Identity()
        num_features = 2048 if "efficientnet-b5" in name else 1280

        self.decoder = Decoder(num_classes=128, num_features=num_features, bottleneck_features=num_features,
                               mode=block.get("mode"), encoder_name=name,
                               do_final_upscale=block.get("do_final_upscale"))
        keep = tuple(self.decoder.feature_select) + ((0,) if block.get("do_final_upscale") else ())
        self.encoder = Encoder(backbone, keep=keep)
        self._encoder_params_module_list.append(self.encoder)
        self._non_encoder_params_module_list.append(self.decoder)

    def skip_prepass(self, image, extra=None) -> Optional[SkipPrepass]:
        """The fork of this forward's encoder call (``SkipPrepass``; ``extra``: a callable issued on the same side stream behind the
        skip-part convolutions), or None: switched off (hip_ops.skip_overlap_enabled), not a GPU inference call, or the decoder's
        element type not settled yet."""
        fast = image.device.type == "cuda" and not self.training and not torch.is_grad_enabled()
        if not (fast and hip_ops.skip_overlap_enabled() and split_bf16_convs_enabled()):
            return None
        pre = SkipPrepass(self.decoder, image.device, extra)
        return pre if pre.stage_of else None

    def encode(self, image, pre: Optional[SkipPrepass] = None):
        """The encoder call of ``forward`` (``pre``: see ``skip_prepass``; whoever passes it joins it or hands it to the decoder)."""
        fast = image.device.type == "cuda" and not self.training and not torch.is_grad_enabled()
        if not fast:
            return self.encoder(image)
        return self.encoder(image, _defer_head=True, _on_feature=None if pre is None else pre.on_feature)

    def forward(self, image, _split_only: bool = False):
        fast = image.device.type == "cuda" and not self.training and not torch.is_grad_enabled()
        pre = self.skip_prepass(image)
        return self.decoder(self.encode(image, pre), _split_only=_split_only and fast, _skip_pre=pre)
