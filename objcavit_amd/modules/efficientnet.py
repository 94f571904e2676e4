"""EfficientNet-B5 ("tf_efficientnet_b5_ap" layout) backbone, defined locally.

The reference obtains this network with ``torch.hub.load('rwightman/
gen-efficientnet-pytorch', 'tf_efficientnet_b5_ap', pretrained=True)``
(modules/DenseFeatureExtractor.py:149) -- a run-time fetch of third-party
source and weights.  Here the architecture is a local module whose child
order (conv_stem, bn1, act1, blocks, conv_head, bn2, act2, global_pool,
classifier) and parameter names follow that model family, so that

* the reference's ``Encoder`` wrapper (modules/DenseFeatureExtractor.py:18-27)
  collects the same 16 activations with feature_select = [4, 5, 6, 8, 11]
  giving 24/40/64/176/2048 channels, and
* a checkpoint of the reference (``...encoder.original_model.blocks.3.2.conv_dw
  .weight`` etc.) loads by key.

Inference plan on the GPU (eval + no_grad): every
BatchNorm is folded into the preceding convolution once and cached per
parameter version; stem, 1x1 expand / project (+ gate + skip), depthwise
(+ squeeze-excite pooling) and the gate all run as hand-written NHWC HIP
kernels (csrc/stem.hip, pointwise_split.hip, depthwise_se.hip,
mbconv_fused.hip) -- nothing of the encoder reaches MIOpen / hipBLASLt.  In
training mode or on the CPU the plain PyTorch module graph runs (that is also
what the golden generator wraps in the reference's ``Encoder``).
Architecture table: oracle/effnet_ref.py header (width x1.6, depth x2.2,
TF "SAME" padding, BN eps 1e-3, swish, squeeze-excite 0.25 of block input).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops

BN_EPS = 1e-3
BN_MOMENTUM = 0.01

# (kind, repeats, kernel, stride, expand, out_channels)
B5_STAGES = (("ds", 3, 3, 1, 1, 24), ("ir", 5, 3, 2, 6, 40), ("ir", 5, 5, 2, 6, 64), ("ir", 7, 3, 2, 6, 128),
             ("ir", 7, 5, 1, 6, 176), ("ir", 9, 5, 2, 6, 304), ("ir", 3, 3, 1, 6, 512))
B5_STEM, B5_HEAD = 48, 2048


class Conv2dSame(nn.Conv2d):
    """Conv2d with TensorFlow 'SAME' padding computed from the input size."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, k, stride=stride, padding=0, groups=groups, bias=bias)

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        ih, iw = x.shape[-2:]
        ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
        pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
        if ph == pw and ph % 2 == 0:          # symmetric: let the conv kernel pad
            return F.conv2d(x, self.weight, self.bias, self.stride, ph // 2, 1, self.groups)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, 1, self.groups)


class SqueezeExcite(nn.Module):
    def __init__(self, chs, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, reduced, 1, bias=True)
        self.act1 = nn.SiLU()
        self.conv_expand = nn.Conv2d(reduced, chs, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = self.conv_expand(self.act1(self.conv_reduce(s)))
        return x * torch.sigmoid(s)


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOMENTUM)


def fold_bn(conv: nn.Conv2d, bn: nn.Module):
    """(weight, bias) of conv followed by eval-mode BatchNorm (identity if ``bn`` is not a BatchNorm)."""
    w = conv.weight.detach()
    b = conv.bias.detach() if conv.bias is not None else None
    if not isinstance(bn, nn.BatchNorm2d):
        return w.contiguous(), b
    s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    b0 = b if b is not None else torch.zeros_like(bn.running_mean)
    return (w * s.view(-1, 1, 1, 1)).contiguous(), ((b0 - bn.running_mean) * s + bn.bias.detach()).contiguous()


def _dw_tap_major(w: torch.Tensor) -> torch.Tensor:
    """depthwise weight [C, 1, k, k] -> [k*k, C] (the NHWC kernel reads 4 channels of one tap per lane)."""
    return w.flatten(1).t().contiguous()


def _se_params(se: "SqueezeExcite"):
    return (se.conv_reduce.weight.detach().flatten(1).contiguous(), se.conv_reduce.bias.detach().contiguous(),
            se.conv_expand.weight.detach().flatten(1).t().contiguous(), se.conv_expand.bias.detach().contiguous())


class Conv1x1(nn.Conv2d):
    """1x1 convolution without bias (conv_head); on the GPU in eval mode it runs as the NHWC pointwise kernel."""

    def forward(self, x):
        if (not self.training) and (not torch.is_grad_enabled()) and x.device.type == "cuda" and self.bias is None \
                and x.shape[1] % 8 == 0:
            c = self.__dict__.get("_pw_cache")
            key = (x.device, self.weight._version, self.weight.data_ptr())
            if c is None or c[0] != key:
                c = (key, hip_ops.pointwise_weight(self.weight))
                self.__dict__["_pw_cache"] = c
            return hip_ops.pointwise_nhwc(x, c[1], None, hip_ops.ACT_NONE)
        return super().forward(x)


class _FoldedMixin:
    """Caches BN-folded weights for the inference fast path; dropped on train() / load_state_dict / device moves."""

    def _fast(self, x: torch.Tensor) -> bool:
        return (not self.training) and (not torch.is_grad_enabled()) and x.device.type == "cuda"

    def _fold_key(self, x: torch.Tensor):
        """Identity AND version of every parameter / buffer that enters the fold: an in-place update in eval mode
        (param.copy_(), EMA / SWA weight swap, edited BN statistics) must re-fold, like every other weight cache here."""
        ts = list(self.parameters(recurse=True)) + list(self.buffers(recurse=True))
        return (x.device,) + tuple((t.data_ptr(), t._version) for t in ts)

    def _folded(self, x: torch.Tensor):
        c = self.__dict__.get("_fold_cache")
        key = self._fold_key(x)
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("BN-folded weights are stale during graph capture: run one eager warm-up call first")
            with torch.no_grad():
                c = (key, self._fold())
            self.__dict__["_fold_cache"] = c
        return c[1]

    def train(self, mode: bool = True):
        self.__dict__.pop("_fold_cache", None)
        return super().train(mode)

    def _load_from_state_dict(self, *a, **kw):
        self.__dict__.pop("_fold_cache", None)
        return super()._load_from_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_fold_cache", None)
        return super()._apply(fn, *a, **kw)


class DepthwiseSeparableConv(_FoldedMixin, nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.has_residual = stride == 1 and cin == cout
        self.conv_dw = Conv2dSame(cin, cin, k, stride, groups=cin)
        self.bn1 = _bn(cin)
        self.act1 = nn.SiLU()
        self.se = SqueezeExcite(cin, max(1, int(cin * 0.25)))
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = _bn(cout)
        self.act2 = nn.Identity()

    def _fold(self):
        wd, bd = fold_bn(self.conv_dw, self.bn1)
        wp, bp = fold_bn(self.conv_pw, self.bn2)
        return (_dw_tap_major(wd), bd, hip_ops.pointwise_weight(wp), bp) + _se_params(self.se)

    def forward(self, x):
        if self._fast(x):
            # NHWC plan: depthwise+BN+SiLU -> squeeze-excite gate -> 1x1 (+BN) with the gate on its input + skip add
            wd, bd, wp, bp, s1, sb1, s2, sb2 = self._folded(x)
            k = self.conv_dw.kernel_size[0]
            y, g = hip_ops.depthwise_se_gate(x, wd, bd, k, self.conv_dw.stride[0], s1, sb1, s2, sb2)
            return hip_ops.pointwise_nhwc(y, wp, bp, hip_ops.ACT_NONE, gate=g, residual=x if self.has_residual else None)
        y = self.act1(self.bn1(self.conv_dw(x)))
        y = self.act2(self.bn2(self.conv_pw(self.se(y))))
        return y + x if self.has_residual else y


class InvertedResidual(_FoldedMixin, nn.Module):
    def __init__(self, cin, cout, k, stride, expand):
        super().__init__()
        mid = cin * expand
        self.has_residual = stride == 1 and cin == cout
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = _bn(mid)
        self.act1 = nn.SiLU()
        self.conv_dw = Conv2dSame(mid, mid, k, stride, groups=mid)
        self.bn2 = _bn(mid)
        self.act2 = nn.SiLU()
        self.se = SqueezeExcite(mid, max(1, int(cin * 0.25)))
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = _bn(cout)

    def _fold(self):
        we, be = fold_bn(self.conv_pw, self.bn1)
        wd, bd = fold_bn(self.conv_dw, self.bn2)
        wl, bl = fold_bn(self.conv_pwl, self.bn3)
        # wl also as the plain fp32 matrix: the pre-split route folds the squeeze-excite gate into it per image
        return (hip_ops.pointwise_weight(we), be, _dw_tap_major(wd), bd, hip_ops.pointwise_weight(wl), bl) + _se_params(self.se) + \
            (wl.flatten(1).contiguous(),)

    def forward(self, x):
        if self._fast(x):
            # NHWC plan, 4 launches: expand 1x1 (+BN+SiLU) -> depthwise (+BN+SiLU, + pooling partials) -> gate ->
            # project 1x1 (+BN) with the gate applied to its input rows and the skip connection added in the epilogue
            we, be, wd, bd, wl, bl, s1, sb1, s2, sb2, wl_f32 = self._folded(x)
            k, stride = self.conv_dw.kernel_size[0], self.conv_dw.stride[0]
            B, cin, H, W = x.shape
            mid, cout = self.conv_pw.out_channels, self.conv_pwl.out_channels
            Ho, Wo = -(-H // stride), -(-W // stride)
            split_w = isinstance(we, hip_ops.SplitWeight)
            res = x if self.has_residual else None
            if hip_ops.expand_depthwise_fusable(cin, we, k):
                # 3 launches: the expanded tensor stays in LDS (csrc/mbconv_fused.hip)
                y, g = hip_ops.expand_depthwise_se_gate(x, we, be, wd, bd, k, stride, s1, sb1, s2, sb2)
                return hip_ops.pointwise_nhwc(y, wl, bl, hip_ops.ACT_NONE, gate=g, residual=res)
            y = hip_ops.pointwise_nhwc(x, we, be, hip_ops.ACT_SILU)
            if split_w and hip_ops.pointwise_hl_project_pays(B, Ho * Wo, mid, cout):
                # stage 5's projects (long K, weights well under the rows' traffic): the depthwise output written ONCE, pre-split
                # (hl32), read by LDS-DMA, the gate folded into per-image project weights (csrc/pointwise_hl.hip)
                y_hl, wg = hip_ops.depthwise_se_gate_weights(y, wd, bd, k, stride, s1, sb1, s2, sb2, wl_f32)
                return hip_ops.pointwise_hl(y_hl, wg, bl, hip_ops.ACT_NONE, residual=res, out_fp32=True)
            y, g = hip_ops.depthwise_se_gate(y, wd, bd, k, stride, s1, sb1, s2, sb2)
            return hip_ops.pointwise_nhwc(y, wl, bl, hip_ops.ACT_NONE, gate=g, residual=res)
        y = self.act1(self.bn1(self.conv_pw(x)))
        y = self.act2(self.bn2(self.conv_dw(y)))
        y = self.bn3(self.conv_pwl(self.se(y)))
        return y + x if self.has_residual else y


class GenEfficientNet(nn.Module):
    def __init__(self, stages=B5_STAGES, stem=B5_STEM, head=B5_HEAD, num_classes=1000):
        super().__init__()
        self.conv_stem = Conv2dSame(3, stem, 3, stride=2)
        self.bn1 = _bn(stem)
        self.act1 = nn.SiLU()
        blocks, cin = [], stem
        for kind, reps, k, s, e, cout in stages:
            stage = []
            for r in range(reps):
                stride = s if r == 0 else 1
                stage.append(DepthwiseSeparableConv(cin, cout, k, stride) if kind == "ds"
                             else InvertedResidual(cin, cout, k, stride, e))
                cin = cout
            blocks.append(nn.Sequential(*stage))
        self.blocks = nn.Sequential(*blocks)
        self.conv_head = Conv1x1(cin, head, 1, bias=False)
        self.bn2 = _bn(head)
        self.act2 = nn.SiLU()
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Linear(head, num_classes)

    def forward(self, x):
        x = self.act1(self.bn1(self.conv_stem(x)))
        x = self.act2(self.bn2(self.conv_head(self.blocks(x))))
        return self.classifier(self.global_pool(x).flatten(1))


def tf_efficientnet_b5_ap(pretrained: bool = False) -> GenEfficientNet:
    """Local constructor standing in for the hub entry point of the same name.
    There is no network here: ``pretrained`` weights must be loaded by the
    caller from a state_dict / checkpoint."""
    return GenEfficientNet()
