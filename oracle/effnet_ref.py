"""Functional restatement of the ``tf_efficientnet_b5_ap`` encoder as wrapped by
the reference's ``Encoder`` (ORACLE -- test infrastructure only).

PARITY UNPINNED for the encoder arithmetic.  The reference fetches the
architecture source and the weights at run time with
``torch.hub.load('rwightman/gen-efficientnet-pytorch', 'tf_efficientnet_b5_ap',
pretrained=True)`` (modules/DenseFeatureExtractor.py:149; hub default branch,
no pinned version); neither is present under /root/reference or in this
container.  What follows restates the published architecture of that model
family (EfficientNet, Tan & Le 2019; geffnet "tf_" variants):

  stem   conv3x3 s2, 3 -> 48, TF "SAME" padding, BN(eps 1e-3), swish
  stage  type  repeats  kernel  stride  expand  out-ch   (B5: width x1.6, depth x2.2)
    0    DS       3       3       1       1       24
    1    IR       5       3       2       6       40
    2    IR       5       5       2       6       64
    3    IR       7       3       2       6      128
    4    IR       7       5       1       6      176
    5    IR       9       5       2       6      304
    6    IR       3       3       1       6      512
  head   conv1x1 512 -> 2048 (bn2 / act2 / pool / classifier are replaced by
         nn.Identity at modules/DenseFeatureExtractor.py:152-156)
  squeeze-excite in every block, reduction = 0.25 x block INPUT channels,
  swish inside, sigmoid gate; residual when stride 1 and in == out.

What IS pinned: the order in which ``Encoder.forward`` collects activations
(modules/DenseFeatureExtractor.py:18-27) -- tests/golden G4 runs the
reference's own ``Encoder`` + ``Decoder`` classes around a local backbone.
Key names follow geffnet's module names so a real checkpoint's keys line up.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

BN_EPS = 1e-3
STEM = 48
HEAD = 2048
#            type  repeats k  s  expand out
STAGES = (("ds", 3, 3, 1, 1, 24),
          ("ir", 5, 3, 2, 6, 40),
          ("ir", 5, 5, 2, 6, 64),
          ("ir", 7, 3, 2, 6, 128),
          ("ir", 7, 5, 1, 6, 176),
          ("ir", 9, 5, 2, 6, 304),
          ("ir", 3, 3, 1, 6, 512))


def _same_pad(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """TensorFlow 'SAME' padding: total = max((ceil(i/s)-1)*s + k - i, 0), extra on the bottom/right."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph or pw:
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return x


def _conv_same(x, w, s=1, groups=1, bias=None):
    return F.conv2d(_same_pad(x, w.shape[-1], s), w, bias, stride=s, groups=groups)


def _bn(x, sd: SD, pfx: str):
    return F.batch_norm(x, sd[pfx + "running_mean"], sd[pfx + "running_var"], sd[pfx + "weight"], sd[pfx + "bias"],
                        False, 0.0, BN_EPS)


def _swish(x):
    return x * torch.sigmoid(x)


def _se(x, sd: SD, pfx: str):
    s = x.mean((2, 3), keepdim=True)
    s = _swish(F.conv2d(s, sd[pfx + "conv_reduce.weight"], sd[pfx + "conv_reduce.bias"]))
    s = F.conv2d(s, sd[pfx + "conv_expand.weight"], sd[pfx + "conv_expand.bias"])
    return x * torch.sigmoid(s)


def _block(x, sd: SD, pfx: str, kind: str, k: int, s: int, cin: int, cout: int):
    sc = x
    if kind == "ds":
        x = _swish(_bn(_conv_same(x, sd[pfx + "conv_dw.weight"], s, groups=cin), sd, pfx + "bn1."))
        x = _se(x, sd, pfx + "se.")
        x = _bn(F.conv2d(x, sd[pfx + "conv_pw.weight"]), sd, pfx + "bn2.")
    else:
        x = _swish(_bn(F.conv2d(x, sd[pfx + "conv_pw.weight"]), sd, pfx + "bn1."))
        mid = x.shape[1]
        x = _swish(_bn(_conv_same(x, sd[pfx + "conv_dw.weight"], s, groups=mid), sd, pfx + "bn2."))
        x = _se(x, sd, pfx + "se.")
        x = _bn(F.conv2d(x, sd[pfx + "conv_pwl.weight"]), sd, pfx + "bn3.")
    if s == 1 and cin == cout:
        x = x + sc
    return x


def encoder_features(image: torch.Tensor, sd: SD, pfx: str) -> List[torch.Tensor]:
    """Encoder.forward (modules/DenseFeatureExtractor.py:18-27): the activation
    after EVERY child of the backbone (children of ``blocks`` expanded):
    [x, conv_stem, bn1, act1, blocks.0 .. blocks.6, conv_head, bn2, act2,
    global_pool, classifier] -> 16 entries; feature_select = [4,5,6,8,11]."""
    feats = [image]
    feats.append(_conv_same(feats[-1], sd[pfx + "conv_stem.weight"], 2))
    feats.append(_bn(feats[-1], sd, pfx + "bn1."))
    feats.append(_swish(feats[-1]))
    cin = STEM
    for si, (kind, reps, k, s, _e, cout) in enumerate(STAGES):
        x = feats[-1]
        for r in range(reps):
            x = _block(x, sd, f"{pfx}blocks.{si}.{r}.", kind, k, s if r == 0 else 1, cin, cout)
            cin = cout
        feats.append(x)
    feats.append(F.conv2d(feats[-1], sd[pfx + "conv_head.weight"]))
    feats += [feats[-1]] * 4          # bn2, act2, global_pool, classifier == nn.Identity
    return feats
