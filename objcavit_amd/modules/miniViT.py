"""Drop-in for the reference's ``modules/miniViT.py`` (class ``mViT``): same
constructor, forward signature, return values and state_dict keys; the
patch-embedding convolution, the four transformer layers, the regressor and
the pixel-wise dot product run as HIP kernels, the 3x3 convolution as the
split-bf16 implicit GEMM of csrc/conv_igemm.hip (exact fp32 on request:
OCV_CONV=exact).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from .. import hip_ops
from .layers import PatchTransformerEncoder, PixelWiseDotProduct


def regress_bin_widths(regressor: nn.Sequential, head: torch.Tensor, norm: str, depth_range=None) -> torch.Tensor:
    """regressor MLP (Linear, LeakyReLU, Linear, LeakyReLU, Linear) + normalisation
    (reference modules/miniViT.py:33-42 == modules/ObjCAViT.py:378-388).  ``depth_range`` = (min_depth, max_depth): the
    normalisation, the bin edges and the bin centres come out of ONE launch (csrc/bin_edges.hip); edges and centres ride along
    on the returned tensor for ``AdaBins.bin_edges_and_centers`` (reference modules/AdaBins.py:79-83)."""
    # the one-launch form takes exactly the reference's stack: Linear, LeakyReLU, Linear, LeakyReLU (the same slope), Linear, with
    # contiguous fp32 parameters; anything else (another activation, a bias-free or strided layer) goes layer by layer below
    fits = (len(regressor) == 5
            and all(isinstance(regressor[i], nn.Linear) and regressor[i].bias is not None and regressor[i].in_features % 4 == 0
                    and regressor[i].in_features <= 1024
                    and all(t.dtype == torch.float32 and t.is_contiguous() for t in (regressor[i].weight, regressor[i].bias))
                    for i in (0, 2, 4))
            and regressor[4].out_features <= 4096
            and all(isinstance(regressor[i], nn.LeakyReLU) for i in (1, 3))
            and regressor[1].negative_slope == regressor[3].negative_slope)
    if depth_range is not None and norm != "softmax" and head.is_cuda and head.dim() == 2 and head.stride(1) == 1 and fits:
        # ONE launch for the three layers, the normalisation, the edges and the centres (csrc/bin_edges.hip: regressor_bins_kernel)
        lo, hi = float(depth_range[0]), float(depth_range[1])
        par = [t.detach() for i in (0, 2, 4) for t in (regressor[i].weight, regressor[i].bias)]
        w, edges, centers = hip_ops.regressor_bins(head, *par, "linear" if norm == "linear" else "sigmoid", lo, hi,
                                                   leaky_slope=regressor[1].negative_slope)
        w._ocv_bins = ((lo, hi), edges, centers)
        return w
    for i in (1, 3):                                          # (hip_ops.linear's LeakyReLU is the reference's: slope 0.01)
        if not isinstance(regressor[i], nn.LeakyReLU) or regressor[i].negative_slope != 0.01:
            raise NotImplementedError(f"regressor[{i}] = {regressor[i]!r}: the HIP path implements the reference's LeakyReLU(0.01) "
                                      "(modules/miniViT.py:17-19, modules/ObjCAViT.py:299-303)")
    y = hip_ops.linear(head.contiguous(), regressor[0].weight.detach(), regressor[0].bias.detach(), hip_ops.ACT_LEAKY_RELU)
    y = hip_ops.linear(y, regressor[2].weight.detach(), regressor[2].bias.detach(), hip_ops.ACT_LEAKY_RELU)
    y = hip_ops.linear(y, regressor[4].weight.detach(), regressor[4].bias.detach(), hip_ops.ACT_NONE)
    if depth_range is not None:
        lo, hi = float(depth_range[0]), float(depth_range[1])
        mode = "linear" if norm == "linear" else ("none" if norm == "softmax" else "sigmoid")
        w, edges, centers = hip_ops.bin_edges(torch.softmax(y, dim=1) if norm == "softmax" else y, mode, lo, hi)
        w._ocv_bins = ((lo, hi), edges, centers)
        return w
    if norm == "linear":
        y = torch.relu(y) + 0.1
    elif norm == "softmax":
        return torch.softmax(y, dim=1)
    else:
        y = torch.sigmoid(y)
    return y / y.sum(dim=1, keepdim=True)


class mViT(nn.Module):
    def __init__(self, in_channels, n_query_channels=128, patch_size=16, dim_out=256,
                 embedding_dim=128, num_heads=4, norm='linear', max_seq_len=500):
        super().__init__()
        self.norm = norm
        self.n_query_channels = n_query_channels
        self.patch_transformer = PatchTransformerEncoder(in_channels, patch_size, embedding_dim, num_heads, max_seq_len)
        self.dot_product_layer = PixelWiseDotProduct()
        self.conv3x3 = nn.Conv2d(in_channels, embedding_dim, kernel_size=3, stride=1, padding=1)
        self.regressor = nn.Sequential(nn.Linear(embedding_dim, 256), nn.LeakyReLU(),
                                       nn.Linear(256, 256), nn.LeakyReLU(),
                                       nn.Linear(256, dim_out))

    def forward_parts(self, x: torch.Tensor, depth_range=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (bin_widths_normed B x dim_out, conv3x3 features B x E x h x w, queries B x n_query x E (view)).
        Used by AdaBins.forward so that the range-attention maps are never materialised (``depth_range``: regress_bin_widths)."""
        def tokens():
            tok = self.patch_transformer.forward_batch_first(x)       # B x S x E
            if tok.shape[1] < self.n_query_channels + 1:
                raise ValueError(f"need at least {self.n_query_channels + 1} patches, got {tok.shape[1]}")
            return tok, regress_bin_widths(self.regressor, tok[:, 0, :], self.norm, depth_range)

        if x.is_cuda and hip_ops.head_overlap_enabled():
            # the 3x3 convolution fills the chip on this stream while the token chain's small launches run beside it
            # (hip_ops.head_overlap_enabled; joined in front of the bin head)
            main = torch.cuda.current_stream(x.device)
            tst = hip_ops.side_stream(x.device, 1)
            tst.wait_stream(main)
            with torch.cuda.stream(tst):
                tok, y = tokens()
            with hip_ops.islands_suspended():
                feat = self._conv3x3_nhwc(x)
            main.wait_stream(tst)
            for t in (tok, y):
                t.record_stream(main)
        else:
            tok, y = tokens()
            feat = self._conv3x3_nhwc(x)
        return y, feat, tok[:, 1:self.n_query_channels + 1, :]

    def _conv3x3_nhwc(self, x):
        from .DenseFeatureExtractor import SplitConv3x3
        plan = self.__dict__.get("_split3x3")
        if plan is None:
            plan = self.__dict__["_split3x3"] = SplitConv3x3(self.conv3x3)
        if plan.usable(x.shape[1]):
            pre = getattr(x, "_ocv_split", None)                      # the decoder's conv3 leaves its split copy here
            if pre is not None and tuple(pre.shape) == tuple(x.shape):
                from .DenseFeatureExtractor import Fp16Unsafe
                try:
                    return plan.run_split(pre)
                except Fp16Unsafe as e:                                # (reported, not silent; the fp32 map takes the bf16-pair kernel)
                    if torch.cuda.is_current_stream_capturing():
                        raise
                    hip_ops.ROUTE_REPORT["heads.conv3x3"] = f"bf16 pairs on the fp32 map: {e}"
            return plan(hip_ops.fp32_map(x))                          # split-bf16 implicit GEMM, NHWC in / out
        return plan.exact(hip_ops.fp32_map(x))                        # OCV_CONV=exact, or channels not a multiple of 4

    def forward(self, x):
        y, feat, queries = self.forward_parts(x)
        return y, self.dot_product_layer(feat, queries)
