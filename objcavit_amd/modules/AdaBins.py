"""Drop-in for the reference's ``modules/AdaBins.py`` (boundary A, no objects):
``AdaBins(args).forward(image) -> ReturnType(depth_pred, bin_edges)``.

The head ``conv_out`` (1x1 conv 128 -> n_bins + Softmax), the bin-centre
weighting and the pixel-wise dot product of the mini-ViT are ONE HIP kernel
(``ocv_bin_head_fwd``): range-attention maps, logits and probabilities are never
written to HBM (reference modules/AdaBins.py:77-87 + modules/layers.py:31-36).
"""
from __future__ import annotations

import logging
from collections import namedtuple

import torch
import torch.nn as nn

from .. import hip_ops
from .DenseFeatureExtractor import DenseFeatureExtractor
from .miniViT import mViT


def bin_edges_and_centers(bin_widths_normed: torch.Tensor, min_depth: float, max_depth: float):
    """widths -> edges -> centres (reference modules/AdaBins.py:79-83 == modules/GraphBins.py:111-115): what
    ``regress_bin_widths(..., depth_range)`` already left on the tensor (one launch with the normalisation), one launch of
    csrc/bin_edges.hip for other GPU tensors, the reference's torch formulation on the CPU."""
    stash = getattr(bin_widths_normed, "_ocv_bins", None)
    if stash is not None and stash[0] == (float(min_depth), float(max_depth)):
        return stash[1], stash[2]
    if bin_widths_normed.is_cuda:
        return hip_ops.bin_edges(bin_widths_normed.contiguous(), "none", min_depth, max_depth)[1:]
    widths = (max_depth - min_depth) * bin_widths_normed
    widths = nn.functional.pad(widths, (1, 0), mode='constant', value=min_depth)
    edges = torch.cumsum(widths, dim=1)
    centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
    return edges, centers.contiguous()


class AdaBins(nn.Module):
    images_are_independent = True      # an image's result does not depend on its batch mates (per object group: SURVEY.md Q3)

    def __init__(self, args, backbone: nn.Module = None):
        super().__init__()
        self.args = args
        self.logger = logging.getLogger(__name__)
        self.n_bins = self.args.adabins.n_bins
        self.num_decoded_channels = 128
        self._encoder_params_module_list = []
        self._non_encoder_params_module_list = []
        self._frozen_params_module_list = []
        self.ReturnType = namedtuple('ReturnType', ['depth_pred', 'bin_edges'])

        self.dense_feature_extractor = DenseFeatureExtractor(self.args, backbone=backbone)
        self._encoder_params_module_list.append(self.dense_feature_extractor.encoder)
        self._non_encoder_params_module_list.append(self.dense_feature_extractor.decoder)

        max_seq_len = 1200 if self.args[self.args.model.name].get('do_final_upscale') else 500
        self.adaptive_bins_layer = mViT(self.num_decoded_channels, n_query_channels=128, patch_size=16,
                                        dim_out=self.n_bins, embedding_dim=128, norm='linear', max_seq_len=max_seq_len)
        self._non_encoder_params_module_list.append(self.adaptive_bins_layer)

        self.conv_out = nn.Sequential(nn.Conv2d(128, self.n_bins, kernel_size=1, stride=1, padding=0), nn.Softmax(dim=1))
        self._non_encoder_params_module_list.append(self.conv_out)

    def get_encoder_params(self):
        for m in self._encoder_params_module_list:
            yield from m.parameters()

    def get_non_encoder_params(self):
        for m in self._non_encoder_params_module_list:
            yield from m.parameters()

    def get_frozen_params(self):
        for m in self._frozen_params_module_list:
            yield from m.parameters()

    def forward(self, image):
        """reference :73-89; on the GPU in eval / no_grad under the fp16 range guard (hip_ops.guarded_forward)."""
        return hip_ops.guarded_forward(self, self.dense_feature_extractor.decoder, image.device, lambda: self._forward(image))

    def _forward(self, image):
        unet_out = self.dense_feature_extractor(image, _split_only=True)        # (the heads read the split copy: hip_ops.map_placeholder)
        ds = self.args[self.args.basic.dataset]
        bin_widths_normed, feat, queries = self.adaptive_bins_layer.forward_parts(unet_out, (ds.min_depth, ds.max_depth))
        bin_edges, centers = bin_edges_and_centers(bin_widths_normed, ds.min_depth, ds.max_depth)
        conv = self.conv_out[0]
        depth_pred = hip_ops.bin_head(feat, queries, conv.weight.detach(), conv.bias.detach(), centers)
        return self.ReturnType(depth_pred=depth_pred, bin_edges=bin_edges)
