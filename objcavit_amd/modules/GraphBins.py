"""Drop-in for the reference's ``modules/GraphBins.py`` (boundary A):
``GraphBins(args).forward(image) -> ReturnType(depth_pred, bin_edges, detections)``.

The reference builds three frozen, no-grad producers in its constructor --
YOLOv7-seg, the WordNet phrase builder and CLIP (modules/GraphBins.py:33-35,
90-103).  They are inputs to the hot path, not part of it (SURVEY.md section 8,
rows N4 / out of scope), and none of their weights or sources exist offline,
so here they are ONE injectable callable:

    object_provider(image) -> (object_features: list[B] of N_i x 512 float,
                               object_xywh_list: list[B] of N_i x 4 (or None),
                               detections: B x 3 x H x W uint8 or None)

``SyntheticObjectProvider`` (the default) produces the deterministic boxes and
features the benchmark configurations use; a real detector + text encoder can
be plugged in without touching the model.
"""
from __future__ import annotations

import logging
import os
import sys
from collections import namedtuple
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import hip_ops
from .AdaBins import bin_edges_and_centers
from .DenseFeatureExtractor import DenseFeatureExtractor
from .ObjCAViT import ObjCAViT, PaddedObjects


class SyntheticObjectProvider:
    """Fixed number of objects per image; xywh uniform in the image (full-resolution pixels, the convention of
    modules/Yolov7Wrapper.py:118-123); features zeros for ``control_obj_zeros_512``
    (modules/LanguageEmbeddingWrapper.py:56-61) or N(0,1) * 10/sqrt(512) standing in for un-normalised CLIP text
    features.  Everything is generated once per (batch, size) on the host and cached on the device, so a forward
    contains no host->device traffic (hipGraph-friendly)."""

    def __init__(self, n_objects: int = 16, language: str = "control_obj_zeros_512", seed: int = 42, dim: int = 512):
        if language not in ("control_obj_zeros_512", "clip"):
            sys.exit(f"Error: Language model {language} not recognised")
        self.n, self.language, self.seed, self.dim = n_objects, language, seed, dim
        self._cache = {}

    def __call__(self, image: torch.Tensor):
        B, _, H, W = image.shape
        key = (B, H, W, image.device)
        if key not in self._cache:
            rs = np.random.RandomState(self.seed)
            boxes = np.stack([rs.uniform(0, W, (B, self.n)), rs.uniform(0, H, (B, self.n)),
                              rs.uniform(8, W / 2, (B, self.n)), rs.uniform(8, H / 2, (B, self.n))], axis=-1)
            if self.language == "clip":
                feats = rs.standard_normal((B, self.n, self.dim)) * (10.0 / np.sqrt(self.dim))
            else:
                feats = np.zeros((B, self.n, self.dim))
            self._cache[key] = (torch.from_numpy(feats.astype(np.float32)).to(image.device),
                                torch.from_numpy(boxes.astype(np.float32)).to(image.device))
        feats, boxes = self._cache[key]
        return [f for f in feats], [b for b in boxes], None

    def padded(self, image: torch.Tensor) -> PaddedObjects:
        """The same objects in the shape-static form (built once per (batch, size), counts on the device): GraphBins prefers
        this entry point when a provider has one -- no per-forward stack / pad / copy launches."""
        B = image.shape[0]
        key = ("padded", B, image.shape[2], image.shape[3], image.device)
        if key not in self._cache:
            self(image)
            feats, boxes = self._cache[(B, image.shape[2], image.shape[3], image.device)]
            self._cache[key] = PaddedObjects(feats, boxes, torch.full((B,), self.n, dtype=torch.int32, device=image.device))
        return self._cache[key]


def _record_on(result, stream) -> None:
    """``record_stream`` for every tensor of a (nested) result produced on another stream: without it the caching allocator hands a
    freed block back to ITS stream's next allocation while this stream's launches may still be reading it."""
    if isinstance(result, torch.Tensor):
        if result.is_cuda:
            result.record_stream(stream)
    elif isinstance(result, PaddedObjects):
        for t in (result.features, result.xywh, result.counts):
            _record_on(t, stream)
    elif isinstance(result, (tuple, list)):
        for t in result:
            _record_on(t, stream)


class GraphBins(nn.Module):
    images_are_independent = True      # an image's result does not depend on its batch mates (per object group: SURVEY.md Q3)

    def __init__(self, args, object_provider: Optional[Callable] = None, backbone: nn.Module = None):
        super().__init__()
        self.args = args
        self.logger = logging.getLogger(__name__)
        self._encoder_params_module_list = []
        self._non_encoder_params_module_list = []
        self._frozen_params_module_list = []
        self.ReturnType = namedtuple('ReturnType', ['depth_pred', 'bin_edges', 'detections'])

        self.dense_feature_extractor = DenseFeatureExtractor(self.args, backbone=backbone)
        self._encoder_params_module_list.append(self.dense_feature_extractor.encoder)
        self._non_encoder_params_module_list.append(self.dense_feature_extractor.decoder)

        oc = self.args[self.args.model.name].objcavit
        self.object_provider = object_provider or SyntheticObjectProvider(
            16, oc.get("language_embedding_strategy") or "control_obj_zeros_512")

        max_seq_len = 1200 if self.args[self.args.model.name].get('do_final_upscale') else 500
        self.objcavit = ObjCAViT(self.args, n_query_channels=128, patch_size=16, im_feature_dim=128,
                                 obj_feature_dim=512, embedding_dim=oc.embedding_dim,
                                 dim_out=self.args.graphbins.n_bins, norm='linear', max_seq_len=max_seq_len)
        self._non_encoder_params_module_list.append(self.objcavit)

        self.conv_out = nn.Sequential(nn.Conv2d(oc.embedding_dim, self.args.graphbins.n_bins, kernel_size=1, stride=1, padding=0),
                                      nn.Softmax(dim=1))
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

    def forward_until_head(self, image, object_features=None, object_xywh_list: Optional[List[Optional[torch.Tensor]]] = None,
                           pad_objects_to: Optional[int] = None, object_group: Optional[int] = None):
        """Everything up to the inputs of the fused bin head: (feat, queries, centers, bin_edges, detections).
        Split out so that a hipGraph can capture it while the head kernel stays individually timeable.
        ``object_features``: None (ask the provider), the reference's list of N_i x 512 tensors (with ``object_xywh_list``), or a
        ``PaddedObjects`` (shape-static, counts on the device: what a captured graph holds).  ``object_group``: images per
        reference call when the batch is several calls' batches at once (ValidationStep: image + mirror)."""
        detections = None
        if object_features is None:
            with torch.no_grad():
                padded = getattr(self.object_provider, "padded", None)
                if padded is not None:
                    object_features = padded(image)
                else:
                    object_features, object_xywh_list, detections = self.object_provider(image)
        if not isinstance(object_features, PaddedObjects):
            object_features = [nf.float() for nf in object_features]
        pre = None
        if image.is_cuda and self.objcavit.can_prepass() and hip_ops.object_prepass_enabled():
            # The object branch (embedding, positional MLP, first self-attention stack: ~20 launches of a few workgroups each,
            # ~0.5 ms) does not depend on the image: it runs on a second stream beside the encoder and is joined before the decoder
            # starts (inside one hipGraph segment when the forward is captured).  On its own that gains nothing (round 2, bs 16,
            # alternating runs: 21.71 / 21.76 ms with it, 21.67 / 21.61 without -- the encoder's launches fill the chip); it is the
            # default since round 4 because it leaves ONE side chain behind the decoder, the image tokens', which then runs beside
            # the heads' 3x3 convolution (hip_ops.head_overlap_enabled; ObjCAViT.forward_parts).
            # With the decoder's skip-part convolutions on the same side stream (SkipPrepass) the two share ONE fork, behind the
            # encoder's fourth stage: a side branch with two incoming edges from the main chain replays pathologically slowly.
            main = torch.cuda.current_stream(image.device)
            side = hip_ops.side_stream(image.device)
            dfe = self.dense_feature_extractor

            def object_branch():
                return self.objcavit.object_prepass(object_features, object_xywh_list, image.device, pad_objects_to)

            skip_pre = dfe.skip_prepass(image, extra=object_branch)
            if skip_pre is None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pre = object_branch()
            encoded = dfe.encode(image, skip_pre)
            main.wait_stream(side)
            if skip_pre is not None:
                skip_pre.joined = True             # (same side stream: the wait above is its join too)
                pre = skip_pre.extra_result
            _record_on(pre, main)                  # allocated on the side stream, read on this one: the allocator must know
            dense_features = dfe.decoder(encoded, _split_only=not torch.is_grad_enabled() and not self.training, _skip_pre=skip_pre)
        else:
            dense_features = self.dense_feature_extractor(image, _split_only=True)   # (the heads read the split copy: hip_ops.map_placeholder)
        bin_widths_normed, feat, queries = self.objcavit.forward_parts(dense_features, object_features, object_xywh_list, pre=pre,
                                                                     pad_objects_to=pad_objects_to, object_group=object_group)
        ds = self.args[self.args.basic.dataset]
        bin_edges, centers = bin_edges_and_centers(bin_widths_normed, ds.min_depth, ds.max_depth)
        return feat, queries, centers, bin_edges, detections

    def head(self, feat, queries, centers):
        conv = self.conv_out[0]
        return hip_ops.bin_head(feat, queries, conv.weight.detach(), conv.bias.detach(), centers)

    def forward(self, image, object_features=None, object_xywh_list: Optional[List[Optional[torch.Tensor]]] = None,
                pad_objects_to: Optional[int] = None, object_group: Optional[int] = None):
        """``pad_objects_to``: the longest object list of the GLOBAL batch when ``image`` is one rank's shard of it
        (objcavit_amd/dp.py ``sharded_forward``); None = this batch's own maximum, as the reference pads.
        ``object_group``: see ``forward_until_head``."""
        def run():
            feat, queries, centers, bin_edges, detections = self.forward_until_head(image, object_features, object_xywh_list,
                                                                                    pad_objects_to, object_group)
            depth_pred = self.head(feat, queries, centers)
            return self.ReturnType(depth_pred=depth_pred, bin_edges=bin_edges, detections=detections)

        # (GPU, eval, no_grad: under the fp16 range guard -- a batch beyond the fp16 pairs' range is re-run on bf16 pairs and reported)
        return hip_ops.guarded_forward(self, self.dense_feature_extractor.decoder, image.device, run)
