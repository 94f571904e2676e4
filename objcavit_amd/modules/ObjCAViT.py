"""Drop-in for the reference's ``modules/ObjCAViT.py``: ``ObjCAViT``,
``SelfAttnCrossAttn`` and ``GridRandomPositionalEmbeddings`` with the same
constructor arguments, forward signatures, return values and state_dict keys
(including the prototype-layer keys ``image_encoder_layers.*`` /
``obj_encoder_layers.*`` of reference :155-161).

What runs where
---------------
HIP (libobjcavit_hip.so): patch-embedding convolution fused with bias +
  positional embedding + token layout, every transformer layer (packed QKV
  projection, masked multi-head attention, out-proj + residual + LayerNorm,
  FFN + residual + LayerNorm), both cross-attentions, every Linear (object
  embedding, positional MLP, regressor) and the pixel-wise dot product.
  The 3x3 convolution is the split-bf16 implicit GEMM of csrc/conv_igemm.hip
  (ocv_conv_nhwc_split_fwd) and the grid_random / grid_random_roi_align
  positional strategies are ocv_pos_grid_sample_fwd (csrc/pos_sample.hip).
PyTorch glue (a few KB of data): padding ragged object lists, bin-width
  normalisation.

Reference behaviours that are reproduced on purpose:
Q1 key rows are FRONT-padded with 1e-4 while the mask is BACK-padded; Q2 the
first cross-attention takes V from the IMAGE tokens; Q3 ``saca_2`` receives a
B x S x E tensor; Q4 padded object rows leave the object encoder as exact
zeros; Q6 the coordinate normalisation of ``grid_random``.
"""
from __future__ import annotations

import math
import sys
from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops
from .layers import HipEncoderStack, PatchTransformerEncoder, PixelWiseDotProduct  # noqa: F401
from .miniViT import regress_bin_widths

PAD_VALUE = 0.0001          # reference :183,194


class _PerDevice:
    """Small constant device tensors built on first use (outside graph capture) and reused: building them inside a
    forward would be a host -> device copy, which a hipGraph capture cannot contain.  The cache is a small LRU
    (``capacity`` entries): with a real detector almost every batch has a new tuple of object counts, and an unbounded
    dict would grow by one device tensor per batch for the life of the process."""

    def __init__(self, make, capacity: int = 16):
        self._make, self._vals, self._cap, self._pinned = make, OrderedDict(), capacity, {}

    def get(self, dev, *key):
        k = (str(dev),) + key
        v = self._vals.get(k)
        if v is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("constant not cached yet: run one eager warm-up call before graph capture")
            v = self._vals[k] = self._make(dev, *key)
            while len(self._vals) > self._cap:
                self._vals.popitem(last=False)          # (tensors a captured graph reads are held in _pinned)
        else:
            self._vals.move_to_end(k)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self._pinned[k] = v                         # a hipGraph now reads this address on every replay: never freed
        elif isinstance(v, torch.Tensor) and v.is_cuda:
            # built on whichever stream was current at first use, read on any (side streams, slot streams): the caching allocator
            # must not hand an evicted entry's block out again while a launch of another stream still reads it
            v.record_stream(torch.cuda.current_stream(v.device))
        return v

    def __len__(self):
        return len(self._vals)


_NO_BOX = _PerDevice(lambda dev: torch.full((1, 4), -1.0, device=dev))                      # reference :313
_RAGGED_MASK = _PerDevice(lambda dev, counts, nmax: (torch.arange(nmax)[None, :] >= torch.tensor(counts)[:, None]).to(dev))
_COUNTS = _PerDevice(lambda dev, counts: torch.tensor(counts, dtype=torch.int32).to(dev), capacity=64)


class PaddedObjects:
    """The object inputs of a batch in shape-static form: ``features`` [B, cap, F] fp32, ``xywh`` [B, cap, >= 2 | 4] fp32,
    ``counts`` int32 [B] ON THE DEVICE with 1 <= counts[b] <= cap (rows beyond an image's count are ignored; an image without
    detections has ONE row, the <UNK> feature with the box (-1, -1, -1, -1): reference :311-315).  Nothing on the path reads the
    counts on the host, so a captured hipGraph replays with whatever the buffers hold (objcavit_amd/graph.py); an object provider
    may return one directly instead of the reference's two lists."""
    __slots__ = ("features", "xywh", "counts", "max_count")

    def __init__(self, features: torch.Tensor, xywh: torch.Tensor, counts: torch.Tensor, max_count: Optional[int] = None):
        if features.dim() != 3 or xywh.dim() != 3 or xywh.shape[:2] != features.shape[:2] or counts.shape != features.shape[:1]:
            raise ValueError("PaddedObjects: expected features [B, cap, F], xywh [B, cap, k], counts [B]")
        if counts.dtype != torch.int32:
            raise TypeError("PaddedObjects: counts must be int32")
        self.features, self.xywh, self.counts = features, xywh, counts
        self.max_count = max_count          # the longest list, when the host knows it (built from lists): argument checks only

    @property
    def capacity(self) -> int:
        return int(self.features.shape[1])

    @staticmethod
    def from_lists(object_features, object_xywh_list, device, capacity: Optional[int] = None) -> "PaddedObjects":
        """The reference's two lists (N_i x F features, N_i x 4 boxes or None) -> padded tensors + device counts.  The counts
        tensor comes from a small LRU keyed on the tuple of counts (a host -> device copy: not inside a graph capture)."""
        B = len(object_features)
        if object_xywh_list is None:
            object_xywh_list = [None] * B          # no boxes at all: every image carries the <UNK> box (reference :313)
        if len(object_xywh_list) != B:
            raise ValueError("object_features / object_xywh_list must have one entry per image")
        boxes = [_NO_BOX.get(device) if b is None else b.to(device, torch.float32) for b in object_xywh_list]    # :313
        counts = [int(f.shape[0]) for f in object_features]
        for c, b in zip(counts, boxes):
            if b.dim() != 2 or b.shape[0] != c or c < 1:
                raise ValueError("object_features and object_xywh_list disagree on the number of objects (or an image has none: "
                                 "the reference hands such an image ONE <UNK> row)")
        cap = max(max(counts), int(capacity or 0))
        width = min(b.shape[1] for b in boxes)
        feats = [f.to(device, torch.float32) for f in object_features]
        if all(c == cap for c in counts):
            f3, b3 = torch.stack(feats, 0), torch.stack([b[:, :width] for b in boxes], 0)
        else:
            f3 = nn.utils.rnn.pad_sequence(feats, batch_first=True)
            b3 = nn.utils.rnn.pad_sequence([b[:, :width] for b in boxes], batch_first=True)
            if f3.shape[1] < cap:
                f3, b3 = F.pad(f3, (0, 0, 0, cap - f3.shape[1])), F.pad(b3, (0, 0, 0, cap - b3.shape[1]))
        return PaddedObjects(f3.contiguous(), b3.contiguous(), _COUNTS.get(device, tuple(counts)), max(counts))


# ---------------------------------------------------------------------------
# positional embeddings
# ---------------------------------------------------------------------------
class GridRandomPositionalEmbeddings(nn.Module):
    """One learnable vector per image patch (reference :18-147).  ``forward`` is ONE launch of
    ``ocv_pos_grid_sample_fwd`` (csrc/pos_sample.hip) whatever the mode: the reference's coordinate normalisation
    (SURVEY.md Q6), F.grid_sample and torchvision's ps_roi_align (third-party, restated: DESIGN.md section 2) all
    happen inside the kernel -- no host synchronisation, no per-image loop, capturable."""

    def __init__(self, args, embedding_dim, patch_size, mode="centre"):
        super().__init__()
        self.args = args
        self.embedding_dim = embedding_dim
        self.patch_size = patch_size
        self.mode = mode
        assert self.mode in ["centre", "roi_align"], "Error: unrecognised GridRandomPositionalEmbeddings mode."
        ds = self.args[self.args.basic.dataset]
        lengths = [math.ceil(d[0] / patch_size) * math.ceil(d[1] / patch_size)
                   for d in (ds.dimensions_train, ds.dimensions_test)]
        self.sequence_length = max(lengths)
        self.positional_encodings = nn.Parameter(torch.rand(self.sequence_length, self.embedding_dim), requires_grad=True)

    def forward(self, coords, image_features, input_coord_space="img", factor=2.0, addend=None):
        """coords N x {2,4} ("obj", full-resolution pixels) -> N x E, or B x S x {2,4} ("img") -> B x S x E.
        ``addend`` (same shape as the result) is added to the samples inside the kernel."""
        fh, fw = image_features.shape[2], image_features.shape[3]
        gh, gw = math.ceil(fh / self.patch_size), math.ceil(fw / self.patch_size)          # :78-79
        table = self.positional_encodings.detach()
        need = 2 if self.mode == "centre" else 4
        if coords.shape[-1] < need:
            raise ValueError(f"GridRandomPositionalEmbeddings({self.mode}): coords need {need} columns")
        if input_coord_space == "img":
            B, S = coords.shape[:2]
            flat = coords.reshape(B * S, coords.shape[-1])
            add = None if addend is None else addend.reshape(B * S, -1)
            if self.mode == "centre":
                out = hip_ops.pos_grid_sample(table, gh, gw, flat, hip_ops.POS_CENTRE_IMG, gh, gw, rows_per_image=S, addend=add)
            else:
                out = hip_ops.pos_grid_sample(table, gh, gw, flat, hip_ops.POS_ROI, 1.0 / self.patch_size, addend=add)  # :128
            return out.view(B, S, -1)
        if self.mode == "centre":
            return hip_ops.pos_grid_sample(table, gh, gw, coords, hip_ops.POS_CENTRE_OBJ, fh * factor, fw * factor,
                                           addend=addend)                                                            # :104-109
        return hip_ops.pos_grid_sample(table, gh, gw, coords, hip_ops.POS_ROI, 1.0 / (self.patch_size * factor),
                                       addend=addend)                                                                # :144


def _mlp_hip(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(Linear, LeakyReLU, ..., Linear) through ocv_linear_fwd."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    y = x.contiguous()
    for i, m in enumerate(lin):
        y = hip_ops.linear(y, m.weight.detach(), m.bias.detach(),
                           hip_ops.ACT_LEAKY_RELU if i + 1 < len(lin) else hip_ops.ACT_NONE)
    return y


# ---------------------------------------------------------------------------
# self-attention + cross-attention block
# ---------------------------------------------------------------------------
class SelfAttnCrossAttn(nn.Module):
    def __init__(self, args, embedding_dim=128, num_heads=4, dim_feedforward=1024):
        super().__init__()
        self.args = args
        self.image_encoder_layers = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=dim_feedforward, batch_first=True)
        self.image_transformer_encoder = nn.TransformerEncoder(self.image_encoder_layers, num_layers=4, enable_nested_tensor=False)
        self._img_stack = HipEncoderStack(self.image_transformer_encoder)
        self._obj_stack = None
        if self.args.graphbins.objcavit.get("no_obj_sa") != True:  # noqa: E712  (config value may be None)
            self.obj_encoder_layers = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=dim_feedforward, batch_first=True)
            self.obj_transformer_encoder = nn.TransformerEncoder(self.obj_encoder_layers, num_layers=4, enable_nested_tensor=False)
            self._obj_stack = HipEncoderStack(self.obj_transformer_encoder)
        self.cross_attn_obj_im = nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=4, batch_first=True)
        self.cross_attn_im_obj = nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=4, batch_first=True)
        self._ca1_p3, self._ca2_p3 = {}, {}          # packed three-term-split projection weights, keyed on the weights' versions

    @staticmethod
    def _pad_objects(object_features, device, pad_to: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """list of N_i x E (or a B x S x E tensor, iterated over its batch dim) ->
        (B x Nmax x E padded with 1e-4, B x Nmax bool mask, True = padding)  (reference :180-183).
        ``pad_to``: pad to that many rows instead of the largest count of THIS batch -- what the reference computes for
        these images inside a larger batch whose longest object list has ``pad_to`` entries (with
        ``use_2_saca`` an image's result depends on the batch's Nmax; a data-parallel shard passes the global Nmax so
        that it reproduces the single-process batch, objcavit_amd/dp.py)."""
        if isinstance(object_features, torch.Tensor):
            B, N = object_features.shape[:2]
            if pad_to is not None and int(pad_to) != N:
                raise ValueError("pad_to applies to object LISTS; a B x S x E tensor (saca_2's input) is never padded")
            return object_features.contiguous(), torch.zeros(B, N, dtype=torch.bool, device=device)
        counts = [int(o.shape[0]) for o in object_features]
        nmax = max(counts)
        if pad_to is not None:
            if int(pad_to) < nmax:
                raise ValueError(f"pad_to = {pad_to} is smaller than the longest object list of the batch ({nmax})")
            nmax = int(pad_to)
        if all(c == nmax for c in counts):
            return torch.stack(list(object_features), dim=0), torch.zeros(len(counts), nmax, dtype=torch.bool, device=device)
        feats = nn.utils.rnn.pad_sequence(list(object_features), batch_first=True, padding_value=PAD_VALUE)
        if feats.shape[1] < nmax:
            feats = F.pad(feats, (0, 0, 0, nmax - feats.shape[1]), value=PAD_VALUE)
        return feats, _RAGGED_MASK.get(device, tuple(counts), nmax)     # True = padding (:180-181)

    def object_self_attention(self, object_features, device, pad_to: Optional[int] = None, counts: Optional[torch.Tensor] = None):
        """The object half of ``forward`` -- pad + mask, self-attention stack -- which does not depend on the image tokens:
        (att_obj [B, cap, E], mask [B, cap], counts or None).  May be issued ahead of time on another stream.
        ``object_features``: the reference's list of N_i x E tensors; or a [B, cap, E] tensor with ``counts`` (int32 [B] on the
        device; rows >= counts[b] arbitrary): the same thing shape-static, nothing read on the host; or a B x S x E tensor
        without counts (what saca_2 receives, SURVEY.md Q3: every row counts as an object)."""
        if not isinstance(object_features, torch.Tensor):
            lens = [int(o.shape[0]) for o in object_features]
            cap = max(lens)
            if pad_to is not None:
                if int(pad_to) < cap:
                    raise ValueError(f"pad_to = {pad_to} is smaller than the longest object list of the batch ({cap})")
                cap = int(pad_to)
            object_features = self._pad_objects(object_features, device, cap)[0]
            counts = _COUNTS.get(device, tuple(lens))
        if counts is None:
            feats, mask = self._pad_objects(object_features, device, pad_to)               # tensor input: nothing is padding
        else:
            feats, mask = hip_ops.object_tokens_pad(object_features.contiguous(), counts, PAD_VALUE)    # :180-183
        if self._obj_stack is None:
            return feats, mask, counts                                                    # :186
        return self._obj_stack(feats, mask), mask, counts                                 # :188 (padded rows -> 0)

    def forward(self, image_patch_embeddings, object_features, want_object_output: bool = True, pre_obj=None,
                pad_objects_to: Optional[int] = None, counts: Optional[torch.Tensor] = None, group: Optional[int] = None,
                pre_join=None):
        """``counts`` / ``group``: see ``object_self_attention`` and hip_ops.object_front_pad -- ``group`` consecutive images
        form one call of the reference (their longest list is the Nmax the key rows are front-padded to); None = the batch.
        ``pre_obj`` / ``pre_join``: the object half already issued by the caller, on the stream ``pre_join`` (joined here, behind
        the image tokens' stack)."""
        x = image_patch_embeddings.contiguous()
        B, S, E = x.shape
        main = torch.cuda.current_stream(x.device) if x.is_cuda else None
        if pre_obj is None and x.is_cuda and self._obj_stack is not None and hip_ops.token_overlap_enabled():
            # the object tokens' self-attention stack beside the image tokens': two independent chains of small launches
            side = hip_ops.side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pre_obj = self.object_self_attention(object_features, x.device, pad_objects_to, counts)
            pre_join = side
        att_img = self._img_stack(x)                                                      # reference :169
        if pre_join is not None:
            main.wait_stream(pre_join)
            for t in pre_obj:
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        att_obj, mask, counts = pre_obj if pre_obj is not None else \
            self.object_self_attention(object_features, x.device, pad_objects_to, counts)
        cap = att_obj.shape[1]
        amt = S - cap                                                                     # :192
        if amt < 0:
            raise ValueError(f"more objects per image ({cap}) than image tokens ({S})")
        if counts is not None:
            # mask padded at the BACK, rows padded at the FRONT to the longest list of the (group's) batch -- on the device
            att_obj_p, kpm = hip_ops.object_front_pad(att_obj, counts, S, PAD_VALUE, group=group,
                                                      nmax=int(pad_objects_to or 0))     # :193-194
        elif amt == 0:
            att_obj_p, kpm = att_obj.contiguous(), mask
        else:
            kpm = F.pad(mask, (0, amt), value=True)                                       # :193  mask padded at the BACK
            att_obj_p = F.pad(att_obj, (0, 0, amt, 0), value=PAD_VALUE).contiguous()      # :194  rows padded at the FRONT
        ca1 = self.cross_attn_obj_im
        # every key at position >= cap is masked (the mask is True beyond each image's object count): the kernel only projects /
        # scores the first cap keys -- same result, ~10x less work
        final_img = hip_ops.mha(att_img, att_obj_p, att_img, ca1.in_proj_weight.detach(), ca1.in_proj_bias.detach(),
                                ca1.out_proj.weight.detach(), ca1.out_proj.bias.detach(), kpm, ca1.num_heads,
                                kv_limit=int(cap), packed=self._ca1_p3)                                       # :195-201
        final_obj = None
        if want_object_output:
            ca2 = self.cross_attn_im_obj
            final_obj = hip_ops.mha(att_obj_p, att_img, att_obj_p, ca2.in_proj_weight.detach(), ca2.in_proj_bias.detach(),
                                    ca2.out_proj.weight.detach(), ca2.out_proj.bias.detach(), None, ca2.num_heads,
                                    packed=self._ca2_p3)                                                      # :202-207
        return final_img, final_obj


# ---------------------------------------------------------------------------
# ObjCAViT
# ---------------------------------------------------------------------------
_MLP_IN = {"learned": 2, "learned_bbox_wh": 4}


class ObjCAViT(nn.Module):
    def __init__(self, args, im_feature_dim=128, obj_feature_dim=512, n_query_channels=128, patch_size=16, dim_out=256,
                 embedding_dim=128, num_heads=4, norm='linear', max_seq_len=50):
        super().__init__()
        self.args = args
        self.norm = norm
        self.n_query_channels = n_query_channels
        self.patch_size = patch_size
        self.half_patch_size = self.patch_size // 2
        self.obj_feature_dim = obj_feature_dim
        self.strategy = self.args[self.args.model.name].objcavit.positional_embedding_strategy

        if self.strategy == "grid_random":
            self.positional_encoder = GridRandomPositionalEmbeddings(args, embedding_dim=embedding_dim, patch_size=patch_size, mode="centre")
        elif self.strategy == "grid_random_roi_align":
            self.positional_encoder = GridRandomPositionalEmbeddings(args, embedding_dim=embedding_dim, patch_size=patch_size, mode="roi_align")
        elif self.strategy in _MLP_IN:
            widths = (_MLP_IN[self.strategy], 32, 64, 128, 256, embedding_dim)
            layers: List[nn.Module] = []
            for i in range(5):
                layers.append(nn.Linear(widths[i], widths[i + 1], bias=True))
                if i < 4:
                    layers.append(nn.LeakyReLU())
            self.positional_encoder = nn.Sequential(*layers)
        else:
            sys.exit("Error: ObjCAViT positional embedding strategy not recognised.")

        self.image_embedding_convPxP = nn.Conv2d(im_feature_dim, embedding_dim, kernel_size=self.patch_size, stride=patch_size, padding=0)
        self.obj_embedding_layer = nn.Linear(self.obj_feature_dim, embedding_dim)
        self.saca_1 = SelfAttnCrossAttn(self.args, embedding_dim, num_heads, dim_feedforward=1024)
        self.use_2_saca = self.args.graphbins.objcavit.get("use_2_saca") == True  # noqa: E712
        if self.use_2_saca:
            self.saca_2 = SelfAttnCrossAttn(self.args, embedding_dim, num_heads, dim_feedforward=1024)
        self.dot_product_layer = PixelWiseDotProduct()
        self.conv3x3 = nn.Conv2d(im_feature_dim, embedding_dim, kernel_size=3, stride=1, padding=1)
        self.regressor = nn.Sequential(nn.Linear(embedding_dim, 256), nn.LeakyReLU(),
                                       nn.Linear(256, 256), nn.LeakyReLU(),
                                       nn.Linear(256, dim_out))
        self._img_pos_cache = {}
        self._w_cl = hip_ops.ChannelsLastWeight()
        self._w_pe = hip_ops.PatchEmbedSplitWeight()

    # -- positional embeddings ------------------------------------------------
    def _patch_coords(self, B: int, gh: int, gw: int, device) -> torch.Tensor:
        """B x S x 4 patch centres (feature-map pixels) and sizes (reference :336-347)."""
        xs = torch.arange(gw, device=device).view(1, -1).expand(gh, -1)
        ys = torch.arange(gh, device=device).view(-1, 1).expand(-1, gw)
        pc = (torch.stack([xs, ys], dim=0) * self.patch_size + self.half_patch_size).flatten(1)
        pc = pc.expand(B, -1, -1).permute(0, 2, 1).float()
        return torch.cat([pc, torch.ones_like(pc) * self.patch_size], dim=2)

    def _object_pos(self, xywh: torch.Tensor, image_features: torch.Tensor, addend: torch.Tensor) -> torch.Tensor:
        """addend + positional embedding of every object row of the batch, one launch (reference :316-330)."""
        if self.strategy.startswith("grid_random"):
            return self.positional_encoder(xywh, image_features, "obj", addend=addend)
        return _mlp_hip(self.positional_encoder, xywh[:, 0:_MLP_IN[self.strategy]]) + addend

    def _image_pos(self, image_features: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
        """Positional embedding of the image tokens, [S, E] shared by the batch: it depends only on (gh, gw) and the
        weights -- the patch coordinates are the same for every image (:336-347) -- so in eval it is computed once
        per parameter version and cached."""
        params = [p for p in self.positional_encoder.parameters()]
        key = (gh, gw, tuple(image_features.shape[2:]), image_features.device, tuple(p.data_ptr() for p in params),
               tuple(p._version for p in params))
        if self._img_pos_cache.get("k") != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("image positional embedding not cached yet: run one eager warm-up call before capture")
            pc = self._patch_coords(1, gh, gw, image_features.device)
            if self.strategy in _MLP_IN:
                v = _mlp_hip(self.positional_encoder, pc[0, :, 0:_MLP_IN[self.strategy]])
            else:
                v = self.positional_encoder(pc, image_features[:1], "img")[0]
            self._img_pos_cache = {"k": key, "v": v.contiguous()}
        return self._img_pos_cache["v"]

    # -- forward ---------------------------------------------------------------
    def _embed_objects(self, po: PaddedObjects, image_features) -> torch.Tensor:
        """Linear(512 -> E) + positional embedding of every object row of the batch as ONE flattened [B * cap, .] problem
        (reference :311-330: a Python loop over images) -> [B, cap, E]; rows beyond an image's count hold garbage (possibly
        NaN: a zero box under roi_align) that the padding kernel replaces.  ``image_features`` is only read (for its shape)
        by the grid strategies."""
        B, cap = po.features.shape[:2]
        need = 2 if self.strategy in ("learned", "grid_random") else 4
        if po.xywh.shape[2] < need:
            raise ValueError(f"object boxes have fewer than {need} columns")
        emb = hip_ops.linear(po.features.reshape(B * cap, -1), self.obj_embedding_layer.weight.detach(),
                             self.obj_embedding_layer.bias.detach())
        emb = self._object_pos(po.xywh.reshape(B * cap, -1), image_features, emb)
        return emb.view(B, cap, -1)

    def can_prepass(self) -> bool:
        """Whether the object branch (embedding + first self-attention stack) is independent of the image features --
        true for the MLP positional strategies -- so that it can run beside the encoder on a second stream."""
        return self.strategy in _MLP_IN and not self.training

    def _padded(self, object_features, object_xywh_list, dev, capacity: Optional[int]) -> PaddedObjects:
        if isinstance(object_features, PaddedObjects):
            if capacity is not None and object_features.capacity < int(capacity):
                raise ValueError(f"PaddedObjects of capacity {object_features.capacity} where {capacity} rows were asked for")
            return object_features
        return PaddedObjects.from_lists(object_features, object_xywh_list, dev, capacity)

    def object_prepass(self, object_features, object_xywh_list, dev, pad_objects_to: Optional[int] = None):
        """Steps of ``forward_parts`` that do not need the dense features: (padded inputs, embedded objects, (att_obj, mask, counts))."""
        po = self._padded(object_features, object_xywh_list, dev, pad_objects_to)
        emb = self._embed_objects(po, None)
        return po, emb, self.saca_1.object_self_attention(emb, dev, None, po.counts)

    def forward_parts(self, image_features, object_features, object_xywh_list, pre=None, pad_objects_to: Optional[int] = None,
                      object_group: Optional[int] = None):
        """-> (bin_widths_normed, conv3x3 features, queries view B x n_query x E).
        ``object_features`` / ``object_xywh_list``: the reference's two lists, or a ``PaddedObjects`` (then the second argument is
        ignored).  ``pre``: result of ``object_prepass`` when the caller has already issued the object branch.
        ``pad_objects_to``: the Nmax of the batch when this batch is a shard of a larger one (SelfAttnCrossAttn._pad_objects).
        ``object_group``: images per reference call when several calls' batches run as one (hip_ops.object_front_pad)."""
        if self.training:
            raise RuntimeError("the HIP path implements inference (eval mode) only")
        dev = image_features.device
        B = image_features.shape[0]
        # 1. objects: Linear(512 -> E) + positional embedding, all images in one launch (reference :311-330)
        side = None
        if pre is None and image_features.is_cuda and hip_ops.token_overlap_enabled():
            # the whole object branch -- embedding, positional term, padding, object self-attention -- on the side stream, beside
            # the patch embedding and the image tokens' stack (joined inside saca_1, in front of the cross-attention)
            main = torch.cuda.current_stream(dev)
            side = hip_ops.side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                po = self._padded(object_features, object_xywh_list, dev, pad_objects_to)
                emb = self._embed_objects(po, image_features)
                pre_obj = self.saca_1.object_self_attention(emb, dev, None, po.counts)
            for t in (po.features, po.xywh, po.counts, emb):
                t.record_stream(main)
        elif pre is None:
            po = self._padded(object_features, object_xywh_list, dev, pad_objects_to)
            emb = self._embed_objects(po, image_features)
            pre_obj = None
        else:
            po, emb, pre_obj = pre
        if po.features.shape[0] != B:
            raise ValueError("object_features / object_xywh_list must have one entry per image")
        if pad_objects_to is not None and int(pad_objects_to) > po.capacity:
            raise ValueError(f"pad_objects_to = {pad_objects_to} exceeds the object capacity {po.capacity}")
        if pad_objects_to is not None and po.max_count is not None and int(pad_objects_to) < po.max_count:
            raise ValueError(f"pad_to = {pad_objects_to} is smaller than the longest object list of the batch ({po.max_count})")
        if isinstance(object_features, list):
            # the reference overwrites the caller's list with the embedded objects (:330)
            for i in range(min(B, len(object_features))):
                object_features[i] = emb[i, :int(object_features[i].shape[0])]

        # 2. image tokens: patch conv + bias + positional embedding, token-major (reference :333-364)
        if self.patch_size != 16:
            raise NotImplementedError("the patch-embedding kernel is built for 16x16 patches")
        gh, gw = image_features.shape[2] // 16, image_features.shape[3] // 16
        if gh * gw < self.n_query_channels + 1:
            raise ValueError(f"need at least {self.n_query_channels + 1} patches, got {gh * gw}")
        ds = self.args[self.args.basic.dataset]

        def tokens():
            tok = hip_ops.patch_embed_auto(image_features, self.image_embedding_convPxP.weight.detach(),
                                           self.image_embedding_convPxP.bias.detach(), self._image_pos(image_features, gh, gw),
                                           self._w_cl, self._w_pe)
            # 3. self-attention / cross-attention stacks (reference :366-368)
            tok, obj = self.saca_1(tok, emb, want_object_output=self.use_2_saca, pre_obj=pre_obj, pad_objects_to=pad_objects_to,
                                   counts=po.counts, group=object_group, pre_join=side)
            if self.use_2_saca:
                tok, obj = self.saca_2(tok, obj, want_object_output=False)
            # 4. heads (reference :373-388): the bin regressor on token 0
            return tok, regress_bin_widths(self.regressor, tok[:, 0, :], self.norm, (ds.min_depth, ds.max_depth))

        if image_features.is_cuda and side is None and hip_ops.head_overlap_enabled():
            # the heads' 3x3 convolution does not read the tokens: it fills the chip on this stream while the token chain's small
            # launches run on a second side stream.  Never beside an object branch forked above (three parallel branches replay
            # pathologically slowly from a hipGraph: hip_ops.head_overlap_enabled): only when that branch was issued by the caller
            # (``pre``, beside the encoder) or runs in line
            main = torch.cuda.current_stream(dev)
            tst = hip_ops.side_stream(dev, 1)
            tst.wait_stream(main)
            with torch.cuda.stream(tst), hip_ops.single_chain():       # (a second SA/CA stack forks nothing in here)
                tok, y = tokens()
            with hip_ops.islands_suspended():                          # (a capture cannot be cut while the fork is open)
                feat = self._conv3x3_nhwc(image_features)
            main.wait_stream(tst)
            for t in (tok, y):
                t.record_stream(main)
        else:
            tok, y = tokens()
            feat = self._conv3x3_nhwc(image_features)
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

    def forward(self, image_features, object_features, object_xywh_list, pad_objects_to: Optional[int] = None):
        y, feat, queries = self.forward_parts(image_features, object_features, object_xywh_list, pad_objects_to=pad_objects_to)
        return y, self.dot_product_layer(feat, queries)
