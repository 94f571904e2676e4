"""Object front-end formats (row N4 of SURVEY.md section 8f): what the frozen detector + phrase builder + CLIP hand to
the hot path, and a table-driven provider that replaces the last two with one gather.

Formats (modules/Yolov7Wrapper.py:45-150, modules/GraphBins.py:90-107), per image of a batch, all optional (None = no
detection in that image):
  xywh  [N, 4] fp32   bounding boxes, centre x, centre y, width, height in full-resolution pixels
  cls   [N]    int    class index of the detector (LVIS: 0..1203)
  names N strings     class labels (LVIS: WordNet synsets "chair.n.01")
The reference turns (names, boxes) into a phrase per object (modules/ObjectLanguageStrategy.py:128-179) and the phrase
into a 512-d CLIP text feature (modules/CLIPWrapper.py:21-24, un-normalised, cast .float() at GraphBins.py:106).  For
the strategies whose phrase depends on the class only ("none", "synset_def_wn") that is a fixed [n_classes, 512] table;
for "name_synset_def_wn_rel_sz" the phrase also names the next object and one of 7 size relations, so the table key is
(class, next class, relation) -- a phrase cache filled by whoever owns CLIP.  ``TableObjectProvider`` is the device
side of both: detections in, ``(features_list, xywh_list, None)`` out -- the ``object_provider`` of ``GraphBins`` --
with no CLIP in the loop.

Parity: the relation index below restates ObjectLanguageStrategy.py:69-81 and is PINNED: tests/golden/make_golden.py g7
runs the reference's own get_single_relative_size_clause (nltk, imported at the module top but unused by that method,
replaced by an empty stand-in module) on boxes that include equal areas, both ends of the scale and the half-way points
of the rounding; tests/golden/g7_relsize.npz holds its clauses and relation indices.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

REL_SIZE_SCALE = ("much smaller than", "smaller than", "a bit smaller than", "about the same size as",
                  "a bit bigger than", "bigger than", "much bigger than")          # ObjectLanguageStrategy.py:23-31


def relative_size_index(xywh: torch.Tensor) -> List[int]:
    """Index into REL_SIZE_SCALE for every object of one image against the NEXT object of the list (cyclic),
    ObjectLanguageStrategy.py:69-81: ratio of box areas -> log -> [1/e, e] mapped onto the 5 middle entries, rounded
    (numpy's round-half-to-even), clipped to the 7-entry scale.  Empty for fewer than 2 objects (no clause is made)."""
    n = 0 if xywh is None else int(xywh.shape[0])
    if n <= 1:
        return []
    # the reference multiplies and divides 0-d fp32 TENSORS (obj_xywh[2] * obj_xywh[3], area / next_area) and only then
    # leaves torch (math.log of the fp32 quotient): areas and the ratio are rounded to fp32, the rest is double
    b = xywh.detach().to("cpu", torch.float32)
    area = b[:, 2] * b[:, 3]
    ratio = (area / torch.roll(area, -1)).tolist()
    out = []
    L = len(REL_SIZE_SCALE)
    for j in range(n):
        f = (math.log(ratio[j]) + 1.0) / 2.0 * (L - 3)
        r = round(f) + 1                      # Python's round == numpy's: half to even
        out.append(int(min(max(r, 0), L - 1)))
    return out


class TableObjectProvider:
    """``object_provider`` for ``GraphBins`` backed by precomputed text features.

    ``detector(image) -> (xywh_list, cls_list)`` supplies the detections in the formats above (lists of length B,
    entries None where nothing was found).  ``class_table`` is [n_classes, 512] (phrase depends on the class only);
    alternatively ``phrase_features(cls_i, cls_next, relation) -> 512-d tensor`` serves the relative-size strategy
    from its cache.  An image without detections gets ONE zero feature and no box, as the reference does
    (LanguageEmbeddingWrapper.py:56-61 with cls = [0]; ObjCAViT.py handles xywh None)."""

    def __init__(self, detector: Callable, class_table: Optional[torch.Tensor] = None,
                 phrase_features: Optional[Callable[[int, int, int], torch.Tensor]] = None, dim: int = 512):
        if (class_table is None) == (phrase_features is None):
            raise ValueError("TableObjectProvider: give exactly one of class_table / phrase_features")
        if class_table is not None and (class_table.dim() != 2 or class_table.shape[1] != dim):
            raise ValueError(f"TableObjectProvider: class_table must be [n_classes, {dim}]")
        self.detector, self.table, self.phrase_features, self.dim = detector, class_table, phrase_features, dim

    def __call__(self, image: torch.Tensor):
        xywh_list, cls_list = self.detector(image)
        if len(xywh_list) != image.shape[0] or len(cls_list) != image.shape[0]:
            raise ValueError("TableObjectProvider: the detector must return one entry per image")
        dev = image.device
        feats: List[torch.Tensor] = []
        boxes: List[Optional[torch.Tensor]] = []
        for xywh, cls in zip(xywh_list, cls_list):
            if cls is None or len(cls) == 0:
                feats.append(torch.zeros(1, self.dim, dtype=torch.float32, device=dev))
                boxes.append(None)
                continue
            cls = torch.as_tensor(cls, device=dev).long()
            if xywh is None or xywh.shape != (cls.shape[0], 4):
                raise ValueError("TableObjectProvider: xywh must be [N, 4] for N classes")
            if self.table is not None:
                if int(cls.min()) < 0 or int(cls.max()) >= self.table.shape[0]:
                    raise ValueError("TableObjectProvider: class index outside the table")
                f = self.table.to(dev).index_select(0, cls).float()
            else:
                rel = relative_size_index(xywh)
                c = cls.tolist()
                n = len(c)
                f = torch.stack([self.phrase_features(c[j], c[(j + 1) % n], rel[j] if rel else -1).to(dev).float()
                                 for j in range(n)], 0)
            feats.append(f)
            boxes.append(xywh.to(dev).float())
        return feats, boxes, None
