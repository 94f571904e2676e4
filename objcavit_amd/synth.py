"""Seeded synthetic weights / inputs (there is no checkpoint offline): what the golden vectors, the parity tests, the
benchmark and ``smoke()`` all load.

Values come from ``numpy.random.RandomState`` seeded per key with a CRC of the
key name, so they are stable across numpy/torch versions and independent of
which other keys exist.  Used by tests/golden/make_golden.py (reference side, in
the build container; under the name ``gen``), by the tests (oracle / HIP side,
anywhere) and by bench.py.  No dependency on the reference, none on the oracle.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch


def _rs(seed: int, key: str) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def tensor_for(key: str, shape: Sequence[int], seed: int, gains: Sequence[Tuple[str, float]] = ()) -> torch.Tensor:
    """Value for one state_dict entry, by name suffix."""
    rs = _rs(seed, key)
    shape = tuple(int(s) for s in shape)
    gain = 1.0
    for frag, g in gains:
        if frag in key:
            gain = g
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if key.endswith("running_mean"):
        a = rs.standard_normal(shape) * 0.1
    elif key.endswith("running_var"):
        a = rs.uniform(0.5, 1.5, shape)
    elif key.endswith("positional_encodings"):
        a = rs.uniform(0.0, 1.0, shape) * gain
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = rs.standard_normal(shape) * (gain / np.sqrt(fan_in))
    elif key.endswith("weight"):            # norm scales
        a = 1.0 + 0.1 * rs.standard_normal(shape)
    else:                                   # biases
        a = 0.1 * gain * rs.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def fill(shapes: Mapping[str, Sequence[int]], seed: int, gains: Sequence[Tuple[str, float]] = ()) -> Dict[str, torch.Tensor]:
    return {k: tensor_for(k, s, seed, gains) for k, s in shapes.items()}


def shapes_of(module: torch.nn.Module) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_into(module: torch.nn.Module, seed: int, gains: Sequence[Tuple[str, float]] = ()) -> Dict[str, torch.Tensor]:
    sd = fill(shapes_of(module), seed, gains)
    module.load_state_dict(sd, strict=True)
    return sd


def randn(key: str, shape: Sequence[int], seed: int, scale: float = 1.0) -> torch.Tensor:
    a = _rs(seed, "input:" + key).standard_normal(tuple(shape)) * scale
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def uniform(key: str, shape: Sequence[int], seed: int, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    a = _rs(seed, "input:" + key).uniform(lo, hi, tuple(shape))
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def boxes(key: str, n: int, seed: int, H: int, W: int) -> torch.Tensor:
    """xywh boxes in full-resolution pixels (modules/Yolov7Wrapper.py:118-123 convention)."""
    rs = _rs(seed, "boxes:" + key)
    cx, cy = rs.uniform(0, W, n), rs.uniform(0, H, n)
    w, h = rs.uniform(8, W / 2, n), rs.uniform(8, H / 2, n)
    return torch.from_numpy(np.stack([cx, cy, w, h], 1).astype(np.float32))


def sample_pixels(n_pix: int, count: int, seed: int) -> np.ndarray:
    return np.sort(_rs(seed, "pixels").choice(n_pix, size=count, replace=False)).astype(np.int64)


# Weight gains that make attention and the 256-way bin softmax non-trivial
# (default init gives an almost constant depth map: SURVEY Q12).
PEAKY = (("in_proj_weight", 2.0), ("conv_out", 6.0), ("conv3x3", 1.5), ("regressor.4", 3.0))


VALIDATION_CASES = {
    # tag: (dataset, garg_crop, eigen_crop, (H, W), (h, w), min_depth, max_depth, B, seed)
    "nyu_eigen": ("nyu", False, True, (480, 640), (240, 320), 0.001, 10.0, 3, 61),
    "nyu_nocrop": ("nyu", False, False, (96, 130), (48, 65), 0.001, 10.0, 2, 62),
    "kitti_garg": ("kitti", True, False, (352, 1216), (176, 608), 0.001, 80.0, 2, 63),
    "kitti_eigen": ("kitti", False, True, (352, 1216), (176, 608), 0.001, 80.0, 2, 64),
}


def validation_inputs(tag):
    """(gt [B,1,H,W], pred [B,1,h,w], pred_mirror [B,1,h,w]) of a G6 case: ground truth with invalid pixels on both sides
    of the range, predictions partly outside it, and a NaN, a +inf and a -inf planted in the first prediction."""
    import torch
    dataset, garg, eigen, (H, W), (h, w), dmin, dmax, B, seed = VALIDATION_CASES[tag]
    rs = np.random.RandomState(seed)
    gt = torch.from_numpy(rs.uniform(-0.2 * dmax, 1.1 * dmax, (B, 1, H, W)).astype(np.float32))
    pa = torch.from_numpy(rs.uniform(0.2, 1.2 * dmax, (B, 1, h, w)).astype(np.float32))
    pb = torch.from_numpy(rs.uniform(0.2, 1.2 * dmax, (B, 1, h, w)).astype(np.float32))
    pa[0, 0, 3, 5], pa[0, 0, 7, 1], pa[0, 0, 9, 9] = float("nan"), float("inf"), float("-inf")
    return gt, pa, pb
