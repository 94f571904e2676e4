"""Hand-computed PS-RoI-align (output 1x1, adaptive sampling) cases on a 3 x 4 grid of two channels that are LINEAR in
the cell index: channel 0 = x, channel 1 = 10 * y.  Bilinear interpolation reproduces a linear function exactly between
the first and last cell centre, so the expected values below follow from the sample positions alone -- they were worked
out by hand from the published algorithm (corner * scale - 0.5; n = ceil(extent) samples per axis at
start + (i + .5) * extent / n; samples <= 0 read cell 0, samples in (size - 1, size] read the last cell, samples
outside [-1, size] contribute 0 but still count in the divisor) and are independent of oracle/restate.py, which the
CPU suite checks against them, and of csrc/pos_sample.hip, which the GPU suite checks against them.

Boxes are (cx, cy, w, h) in full-resolution pixels, spatial_scale = 1/32 (patch 16 x factor 2).
"""
import numpy as np
import torch

GH, GW, SCALE = 3, 4, 1.0 / 32.0


def grid_table() -> torch.Tensor:
    """[gh*gw, 2] table whose row y*gw + x holds (x, 10*y)."""
    ys, xs = np.meshgrid(np.arange(GH), np.arange(GW), indexing="ij")
    return torch.from_numpy(np.stack([xs, 10.0 * ys], -1).reshape(GH * GW, 2).astype(np.float32))


# (name, box xywh, expected [ch0, ch1])
CASES = [
    # x: 72..88 -> 2.25..2.75 -> 1.75..2.25, extent .5, 1 sample at 2.0 ; y: 40..56 -> .75..1.25, 1 sample at 1.0
    ("inside one cell", (80.0, 48.0, 16.0, 16.0), (2.0, 10.0)),
    # x: 16..112 -> 0..3, 3 samples at .5, 1.5, 2.5 (mean 1.5) ; y: 16..80 -> 0..2, 2 samples at .5, 1.5 (mean 1.0)
    ("spanning cells", (64.0, 48.0, 96.0, 64.0), (1.5, 10.0)),
    # x: -32 -> clamped to 0, ..48 -> -0.5..1.0, extent 1.5, 2 samples at -0.125 (reads cell 0 = 0) and 0.625
    ("clipped at zero", (8.0, 48.0, 80.0, 16.0), (0.3125, 10.0)),
    # x: 80..160 -> 2.0..4.5, extent 2.5, 3 samples at 2.41667, 3.25 (last cell = 3), 4.08333 (> 4: contributes 0)
    ("beyond the grid", (120.0, 48.0, 80.0, 16.0), ((2.0 + 2.5 / 6 + 3.0 + 0.0) / 3.0, 20.0 / 3.0)),
    # x: 0..3264 -> -0.5..101.5, 102 samples at 0, 1, ..., 101: cells 0 1 2 3 3 then nothing -> 9 / 102 ; y one sample at 1.0
    ("adaptive count, mostly outside", (64.0, 48.0, 6400.0, 16.0), (9.0 / 102.0, 50.0 / 102.0)),
    # y: 0..96 -> -0.5..2.5, 3 samples at 0, 1, 2 -> mean 1.0 ; x: 100..108 -> 2.625..2.875, one sample at 2.75
    ("exactly the grid height", (104.0, 48.0, 8.0, 96.0), (2.75, 10.0)),
]
NAN_BOXES = [(50.0, 50.0, 0.0, 0.0), (-1.0, -1.0, -1.0, -1.0)]      # no extent after clamping: 0 / 0 (reference :313 box)


def vectorised_expected(table: np.ndarray, gh: int, gw: int, boxes: np.ndarray, scale: float) -> np.ndarray:
    """Independent float64 formulation for random boxes: per axis, the (n_samples x size) matrix of bilinear weights,
    then out = sum_y sum_x Wy^T T Wx / (ny * nx) -- separable, no sample loop in common with the oracle or the kernel."""
    T = table[: gh * gw].reshape(gh, gw, -1).astype(np.float64)
    out = np.zeros((boxes.shape[0], T.shape[2]))

    def axis_weights(lo_px, hi_px, size):
        a = np.float32(max(lo_px, 0.0)) * np.float32(scale) - np.float32(0.5)
        b = np.float32(max(hi_px, 0.0)) * np.float32(scale) - np.float32(0.5)
        ext = np.float32(b - a)
        n = int(np.ceil(ext))
        if n <= 0:
            return None, 0
        pos = (a + (np.arange(n, dtype=np.float32) + np.float32(0.5)) * ext / np.float32(n)).astype(np.float64)
        ok = (pos >= -1.0) & (pos <= size)
        p = np.clip(pos, 0.0, size - 1.0)
        lo = np.minimum(np.floor(p).astype(int), size - 1)
        hi = np.minimum(lo + 1, size - 1)
        fr = p - lo
        Wm = np.zeros((n, size))
        Wm[np.arange(n), lo] += (1.0 - fr)
        Wm[np.arange(n), hi] += fr
        return (Wm * ok[:, None]).sum(0), n

    for i, (cx, cy, w, h) in enumerate(boxes.astype(np.float32)):
        wx, nx = axis_weights(cx - w / np.float32(2), cx + w / np.float32(2), gw)
        wy, ny = axis_weights(cy - h / np.float32(2), cy + h / np.float32(2), gh)
        if nx == 0 or ny == 0:
            out[i] = np.nan
        else:
            out[i] = np.einsum("y,yxc,x->c", wy, T, wx) / (nx * ny)
    return out
