"""Shared helpers for the tests (fixtures loading, comparisons)."""
import json
import os

import numpy as np
import torch

import gen  # tests/golden/gen.py (on sys.path via conftest)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def gains_of(meta):
    return [tuple(g) for g in meta.get("gains", [])]


def rel_dev(a, b):
    """max |a-b| / max |b| (both converted to float64 CPU tensors)."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def max_rel(a, b):
    """max elementwise |a-b| / |b| (for strictly positive b such as depth)."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float(((a - b).abs() / b.abs()).max())


def state_dict_from(meta_shapes, seed, gains=()):
    return gen.fill({k: tuple(v) for k, v in meta_shapes.items()}, seed, gains)
