"""Import the reference's own Python modules (THIS CONTAINER ONLY).

ORACLE / test infrastructure.  /root/reference does not exist on the GPU box;
nothing that runs there may import this file.  It is used by
tests/golden/make_golden.py to generate the committed golden vectors and by
the ``-m "not gpu"`` tests that are skipped when /root/reference is absent.

Two third-party imports of the reference are satisfied by stub modules placed
in ``sys.modules``:
  * ``pytorch_lightning`` -- only ``LightningModule`` is used, and only as a
    base class (modules/ObjCAViT.py:150,216) -> ``torch.nn.Module``.
  * ``torchvision``       -- imported at module top (modules/ObjCAViT.py:12,
    modules/AdaBins.py:9, modules/DenseFeatureExtractor.py:8) but only touched
    for ``ops.ps_roi_align`` in roi_align mode and for the v2 encoders; the
    stub has neither, so those code paths raise if reached.
Nothing is fetched; ``torch.hub.load`` is replaced in-process by a function
returning a caller-supplied local backbone when building the reference AdaBins.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules"))


def _install_stubs():
    import torch.nn as nn
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = nn.Module
        sys.modules["pytorch_lightning"] = pl
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.ops = types.ModuleType("torchvision.ops")
        tv.models = types.ModuleType("torchvision.models")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = tv.ops
        sys.modules["torchvision.models"] = tv.models


def load(name: str):
    """Import ``modules.<name>`` from the reference tree."""
    if not available():
        raise RuntimeError("reference tree not present")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module(f"modules.{name}")


def build_reference_adabins(args, backbone):
    """Construct the reference ``AdaBins`` class around a local backbone."""
    import torch
    mod = load("AdaBins")
    orig = torch.hub.load
    torch.hub.load = lambda *a, **k: backbone
    try:
        return mod.AdaBins(args)
    finally:
        torch.hub.load = orig
