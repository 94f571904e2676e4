"""objcavit_amd -- MI355X-native forward depth-inference path of ObjCAViT.

Layout
------
csrc/       hand-written HIP kernels for gfx950 + the C-ABI (include/objcavit_hip.h)
_lib.py     ctypes loader for lib/libobjcavit_hip.so (fails loudly if absent)
hip_ops.py  tensor-level wrappers over the C-ABI (pointers + sizes + stream)
modules/    drop-in nn.Module classes with the reference's names and signatures
config.py   the ``args`` tree the modules read
dp.py       data-parallel sharding + the single metric all-gather

Importing this package does not load the HIP library; the first kernel call
does, and raises if the library or a GPU is missing (there is no CPU fallback).
"""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: the first two streams a process creates get
# queues of their own, later ones share the last) and streams that share a queue run one after the other.  Several
# GraphedGraphBins instances in flight (one stream each, bench.py --inflight) need queues of their own: measured 781 instead
# of 840 img/s on shared queues.  A library must not change runtime-wide policy behind its host's back, so importing this
# package touches the environment ONLY when the host asks for it (OCV_SET_HW_QUEUES=<n>, effective when set before the
# HIP runtime initialises); bench.py sets GPU_MAX_HW_QUEUES itself, INTEGRATION.md holds the recommendation.
if _os.environ.get("OCV_SET_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(_os.environ["OCV_SET_HW_QUEUES"])))

__version__ = "0.4.0"
