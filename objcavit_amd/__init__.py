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
# of 840 img/s on shared queues.  Only a default, only effective when this package is imported before the HIP runtime
# initialises (first GPU call); INTEGRATION.md says so too.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.3.0"
