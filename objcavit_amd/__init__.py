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
__version__ = "0.1.0"
