"""(needs tools/diag/gemm_wide.patch.txt applied: the wide kernel is not in the product -- profiles/r06_gemm_wide.txt)
bench.py with the tap GEMMs on the wide kernel (1, as shipped) or on the general one (0).  `python tools/ab_gemm_wide_bench.py 0|1 [bench args]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
on = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
from objcavit_amd import _lib
assert _lib.load().ocv_gemm_wide_set_dispatch(on) == 0
bench.main()
