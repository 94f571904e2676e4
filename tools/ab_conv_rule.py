"""(needs tools/diag/conv_direct.patch.txt applied)  The whole benchmark with the pre-split convolution kernel's dispatch rule as shipped (1: direct form where 2 - 3 channel tiles per
workgroup are modelled to pay) or switched off (0: the parked epilogue everywhere, rounds 2 - 5).  `python tools/ab_conv_rule.py 0|1
[bench args]`; alternate on one box (boxes differ by +- 2.5 %).  Record: profiles/r06_conv_direct.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
on = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench                                   # (sets its environment before torch loads)
from objcavit_amd import _lib
assert _lib.load().ocv_conv_split_set_dispatch(on, 0) == 0
bench.main()
