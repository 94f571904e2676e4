"""Round 6: the decoder's skip-part convolutions over 24 and 40 encoder channels as PACKED TAPS (7 / 12 K steps of 32
channels instead of 9 / 18: csrc/conv_igemm.hip ConvArgs::gpt) -- the whole benchmark with the rule as shipped (1) or switched off (0: tap-major,
every tap padded to 32 / 64 channels, rounds 2 - 5).  `python tools/ab_skip_packed.py 0|1 [bench args]`; alternate on one box.
Record: profiles/r06_skip_packed_taps.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
on = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench                                   # (sets its environment before torch loads)
from objcavit_amd import hip_ops
if not on:
    hip_ops.packed_taps_pay = lambda Cin: False
    hip_ops.conv.packed_taps_pay = hip_ops.packed_taps_pay
bench.main()
