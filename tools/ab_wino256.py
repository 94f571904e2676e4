"""Would the 256 -> 256 second convolution at 120 x 160 pay as Winograd F(4x4,3x3) IN THE STEP?  Alone it is a tie with the direct kernel
(892 against 932 us: profiles/r03_winograd43.txt) -- but it issues a quarter of the matrix operations, and with three batches in flight the
step is power-bound (profiles/r05_binhead_two_level.txt).  `python tools/ab_wino256.py 0|1 [bench args]`: bench.py's main with the dispatch
rule as shipped (0) or extended to Cin, Cout >= 256 and B H W <= 307200 (1).  Alternate on one box.
Measured (round 5, one box, --no-extras --no-cpu-baseline): shipped rule 1074.9 / 1064.9 img/s (one batch at a time 1007.7 / 1009.0), extended
1058.4 / 1054.3 (996.5 / 995.1): -1.3 %, the transformed input's traffic costs more than the matrix work saves.  The rule stays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
on = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench                                   # (sets its environment before torch loads)
from objcavit_amd import hip_ops
if on:
    rule = lambda B, H, W, Cin, Cout: Cout % 8 == 0 and Cin >= 256 and Cout >= 256 and B * H * W <= 307200   # noqa: E731
    hip_ops.winograd_pays = rule
    hip_ops.conv.winograd_pays = rule
bench.main()
