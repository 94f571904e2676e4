"""bench.py with its slot streams chosen by hip_ops.independent_streams (1: checked pairwise for a shared hardware queue) or taken as the
runtime deals them (0: torch.cuda.Stream() x n, rounds 2 - 5).  `python tools/ab_indep_streams.py 0|1 [bench args]`; alternate on one box.
Record: profiles/r06_stream_queues.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
on = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench                                   # (sets its environment before torch loads)
import torch
from objcavit_amd import hip_ops
if not on:
    plain = lambda n, device, candidates=16: [torch.cuda.Stream(device=device) for _ in range(n)]   # noqa: E731
    hip_ops.independent_streams = plain
    hip_ops._core.independent_streams = plain
bench.main()
