"""-m gpu: the RCCL side of objcavit_amd/dp.py on the one GPU a test box has.  A multi-GPU node is not available to the test
suite, so what can be exercised of backend "nccl" (= RCCL on ROCm) is: the library loads, a communicator initialises over the
127.0.0.1 rendezvous the launcher uses, and the collectives of the job -- the record all-gather, the MAX all-reduce of
dp.agree_object_nmax, the rate all-gather of bench.py -- run on device tensors; with world_size = 1 they must return their input.
(The sharding / padding / ordering logic at world_size 2 is covered on gloo: tests/test_dp_gloo.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from objcavit_amd import dp
rank, local, world = dp.init_from_env("cuda")
assert (rank, local, world) == (0, 0, 1)
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # dp.init_from_env leaves a lone rank without a group
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
rec = torch.arange(40, dtype=torch.float32, device=dev).view(4, 10)
out = torch.empty_like(rec)
dist.all_gather_into_tensor(out, rec)                                     # the job's one data collective
assert torch.equal(out, rec)
t = torch.tensor([17], dtype=torch.int64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                  # dp.agree_object_nmax
assert int(t.item()) == 17
assert dp.agree_object_nmax([3, 17, 5], world, dev) == 17
r = torch.zeros(1, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(r, torch.tensor([123.5], dtype=torch.float64, device=dev))   # bench.py's per-rank rates
assert float(r.item()) == 123.5
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


@pytest.mark.timeout(300)
def test_rccl_initialises_and_runs_the_jobs_collectives_on_one_rank(tmp_path):
    import socket
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    with socket.socket() as sk:                                   # a free port, as tests/test_dp_gloo.py picks one
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
