"""HIP-event time of the split3 cross-attention call (both launches) at one batch: xattn_time.py B S [reps].  With OCV_LIB_PATH
pointing at an ablation build (tools/diag/xattn_ablation.patch.txt) the differences localise the cost inside the tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B, S = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
E, H, N = 128, 4, 32
torch.manual_seed(0)
w = torch.randn(3 * E, E, device="cuda") * 0.1; b = torch.randn(3 * E, device="cuda") * 0.1
wo = torch.randn(E, E, device="cuda") * 0.1; bo = torch.randn(E, device="cuda") * 0.1
x = torch.randn(B, S, E, device="cuda")
k = torch.full((B, S, E), 1e-4, device="cuda"); k[:, S - N:, :] = torch.randn(B, N, E, device="cuda")
mask = torch.ones(B, S, dtype=torch.bool, device="cuda"); mask[:, :N] = False
cache = {}
for _ in range(5):
    hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=cache)
torch.cuda.synchronize()
hip_ops.enable_timing(True)
for _ in range(reps):
    hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=cache)
t = hip_ops.timing_results()["mha_cross"][1] * 1e3
print(f"{os.path.basename(os.environ.get('OCV_LIB_PATH', 'product')):28s} B={B} S={S}: {t:8.1f} us")
