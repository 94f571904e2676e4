"""The split3 cross-attention launches at one batch size, back to back (for tools/pmc_xattn.sh).  Usage: xattn_one.py B S"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B, S = int(sys.argv[1]), int(sys.argv[2])
E, H, N = 128, 4, 32
torch.manual_seed(0)
w = torch.randn(3 * E, E, device="cuda") * 0.1; b = torch.randn(3 * E, device="cuda") * 0.1
wo = torch.randn(E, E, device="cuda") * 0.1; bo = torch.randn(E, device="cuda") * 0.1
x = torch.randn(B, S, E, device="cuda")
k = torch.full((B, S, E), 1e-4, device="cuda"); k[:, S - N:, :] = torch.randn(B, N, E, device="cuda")
mask = torch.ones(B, S, dtype=torch.bool, device="cuda"); mask[:, :N] = False
cache = {}
for _ in range(8):
    hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=cache)
torch.cuda.synchronize()
print("done")
