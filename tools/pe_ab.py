import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objcavit_amd import hip_ops
x = torch.randn(16, 128, 240, 320, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 128, 16, 16, device="cuda") * 0.01; b = torch.zeros(128, device="cuda"); pos = torch.zeros(300, 128, device="cuda")
cache = hip_ops.ChannelsLastWeight()
for _ in range(3): hip_ops.patch_embed(x, w, b, pos, cl_cache=cache)
hip_ops.enable_timing(True)
for _ in range(10): hip_ops.patch_embed(x, w, b, pos, cl_cache=cache)
print("patch_embed", hip_ops.timing_results())
