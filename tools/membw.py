"""HBM baselines for the resize kernel's traffic: fill and copy of a 1.41 GB buffer (torch kernels), event-timed."""
import torch
n = 16 * 240 * 320 * 288 * 2          # bf16 elements of the 240x320 hl32 activation
a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 2 / 1e9
ms = t(lambda: a.zero_()); print(f"fill  {gb:.2f} GB: {ms*1e3:.0f} us  {gb/ms:.2f} TB/s written")
ms = t(lambda: b.copy_(a)); print(f"copy  {gb:.2f} GB: {ms*1e3:.0f} us  {2*gb/ms:.2f} TB/s read+write")
x = torch.randn(n // 2, device="cuda")
ms = t(lambda: torch.sum(x)); print(f"read  {gb:.2f} GB: {ms*1e3:.0f} us  {gb/ms:.2f} TB/s read")
