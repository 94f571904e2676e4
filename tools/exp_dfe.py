"""Experiment: encoder / decoder time under MIOpen settings (default vs benchmark=True vs channels_last)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd.config import make_args
from objcavit_amd.modules.DenseFeatureExtractor import DenseFeatureExtractor

torch.set_grad_enabled(False)
B = 16
def run(tag, bench, cl):
    torch.backends.cudnn.benchmark = bench
    m = DenseFeatureExtractor(make_args()).eval().cuda()
    x = torch.randn(B, 3, 480, 640, device="cuda")
    if cl:
        m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r
    t0 = time.perf_counter()
    te, feats = t(lambda: m.encoder(x))
    td, out = t(lambda: m.decoder(feats))
    print(f"{tag:28s} encoder {te:7.2f} ms  decoder {td:7.2f} ms  total {te+td:7.2f} ms  (setup {time.perf_counter()-t0:.1f}s)", flush=True)

for tag, bench, cl in (("default", False, False), ("benchmark", True, False), ("channels_last", False, True), ("benchmark+channels_last", True, True)):
    if len(sys.argv) > 1 and tag not in sys.argv[1:]:
        continue
    try:
        run(tag, bench, cl)
    except Exception as e:
        print(tag, "FAILED", repr(e)[:200], flush=True)
