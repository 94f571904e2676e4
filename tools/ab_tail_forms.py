"""(needs tools/diag/layer_tail_row_tiles.patch.txt applied: ocv_layer_tail_h2_set_dispatch is not in the product)
Round 6: the token-local layer tail (csrc/token_h2.hip) in its forms -- (row tiles of 32 tokens per workgroup,
feed-forward groups per row block) -- on a whole 4-layer stack (1 + 2 x 4 launches: ocv_encoder_stack_fwd), same process, forms
interleaved in three blocks, best block.  `python tools/ab_tail_forms.py [iters]` -> us per STACK call and the largest deviation from
the (1, 1) form.  Shapes: the image tokens at bs 16 / 8 / 2 / 1 (S = 300), KITTI's at bs 8 (S = 418), the object tokens at bs 16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from objcavit_amd import hip_ops, _lib
from objcavit_amd.modules.layers import HipEncoderStack

IT = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load()
torch.manual_seed(0)
enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(128, 4, 1024, batch_first=True), 4, enable_nested_tensor=False).eval().cuda()
stack = HipEncoderStack(enc)


def timed(fn, n=IT):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


FORMS = [(1, 1), (1, 2), (1, 4), (1, 8), (2, 1), (2, 2), (2, 4), (2, 8), (0, 0)]
print(f"# us per 4-layer stack call ({IT} calls per block); (rt, g): row tiles per workgroup, feed-forward groups; (0, 0) = the shipped rule")
print("# " + " ".join(f"{str(f):>9s}" for f in FORMS))
for (B, S, counts) in [(16, 300, None), (8, 300, None), (8, 418, None), (2, 300, None), (1, 300, None), (16, 32, 32)]:
    x = torch.randn(B, S, 128, device="cuda")
    mask = None
    res = {}
    for f in FORMS:
        assert lib.ocv_layer_tail_h2_set_dispatch(*f) == 0
        res[f] = [1e30, stack(x, mask).clone()]
    for _ in range(3):
        for f in FORMS:
            assert lib.ocv_layer_tail_h2_set_dispatch(*f) == 0
            res[f][0] = min(res[f][0], timed(lambda: stack(x, mask)))
    lib.ocv_layer_tail_h2_set_dispatch(0, 0)
    ref = res[(1, 1)][1]
    dev = max(float((res[f][1] - ref).abs().max() / ref.abs().max()) for f in FORMS)
    print(f"B {B:2d} S {S:4d}: " + " ".join(f"{res[f][0]:9.1f}" for f in FORMS) + f"   max dev from (1, 1): {dev:.1e}", flush=True)
