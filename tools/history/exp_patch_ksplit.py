"""A/B of the patch embedding's K pieces (OCV_PATCH_PARTS=n vs the modelled choice; 16 = one piece per patch row, the form of rounds
2 - 4) at the validation loop's batch sizes.  python tools/exp_patch_ksplit.py   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    from objcavit_amd import hip_ops as ops
    for B in (1, 2, 4, 8, 16):
        x = torch.randn(B, 128, 240, 320, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(128, 128, 16, 16, device="cuda") * 0.005
        b = torch.randn(128, device="cuda")
        pos = torch.randn(300, 128, device="cuda")
        xs = ops.split_act(x, True)
        pw = ops.PatchEmbedSplitWeight().get(w, True)
        f = lambda: ops.patch_embed_split(xs, pw[0], pw[1], b, pos, oscale=pw[2])
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"  bs {B:2d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
    sys.exit(0)
for ks in ("16", "8", "13", "26", "32", "64", "128", ""):
    env = dict(os.environ)
    env.pop("OCV_PATCH_PARTS", None)
    if ks:
        env["OCV_PATCH_PARTS"] = ks
    print(f"OCV_PATCH_PARTS={ks or '(modelled)'}", flush=True)
    subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
