"""what do the fork shapes of a captured forward look like to the runtime, and is the slow replay a property of
the SHAPE alone?

  python tools/graph_shapes.py            every synthetic shape in a process of its own (a crash in hipGraphLaunch stays there), then
                                          the product's captured forward (bs 1, 352x384): topology + the checker's verdict
  python tools/graph_shapes.py one NAME   one synthetic shape (what the parent runs)
  python tools/graph_shapes.py model      the product's capture only

Synthetic shapes: a main chain of 160 tiny element-wise launches (~0.8 ms of GPU time) and a side stream, forked the ways
profiles/r04_skip_overlap.txt describes.  Per shape: node / edge counts and node types as the runtime reports them
(hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType), objcavit_amd.graph_topology.check's verdict, ms per replay over 50
replays, and the same launches issued eagerly on the two streams for comparison.  No shape is replayed in a loop more than once.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = ("chain", "diamond", "three_diamonds", "side_from_top", "shape2_second_incoming_edge", "shape1_four_forks_four_joins",
          "three_branches")
N_MAIN = 160


def build(shape, main, side, side2, bufs):
    """Issue the launches of ``shape`` on ``main`` / ``side`` (current stream = main on entry)."""
    import torch

    def k(i):
        bufs[i % len(bufs)].add_(1.0)

    def on(stream, fn):
        with torch.cuda.stream(stream):
            fn()

    def fork(s):
        s.wait_stream(main)

    def join(s):
        main.wait_stream(s)

    if shape == "chain":
        for i in range(N_MAIN + 24):
            k(0)
    elif shape == "diamond":
        for i in range(N_MAIN):
            if i == 60:
                fork(side)
                on(side, lambda: [k(1) for _ in range(24)])
            if i == 110:
                join(side)
            k(0)
    elif shape == "three_diamonds":
        for i in range(N_MAIN):
            if i in (20, 70, 120):
                fork(side)
                on(side, lambda: [k(1) for _ in range(8)])
            if i in (40, 90, 140):
                join(side)
            k(0)
    elif shape == "side_from_top":
        fork(side)
        on(side, lambda: [k(1) for _ in range(24)])
        for i in range(N_MAIN):
            if i == 110:
                join(side)
            k(0)
    elif shape == "shape2_second_incoming_edge":
        fork(side)                                        # the object branch, forked at the top ...
        on(side, lambda: [k(1) for _ in range(12)])
        for i in range(N_MAIN):
            if i == 60:
                fork(side)                                # ... and the skip convolutions on the SAME side stream behind stage 4
                on(side, lambda: [k(1) for _ in range(12)])
            if i == 110:
                join(side)
            k(0)
    elif shape == "shape1_four_forks_four_joins":
        fork(side)
        on(side, lambda: [k(1) for _ in range(6)])
        for i in range(N_MAIN):
            if i in (30, 50, 70):
                fork(side)
                on(side, lambda: [k(1) for _ in range(6)])
            if i in (90, 105, 120, 135):
                join(side)
            k(0)
    elif shape == "three_branches":
        for i in range(N_MAIN):
            if i == 60:
                fork(side)
                on(side, lambda: [k(1) for _ in range(12)])
                fork(side2)
                on(side2, lambda: [k(2) for _ in range(12)])
            if i == 110:
                join(side)
                join(side2)
            k(0)
    else:
        raise SystemExit(f"unknown shape {shape}")


def run_one(shape):
    import torch
    from objcavit_amd import graph_topology as gt
    torch.zeros(1, device="cuda")
    bufs = [torch.zeros(1 << 14, device="cuda") for _ in range(3)]
    main, side, side2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    out = {"shape": shape}
    with torch.cuda.stream(main):
        build(shape, main, side, side2, bufs)             # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(main):
        for _ in range(20):
            build(shape, main, side, side2, bufs)
        main.wait_stream(side)
        main.wait_stream(side2)
    torch.cuda.synchronize()
    out["eager_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.stream(main):
        g.capture_begin(capture_error_mode="thread_local")
        build(shape, main, side, side2, bufs)
        g.capture_end()
    topo = gt.read(g.raw_cuda_graph())
    out["topology"] = topo.summary() if topo else None
    out["violations"] = gt.check(topo) if topo else None
    out["forks_joins"] = gt.describe(topo).splitlines()[1:12] if topo else None
    print(json.dumps(out), flush=True)                    # (printed BEFORE the replay: a crash below still leaves the topology)
    g.instantiate()
    with torch.cuda.stream(main):
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(main):
        for _ in range(50):
            g.replay()
    torch.cuda.synchronize()
    out["replay_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 3)
    expect = bufs[0].clone()
    print(json.dumps({"shape": shape, "replay_ms": out["replay_ms"], "eager_ms": out["eager_ms"],
                      "value_check": float(expect[0])}), flush=True)


def _shape2_forward_until_head(self, image, object_features=None, object_xywh_list=None, pad_objects_to=None, object_group=None):
    """GraphBins.forward_until_head as round 4's REJECTED shape 2: the object branch forked at the top of the forward AND the
    skip-part convolutions forked behind encoder stage 4 onto the same side stream (its first launch depends on the object
    branch's last launch and on a main-chain launch), one join behind the encoder."""
    import torch
    from objcavit_amd import hip_ops
    from objcavit_amd.modules.AdaBins import bin_edges_and_centers
    object_features = self.object_provider.padded(image)
    main = torch.cuda.current_stream(image.device)
    side = hip_ops.side_stream(image.device)
    dfe = self.dense_feature_extractor
    side.wait_stream(main)
    with torch.cuda.stream(side):
        pre = self.objcavit.object_prepass(object_features, object_xywh_list, image.device, pad_objects_to)
    skip_pre = dfe.skip_prepass(image, extra=None)
    encoded = dfe.encode(image, skip_pre)
    main.wait_stream(side)
    if skip_pre is not None:
        skip_pre.joined = True
    dense = dfe.decoder(encoded, _split_only=True, _skip_pre=skip_pre)
    bw, feat, queries = self.objcavit.forward_parts(dense, object_features, object_xywh_list, pre=pre, pad_objects_to=pad_objects_to,
                                                    object_group=object_group)
    ds = self.args[self.args.basic.dataset]
    bin_edges, centers = bin_edges_and_centers(bw, ds.min_depth, ds.max_depth)
    return feat, queries, centers, bin_edges, None


def run_model_timed(shape):
    """The real forward at bs 1 (352x384) captured in the product's shape or in shape 2: topology, checker verdict, ms per replay."""
    import torch
    from objcavit_amd import graph_topology as gt
    from objcavit_amd import synth as gen
    from objcavit_amd.config import make_args
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
    torch.set_grad_enabled(False)
    H, W = 352, 384
    args = make_args(strategy="learned", language="clip", dimensions_train=[H, W], dimensions_test=[H, W])
    m = GraphBins(args, object_provider=SyntheticObjectProvider(16, "clip")).eval()
    gen.load_into(m, 1, gen.PEAKY)
    m = m.cuda()
    img = gen.randn("img", (1, 3, H, W), 1).cuda()
    m(img)
    m(img)
    if shape == "shape2":
        GraphBins.forward_until_head = _shape2_forward_until_head
    islands = (f"conv3x3|1,{H // 2},{W // 2},128,128",) if os.environ.get("SHAPES_ISLANDS") else ()     # as bench.py captures
    g = GraphedGraphBins(m, img, check_topology=False, eager_ops=islands)
    graphs = [sg for sg in g.segments if not isinstance(sg, tuple)]
    topo = gt.read(graphs[0].raw_cuda_graph())
    print(json.dumps({"model_shape": shape, "segments": len(graphs), "islands": g.islands, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                      "topology": topo.summary(), "violations": gt.check(topo)}), flush=True)
    for _ in range(3):
        g(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g(img)
    torch.cuda.synchronize()
    print(json.dumps({"model_shape": shape, "replay_ms": round((time.perf_counter() - t0) / 30 * 1e3, 3)}), flush=True)


def run_model():
    import torch
    from objcavit_amd import graph_topology as gt
    from objcavit_amd import synth as gen
    from objcavit_amd.config import make_args
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
    torch.set_grad_enabled(False)
    H, W = 352, 384
    args = make_args(strategy="learned", language="clip", dimensions_train=[H, W], dimensions_test=[H, W])
    m = GraphBins(args, object_provider=SyntheticObjectProvider(16, "clip")).eval()
    gen.load_into(m, 1, gen.PEAKY)
    m = m.cuda()
    img = gen.randn("img", (1, 3, H, W), 1).cuda()
    m(img)                                                 # calibrating first call
    g = GraphedGraphBins(m, img, check_topology=False)
    for i, seg in enumerate(g.segments):
        if isinstance(seg, tuple):
            continue
        topo = gt.read(seg.raw_cuda_graph())
        print(f"product capture, segment {i}:", gt.describe(topo) if topo else "unreadable")
        print("  violations:", gt.check(topo) if topo else None, flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        gt.dot_print(seg.raw_cuda_graph(), os.path.join(ROOT, "gpurun_out", f"product_segment{i}.dot"))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "one":
        run_one(sys.argv[2])
    elif len(sys.argv) >= 2 and sys.argv[1] == "model":
        run_model()
    elif len(sys.argv) >= 3 and sys.argv[1] == "model_timed":
        run_model_timed(sys.argv[2])
    else:
        for s in SHAPES:                                   # the parent never touches the GPU
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", s], capture_output=True, text=True, timeout=150)
            print(r.stdout.strip())
            if r.returncode != 0:
                print(json.dumps({"shape": s, "exit": r.returncode, "stderr_tail": r.stderr.strip().splitlines()[-3:]}), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "model"], capture_output=True, text=True, timeout=600)
        print(r.stdout.strip())
        if r.returncode != 0:
            print("model capture failed:", r.stderr.strip().splitlines()[-5:])
