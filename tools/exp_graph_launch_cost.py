"""Host cost of one hipGraphLaunch of the captured forward (bs b): issue K replays back to back on an idle device (K small: the queue never
fills) and time the ISSUE alone; then the four-slot pipelined loop issued from one thread and from one thread per slot.
`python tools/exp_graph_launch_cost.py [batch]`"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from objcavit_amd import hip_ops
from objcavit_amd.graph import GraphedGraphBins
torch.set_grad_enabled(False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nslot = 4
dev = torch.device("cuda:0")
wl = bench.Workload(2, b)
model = bench.build_model(dev, wl)[0]
img = bench.synthetic_images(b, 42, wl.H, wl.W).to(dev)
model(img)
sts = hip_ops.independent_streams(nslot, dev)
slots = [GraphedGraphBins(model, img, in_flight=nslot, stream=sts[k]) for k in range(nslot)]
g = slots[0]
seg = [s for s in g.segments if not isinstance(s, tuple)][0]
for K in (1, 2, 4):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        with torch.cuda.stream(g.stream):
            t0 = time.perf_counter()
            for _ in range(K):
                seg.replay()
            ts.append((time.perf_counter() - t0) / K)
        torch.cuda.synchronize()
    print(f"bs {b}: host time of one hipGraphLaunch ({sum(x for x in g.segment_nodes if x)} nodes), {K} back to back on an idle device: {min(ts) * 1e3:.3f} ms", flush=True)


def replay(k):
    s = slots[k % nslot]
    with torch.cuda.stream(s.stream):
        s(img)


N = 240
for threads in (1, nslot):
    for k in range(2 * nslot):
        replay(k)
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        if threads == 1:
            for k in range(N):
                replay(k)
        else:
            def worker(s):
                for k in range(s, N, nslot):
                    replay(k)
            th = [threading.Thread(target=worker, args=(s,)) for s in range(nslot)]
            [t.start() for t in th]
            [t.join() for t in th]
        torch.cuda.synchronize()
        best = max(best, N * b / (time.perf_counter() - t0))
    print(f"bs {b}, {nslot} slots, {threads} issuing thread(s): {best:.1f} img/s", flush=True)
