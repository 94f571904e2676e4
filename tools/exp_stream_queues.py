"""Which torch streams share a hardware queue?  For N new streams + the null stream: a long kernel on stream i, then a tiny one on stream j;
if j's kernel finishes only after i's, the two streams sit on the same hardware queue.  Prints the collision classes.
`GPU_MAX_HW_QUEUES=4 python tools/exp_stream_queues.py [N]`"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
x = torch.zeros(1, device=dev)
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream() for _ in range(N)]
names = ["null"] + [f"s{i}" for i in range(N)]
torch.cuda.synchronize()
LONG = 3_000_000          # cycles of the 100 MHz counter?  measured below


for st in streams:                      # first use of every stream (its hardware queue is dealt then), in creation order
    with torch.cuda.stream(st):
        x.zero_()
torch.cuda.synchronize()


def probe(i, j):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(streams[i]):
        e0.record()
        torch.cuda._sleep(LONG)
        e1.record()
    with torch.cuda.stream(streams[j]):
        x.zero_()                       # in place: no allocation inside the probe (an allocation waits for the device by itself)
        e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(e2)


long_ms, _ = probe(1, 2)
print(f"# GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}, long kernel = {long_ms:.2f} ms")
cls = {}
for i in range(len(streams)):
    row = []
    for j in range(len(streams)):
        if i == j:
            row.append(" . ")
            continue
        l, t = probe(i, j)
        row.append(" X " if t > 0.5 * l else " - ")
    print(f"{names[i]:>5s} " + "".join(row))
print("# X: the tiny kernel on the column's stream waited for the long kernel on the row's stream (same hardware queue)")
