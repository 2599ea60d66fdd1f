"""Does the ORDER of a frame's tiles matter on one GPU?  python tools/tile_order.py <config> [frames]
Frame period (two in flight) and blocking frame time under (a) the default order (row-major tiles), (b) a one-rank tile table in descending
order of the previous frames' tile costs (xrt_balance_tiles with shard_count 1: the launches start with their longest packets)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
W, H = spec.width, spec.height
outs = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(2)]


def measure(tag):
    frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    for _ in range(4):
        frs[0]()
    ms = [frs[0]()["ms_total"] for _ in range(9)]
    for f in frs:
        f()
    t_open = frs[0].begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, K + 1):
        t_next = frs[i % 2].begin()
        frs[(i - 1) % 2].end(t_open)
        t_open = t_next
    frs[K % 2].end(t_open)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (K + 1) * 1e3
    print("%s %-28s blocking %.4f ms  two in flight %.4f ms" % (name, tag, statistics.median(ms), per), flush=True)
    return outs[0].cpu().numpy().copy()


ref = measure("row-major tiles")
tracer.TileCosts(reset=True)
for _ in range(2):
    tracer.RenderDevice(outs[0].data_ptr())
cost = tracer.TileCosts()
tpr, table = xrt.dist.balanced_table(W, H, 1, cost, slack=0.0)
tracer.SetTileTable(1, tpr, table)
got = measure("descending tile cost")
assert np.array_equal(ref, got), "the frame changed with the tile order"
asc = table[np.argsort(np.where(cost[table] > 0, cost[table], cost[cost > 0].min() if (cost > 0).any() else 0), kind="stable")].astype(np.int32)
tracer.SetTileTable(1, tpr, asc)
got = measure("ascending tile cost")
assert np.array_equal(ref, got)
tracer.SetTileTable(1, tpr, None)
measure("row-major tiles again")
