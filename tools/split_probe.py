"""Split walks (packet.hip) on one config: blocking and pipelined frame time and the split counters per frame.
   python tools/split_probe.py C5 [rank count] -- environment: XRT_PK_SPLIT, XRT_PK_BUDGET, XRT_PK_BUDGET_ITEM (microseconds)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
rank, count = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
N = 24
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
outs = [torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda") for _ in range(2)]
frs = [tracer.PrepareDevice(o.data_ptr(), shard_rank=rank, shard_count=count) for o in outs]
for _ in range(4):
    st = frs[0]()
torch.cuda.synchronize()
scene.SplitStats()
t0 = time.perf_counter()
longest, inter, total = 0.0, 0.0, 0.0
for _ in range(N):
    st = frs[0]()
    longest += st["ms_intersect_longest"] / N; inter += st["ms_intersect"] / N; total += st["ms_total"] / N
torch.cuda.synchronize()
blocking = (time.perf_counter() - t0) / N
g = scene.SplitStats()
t = [frs[0].begin(), frs[1].begin()]
for i in range(4):
    frs[i & 1].end(t[i & 1]); t[i & 1] = frs[i & 1].begin()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2 * N):
    frs[i & 1].end(t[i & 1]); t[i & 1] = frs[i & 1].begin()
frs[0].end(t[0]); frs[1].end(t[1])
piped = (time.perf_counter() - t0) / (2 * N)
env = " ".join("%s=%s" % (k, os.environ[k]) for k in ("XRT_PK_SPLIT", "XRT_PK_BUDGET", "XRT_PK_BUDGET_ITEM", "XRT_PK_LONG", "XRT_PK_BUDGET_LONG") if k in os.environ)
print("%s %d/%d [%s]: blocking %.3f ms (longest launch %.3f, intersect %.3f), two in flight %.3f ms; per frame: %.0f subtrees handed over, %.0f packets split, %.0f written by a taker" % (
    name, rank, count, env, blocking * 1e3, longest, inter, piped * 1e3, g[0] / N, g[2] / N, g[3] / N) + "; GPU ms_total %.3f" % total)
