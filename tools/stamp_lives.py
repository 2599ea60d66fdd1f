"""How long did the waves of each traversal launch of a blocking frame live?  python tools/stamp_lives.py <config> [shard_rank shard_count]
(XRT_STAMP_DUMP: every launch's row of device-clock stamps -- start, wave count, each wave's end; kernels.h STAMP_*)."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "xrt_stamps.bin")
os.environ["XRT_STAMP_DUMP"] = path
import importlib, numpy as np, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
shard_rank, shard_count = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)   # optional: one tile shard of the frame
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr(), shard_rank=shard_rank, shard_count=shard_count)
for _ in range(6):
    st = fr()
torch.cuda.synchronize()
STRIDE = 2 + 8192
rows = np.fromfile(path, dtype=np.uint64).reshape(-1, STRIDE)
for j, r in enumerate(rows):
    n = int(r[1]); t0 = int(r[0])
    ends = (r[2:2 + n].astype(np.int64) - t0) / 100.0   # us (100 MHz clock)
    ends = ends[ends > 0]
    if len(ends) == 0:
        continue
    q = np.percentile(ends, [10, 50, 90, 99])
    print("%s%s launch %d: %5d waves, launch %8.1f us, waves alive %5.1f %% of it on average; end of the 10/50/90/99 %% wave at %.0f / %.0f / %.0f / %.0f us" % (
        name, (" shard %d/%d" % (shard_rank, shard_count)) if shard_count > 1 else "", j, len(ends), ends.max(), 100.0 * ends.mean() / ends.max(), q[0], q[1], q[2], q[3]))
