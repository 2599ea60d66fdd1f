"""Blocking frames of one tile shard (for a profiler): python tools/shard_frames.py <config> <rank> <count> [frames]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
xrt = importlib.import_module("xna-ray-trace_amd")
name, rank, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
tx, ty, tpr = xrt.dist.shard_layout(spec.width, spec.height, count)
out = torch.zeros(max(tpr * 512, spec.width * spec.height if count == 1 else 0), dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr(), shard_rank=rank, shard_count=count)
for _ in range(n):
    st = fr()
torch.cuda.synchronize()
print("%s shard %d/%d: ms_total %.4f ms_intersect %.4f launches %d" % (name, rank, count, st["ms_total"], st["ms_intersect"], st["intersect_launches"]))
