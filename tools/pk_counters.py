"""Event counts of the wave-packet walk (a `make variant NAME=cnt DEFS=-DXRT_PK_COUNTERS` build): python tools/pk_counters.py <config>
   XRT_LIB_VARIANT=libxrt_cnt.so is set here.  Counts are per packet walk (one per packet and mesh visit), averaged over one blocking frame."""
import ctypes as C, importlib, os, sys
os.environ["XRT_LIB_VARIANT"] = "libxrt_cnt.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
# optional: one tile shard of the frame -- python tools/pk_counters.py C5 <rank> <count>
rank, count = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
fr = tracer.PrepareDevice(out.data_ptr(), shard_rank=rank, shard_count=count)
lib = xrt.abi.lib()
buf = (C.c_ulonglong * 16)()
for _ in range(3):
    fr()
torch.cuda.synchronize()
lib.xrt_debug_packet_counters(buf, 1)
ticks = (C.c_ulonglong * 32)()
lib.xrt_debug_packet_ticks(ticks, 1)
worst = (C.c_ulonglong * 16)()
lib.xrt_debug_packet_worst(worst, 1)
st = fr()
torch.cuda.synchronize()
lib.xrt_debug_packet_counters(buf, 1)
lib.xrt_debug_packet_ticks(ticks, 1)
names = ["walks", "blocks entered", "child visits", "leaf children", "leaves some lane may need (bucket rule, facing)", "leaves past their tight box", "run box tests",
         "runs scanned", "triangle steps", "-", "pops", "child visits with keys", "lanes at the root", "lane-visits of children", "lane-triangle tests", "-"]
w = max(buf[0], 1)
print("%s: %d walks in one frame (%d rays traversed)" % (name, buf[0], st["rays_traversed"]))
for i, n in enumerate(names):
    if n != "-":
        print("  %-50s %12d  %8.2f per walk" % (n, buf[i], buf[i] / w))
print("  lanes per child visit %.1f, lanes per triangle step %.1f" % (buf[13] / max(buf[2], 1), buf[14] / max(buf[8], 1)))
print("  packets by duration (device clock, 10 ns ticks; frame %.3f ms, traversal %.3f ms):" % (st["ms_total"], st["ms_intersect"]))
tot = sum(ticks)
for b in range(32):
    if ticks[b]:
        print("    %8.1f .. %8.1f us  %9d packets  %5.1f %%   (their time, at the bucket's middle: %.1f wave-ms)" % (2 ** b / 100.0, 2 ** (b + 1) / 100.0, ticks[b], 100.0 * ticks[b] / tot, ticks[b] * 1.5 * 2 ** b / 1e5))
lib.xrt_debug_packet_worst(worst, 1)
print("  the packet that took longest (any launch of the frame): %.1f us; packet %d of %d (segment %d), %d valid rays: %d mesh walks, %d blocks entered, %d child visits, %d triangle steps, %d run box tests" % (
    worst[0] / 100.0, worst[1], worst[9], worst[2], worst[8], worst[3], worst[4], worst[5], worst[6], worst[7]))
