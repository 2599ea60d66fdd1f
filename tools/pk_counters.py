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
fr = tracer.PrepareDevice(out.data_ptr())
lib = xrt.abi.lib()
buf = (C.c_ulonglong * 16)()
for _ in range(3):
    fr()
torch.cuda.synchronize()
lib.xrt_debug_packet_counters(buf, 1)
st = fr()
torch.cuda.synchronize()
lib.xrt_debug_packet_counters(buf, 1)
names = ["walks", "blocks entered", "child visits", "leaf children", "leaves some lane may need (bucket rule, facing)", "leaves past their tight box", "run box tests",
         "runs scanned", "triangle steps", "-", "pops", "child visits with keys", "lanes at the root", "lane-visits of children", "lane-triangle tests", "-"]
w = max(buf[0], 1)
print("%s: %d walks in one frame (%d rays traversed)" % (name, buf[0], st["rays_traversed"]))
for i, n in enumerate(names):
    if n != "-":
        print("  %-50s %12d  %8.2f per walk" % (n, buf[i], buf[i] / w))
print("  lanes per child visit %.1f, lanes per triangle step %.1f" % (buf[13] / max(buf[2], 1), buf[14] / max(buf[8], 1)))
