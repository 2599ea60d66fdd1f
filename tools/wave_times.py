"""Development aid: XRT_WAVE_TIMES=<file> makes libxrt dump, per intersect launch of the last frame, each wave's start /
out-of-new-rays / exit clocks (100 MHz).  This prints how the waves' lifetimes fill the launch."""
import sys, os, importlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "/tmp/xrt_wave_times.bin"
os.environ["XRT_WAVE_TIMES"] = path
import torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr())
for _ in range(int(os.environ.get("XRT_FRAMES", "12"))):
    fr()
t = np.fromfile(path, dtype=np.uint64).reshape(16, 8192, 3).astype(np.int64)
for k in range(16):
    a = t[k]
    a = a[a[:, 2] > 0]
    if len(a) == 0:
        continue
    t0 = a[:, 0].min()
    end = (a[:, 2] - t0) / 100.0   # us
    dry = (a[:, 1] - t0) / 100.0
    dur = end.max()
    print("launch %d: %d waves, duration %.0f us; waves out of new rays at p50 %.0f us p99 %.0f us; exit p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us; mean alive %.0f%%" % (
        k, len(a), dur, np.percentile(dry, 50), np.percentile(dry, 99), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max(), 100 * (end - (a[:, 0] - t0) / 100.0).mean() / dur))
    hist, edges = np.histogram(end, bins=10, range=(0, dur))
    print("   exits per tenth of the launch:", hist.tolist())
