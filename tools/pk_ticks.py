"""Durations of the packets of one blocking frame at the product build's speed (a `make variant NAME=ticks DEFS=-DXRT_PK_TICKS` build):
   python tools/pk_ticks.py <config> [rank count]"""
import ctypes as C, importlib, os, sys
os.environ["XRT_LIB_VARIANT"] = "libxrt_ticks.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
rank, count = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr(), shard_rank=rank, shard_count=count)
lib = xrt.abi.lib()
for _ in range(4):
    fr()
torch.cuda.synchronize()
ticks, worst = (C.c_ulonglong * 32)(), (C.c_ulonglong * 16)()
lib.xrt_debug_packet_ticks(ticks, 1); lib.xrt_debug_packet_worst(worst, 1)
if "--dump" in sys.argv:
    lib.xrt_debug_packet_dump((C.c_uint * (2 * 65536))(), 2 * 65536)   # (clears it: the measured frame's first big launch is what is dumped)
st = fr()
torch.cuda.synchronize()
lib.xrt_debug_packet_ticks(ticks, 1); lib.xrt_debug_packet_worst(worst, 1)
tot = sum(ticks)
print("%s shard %d/%d: frame %.3f ms, traversal %.3f ms in %d launches, %d packets" % (name, rank, count, st["ms_total"], st["ms_intersect"], st["intersect_launches"], tot))
for b in range(32):
    if ticks[b]:
        print("    %8.1f .. %8.1f us  %9d packets  %5.1f %%   (%.1f wave-ms)" % (2 ** b / 100.0, 2 ** (b + 1) / 100.0, ticks[b], 100.0 * ticks[b] / tot, ticks[b] * 1.5 * 2 ** b / 1e5))
print("  longest packet: %.1f us (packet %d of %d in its launch, segment %d, %d valid rays, %d of them take the literal box test)" % (worst[0] / 100.0, worst[1], worst[9], worst[2], worst[8], worst[10]))
print("  packets with lanes that take the literal box test (a parallel axis or a non-finite component): %d, %.1f us each on average" % (worst[11], worst[12] / 100.0 / max(1, worst[11])))

if "--dump" in sys.argv:   # ticks against block entries, packet by packet (the frame's big launches; the last one wins)
    import numpy as np
    buf = (C.c_uint * (2 * 65536))()
    lib.xrt_debug_packet_dump(buf, 2 * 65536)
    a = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 2).astype(np.float64)
    a = a[a[:, 0] > 0]
    us, blocks = a[:, 0] / 100.0, np.maximum(a[:, 1], 1)
    print("  %d packets: us per block entry -- median %.2f, 90 %% %.2f, 99 %% %.2f, max %.2f" % (len(a), *np.percentile(us / blocks, [50, 90, 99, 100])))
    for lo, hi in ((0, 40), (40, 80), (80, 160), (160, 320), (320, 1e9)):
        m = (us >= lo) & (us < hi)
        if m.any():
            print("    packets of %4.0f .. %4.0f us: %6d, block entries median %5.1f (10 %% %5.1f, 90 %% %5.1f), us per block entry median %.2f" % (
                lo, min(hi, us.max()), m.sum(), np.median(blocks[m]), *np.percentile(blocks[m], [10, 90]), np.median(us[m] / blocks[m])))
    order = np.argsort(-us)[:12]
    print("    the longest: " + ", ".join("%.0f us / %d blocks" % (us[i], blocks[i]) for i in order))
