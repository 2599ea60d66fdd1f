import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import importlib, torch
xrt = importlib.import_module("xna-ray-trace_amd")
for thr in (20, 100):
    spec = xrt.configs.config("C3")
    spec.scene_threshold = thr
    scene, tracer = xrt.configs.build_product(spec)
    out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
    fr = tracer.PrepareDevice(out.data_ptr())
    for _ in range(5): st = fr()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st = fr()
    torch.cuda.synchronize()
    tracer.collect_stats = True
    fr2 = tracer.PrepareDevice(out.data_ptr()); st2 = fr2()
    rays = st2["rays_closest"] + st2["rays_shadow"]
    print("scene_threshold %d: frame %.3f ms intersect %.3f  instance visits/ray %.1f" % (thr, (time.perf_counter() - t0) / 20 * 1e3, st["ms_intersect"], st2["instance_visits"] / rays))
