#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray / octree / triangle hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5|C5_1spp|H100k] [--scale S]

One step = one frame of the workload: every pixel's CastRay tree (closest-hit, shadow and reflection
queries).  Default workload is the configuration BASELINE.json quotes its target on and which fits one GPU:
C5 = the 1M-triangle mesh at 1920x1080, depth 3, 16 sub-rays per pixel; C2 / C3 / C4 are reported as short side
measurements in "other_configs" at N=1.  The `roofline` block carries three fractions of the dominant kernel
(the traversal launches: k_packet / k_intersect) -- reference-algorithm bytes, measured HBM traffic and VALU issue (the latter two from the
committed rocprofv3 PMC passes of THIS build, profiles/pmc_traversal.json, divided by launch durations measured live
with HIP events) -- and names as `bound` whichever physical limit is closest.
For N > 1 the driver starts one process per GPU (torch.distributed.run); the frame is sharded by 64x8 image
tiles (total work fixed -> "strong" scaling), each rank renders its tiles from its own scene replica, and the
per-frame exchange step is one RCCL gather of the tile buffers onto rank 0 followed by a de-tile kernel.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

xrt = importlib.import_module("xna-ray-trace_amd")

WORKLOADS = {
    "C1": "C1: Free_crate mesh (12 triangles), 256x256, depth=1 (MaxReflections 0), 1 spot light",
    "C2": "C2: Free_crate mesh (12 triangles), 1920x1080, depth=3 (reflect+shadow, MaxReflections 2), 1 spp, 1 spot light",
    "C3": "C3: Free_crate x64 instanced grid (tessellated crate n=11, 92,928 instanced triangles), 1920x1080, depth=3",
    "C4": "C4: C3 scene at 3840x2160, depth=3",
    "C5": "C5: 1M-triangle heightfield (999,698 triangles), 1920x1080, depth=3, 16 sub-rays per pixel",
    "C5_1spp": "C5 scene, 1920x1080, depth=3, 1 spp",
    "H100k": "100k-triangle heightfield (100,352 triangles), 1920x1080, depth=3, 1 spp",
    "G1": "G1: the reference's default scene (Game1.cs): 2x2 Transparent spheres (Sphere.fbx, 960 triangles each), 512x512, MaxReflections 8",
    "G2": "G2: the reference's content (monkey, torus, sphere, cube, textured ground with the content project's parameters), 1280x720, MaxReflections 4, two lights",
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# VALU issue peak: 256 CUs x 4 SIMD-32 x 32 lanes per cycle x 2.4 GHz (MI355X_MICROARCH.md: a wave64 VALU instruction issues over
# 2 cycles on a SIMD-32 with >= 2 waves resident, chip table "Max clock 2400 MHz") = 78.6 T lane-operations per second
# (= the 157.3 TFLOP/s vector peak, which counts a fused multiply-add as two; this path is built with contraction off)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
CU_CYCLES_PER_S = 256 * 2.4e9           # CU-cycles per second (256 CUs, 2.4 GHz; the microbenchmarks ran at 2.39-2.40)
# Scalar-side issue rates, measured (tools/microbench/scalar_mix.hip -> profiles/r03/scalar_mix_peak.txt, six waves per SIMD = k_packet's
# occupancy; MI355X_MICROARCH.md gives none).  Instructions per CU per cycle with nothing else running:
#   SALU 0.964 | s_cbranch not taken 0.946, s_branch taken 0.591 (harmonic mean 0.728: a kernel's taken share is not counted) | s_load_dwordx8 0.138
# and the classes ISSUE SIDE BY SIDE: k_packet's mix (80 % SALU / 16 % branch / 4 % SMEM) retires 1.035 instructions per cycle -- more than
# SALU alone, where one shared port would give 0.747 -- so each class is a ceiling of its own and the kernel's scalar-side bound is the
# largest of the three fractions, not their sum (round 2 summed them over the SALU-only peak).
SALU_PEAK_IPC, BRANCH_PEAK_IPC, SMEM_PEAK_IPC, MIX_PK_IPC = 0.964, 0.728, 0.138, 1.035
# VALU: 0.44 wave64 instructions per cycle and SIMD with eight waves per SIMD (tools/microbench/valu_peak.hip, profiles/r02_m)
VALU_MEASURED_TLANEOPS = 70.0
KERNEL_SOURCES = ("kernels.hip", "packet.hip", "kernels.h", "device_util.h", "traverse.h", "xrt_core.h", "xrt_api.cpp")


def build_id():
    """Hash of the sources the traversal kernels are compiled from and of the host code that decides which launches a frame makes:
    PMC figures (per-launch averages) are only quoted for the build they were taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "xna-ray-trace_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(config):
    """Per-launch PMC averages of k_intersect for `config` (tools/pmc_collect.py, separate rocprofv3 --pmc passes), or None
    when the committed file was taken on another build of the kernel."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traversal.json")))
    except Exception:
        return None
    if pmc.get("build_id") != build_id():
        return None
    return pmc.get("configs", {}).get(config)


def intersect_bytes(st):
    """Algorithmic bytes of the traversal kernel for one frame: everything except the shading
    terms (80 B per shaded hit, 4 B per pixel write)."""
    return st["algorithmic_bytes"] - 80 * st["shaded_hits"] - 4 * st["pixels"]


_LONGEST = [0.0, 0]   # xrt_stats.ms_intersect_longest summed over the frames of the last timed region, and their number

GATHER_EVERY = 4   # N > 1: one RCCL gather moves the tiles of this many frames (the collective's fixed cost is ~a frame's GPU time)


def time_frames_sharded(tracer, outs, steps, warmup, rank, world, width, height, gathered_out, table_dev=None, tiles_per_rank=None):
    """N > 1.  W warm-up frames, then K timed frames bracketed by barrier + synchronize on both sides.  Each rank renders
    its tiles of frame i into slot i % M of a group of M tile buffers (two groups); when a group is full (or at the end)
    ONE gather moves it to rank 0 -- the path's exchange step, RCCL over xGMI -- while the next group is being rendered,
    and rank 0 de-tiles its M frames.  Every frame is gathered and de-tiled before the closing barrier."""
    M, n_out = GATHER_EVERY, outs[0].numel()
    nccl = dist.get_backend() == "nccl"
    groups = [torch.zeros(M * n_out, dtype=outs[0].dtype, device="cuda") for _ in range(2)]
    renders = [[tracer.PrepareDevice(g[m * n_out:(m + 1) * n_out].data_ptr(), shard_rank=rank, shard_count=world) for m in range(M)] for g in groups]
    recv = [torch.empty(world * M * n_out, dtype=outs[0].dtype, device="cuda" if nccl else "cpu") if rank == 0 else None for _ in groups]
    pending = [None, None]           # per group: (wait function of its gather, frames in it)
    open_frames, acc = [], [0.0, 0]

    def finish(g):                   # group g's gather is done (its buffers are free again) and its frames are de-tiled
        if pending[g] is None:
            return
        wait, count = pending[g]
        pending[g] = None
        got = wait()
        if nccl:
            # work.wait() only makes torch's stream wait for the collective; the next frames are rendered on libxrt's own
            # stream, so the host must know the tile buffers have been read before they are rendered into again
            torch.cuda.current_stream().synchronize()
        if rank == 0:
            dev = got if nccl else got.cuda()          # gloo: rehearsal on a box with fewer GPUs than ranks
            for m in range(count):
                xrt.dist.detile_device(dev, width, height, world, gathered_out, rank_stride=M * n_out, offset=m * n_out, table_dev=table_dev, tiles_per_rank=tiles_per_rank)

    def flush(g, count):
        src = groups[g] if nccl else groups[g].cpu()
        pending[g] = (xrt.dist.gather_frame_async(src, recv=recv[g]), count)

    def complete():
        t, j = open_frames.pop(0)
        st = renders[(j // M) % 2][j % M].end(t)   # blocking: this rank's tiles of frame j are in HBM
        acc[0] += st["ms_intersect"]
        acc[1] += st["intersect_launches"]
        _LONGEST[0] += st.get("ms_intersect_longest", 0.0); _LONGEST[1] += 1
        if j % M == M - 1:
            flush((j // M) % 2, M)
        return j

    def frame(i):
        g, m = (i // M) % 2, i % M
        if m == 0:
            finish(g)
        open_frames.append((renders[g][m].begin(), i))   # host side of frame i overlaps the GPU side of frame i-1
        while len(open_frames) >= 2:
            complete()

    def drain():
        j = -1
        while open_frames:
            j = complete()
        if j >= 0 and j % M != M - 1:
            flush((j // M) % 2, j % M + 1)               # the last, partly filled group
        finish(0)
        finish(1)
    for i in range(warmup):
        frame(i)
    drain()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    acc[0], acc[1] = 0.0, 0
    _LONGEST[0], _LONGEST[1] = 0.0, 0
    t0 = time.perf_counter()
    for i in range(steps):
        frame(i)
    drain()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, acc[0], acc[1]


def time_frames(tracer, outs, steps, warmup, rank, world, width, height, gathered_out, stream=None, table_dev=None, tiles_per_rank=None):
    if world > 1:
        return time_frames_sharded(tracer, outs, steps, warmup, rank, world, width, height, gathered_out, table_dev, tiles_per_rank)
    """One GPU.  W warm-up frames, then K timed frames bracketed by synchronize on both sides; frame i is enqueued before
    frame i-1 is waited for."""
    # stream None: libxrt decides (single-chunk frames of >= 0.05 ms of GPU time alternate between two streams of its own and
    # overlap on the GPU).  A given stream serialises the frames: used for per-launch timings.
    renders = [tracer.PrepareDevice(o.data_ptr(), stream=stream) for o in outs]   # camera / lights marshalled once
    open_frames, acc = [], [0.0, 0]

    def complete():
        t, j = open_frames.pop(0)
        st = renders[j % 2].end(t)                 # blocking: frame j is in HBM
        acc[0] += st["ms_intersect"]
        acc[1] += st["intersect_launches"]
        _LONGEST[0] += st.get("ms_intersect_longest", 0.0); _LONGEST[1] += 1

    def frame(i):
        open_frames.append((renders[i % 2].begin(), i))   # host side of frame i overlaps the GPU side of frame i-1
        while len(open_frames) >= 2:
            complete()

    def drain():
        while open_frames:
            complete()
    for i in range(warmup):
        frame(i)
    drain()
    torch.cuda.synchronize()
    acc[0], acc[1] = 0.0, 0
    _LONGEST[0], _LONGEST[1] = 0.0, 0
    t0 = time.perf_counter()
    for i in range(steps):
        frame(i)
    drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, acc[0], acc[1]


def run_config(name, scale, steps, warmup, rank, local_rank, world, with_stats=True, n_gpus_in_library=1):
    spec = xrt.configs.config(name, scale)
    torch.cuda.synchronize()
    mem0 = torch.cuda.mem_get_info()[0]   # free bytes of the device before the scene exists (hipMemGetInfo: everything on the card counts)
    t0 = time.perf_counter()
    scene, tracer = xrt.configs.build_product(spec, device=local_rank)
    torch.cuda.synchronize()
    mem1 = torch.cuda.mem_get_info()[0]
    tracer.NumGpus = n_gpus_in_library   # > 1: every frame below is ONE xrt_render_device call that the library spreads over that many devices
    tracer.BalanceTiles = n_gpus_in_library > 1 and os.environ.get("XRT_BENCH_BALANCE", "1") != "0"   # ... its tiles dealt by the previous frame's costs
    build_s = time.perf_counter() - t0
    W, H = spec.width, spec.height
    tx, ty, tpr = xrt.dist.shard_layout(W, H, world)
    n_out = tpr * 512 if world > 1 else W * H
    outs = [torch.zeros(n_out, dtype=torch.int32, device="cuda") for _ in range(2)]   # two frames are in flight (one per frame context / stream)
    final = torch.zeros(W * H, dtype=torch.int32, device="cuda") if (world > 1 and rank == 0) else None
    torch.cuda.synchronize()
    mem1b = torch.cuda.mem_get_info()[0]   # (the output buffers above are not work buffers)
    # untimed: exact reference-work counters of this rank's shard (algorithmic bytes, ray counts)
    tracer.collect_stats = with_stats
    st0 = tracer.RenderDevice(outs[0].data_ptr(), shard_rank=rank, shard_count=world)
    tracer.collect_stats = False
    # N > 1: cost-aware tile assignment (xrt.h xrt_scene_set_tile_table) -- two untimed frames under the round-robin layout leave the cost of
    # every tile on the rank that rendered it; summed over the ranks, every rank derives the same longest-first table (xrt_balance_tiles) and
    # renders its row of it from then on.  The static counterpart of the reference's dynamic row stealing (RT:48-52); XRT_BENCH_BALANCE=0: round-robin.
    table_dev, tprb, balance = None, tpr, None
    if world > 1 and os.environ.get("XRT_BENCH_BALANCE", "1") != "0":
        tracer.TileCosts(reset=True)
        for _ in range(2):
            tracer.RenderDevice(outs[0].data_ptr(), shard_rank=rank, shard_count=world)
        cost = torch.from_numpy(tracer.TileCosts())
        if dist.get_backend() == "nccl":
            cost = cost.cuda()
        dist.all_reduce(cost)
        cost = cost.cpu().numpy()
        if cost.any():
            tprb, table = xrt.dist.balanced_table(W, H, world, cost)
            tracer.SetTileTable(world, tprb, table)
            table_dev = torch.from_numpy(table).cuda()
            loads = np.array([cost[r[r >= 0]].sum() for r in table.reshape(world, tprb)])
            rr = np.array([cost[np.arange(cost.size) % world == k].sum() for k in range(world)])
            balance = {"by_cost_round_robin": round(float(rr.mean() / rr.max()), 4), "by_cost_table": round(float(loads.mean() / loads.max()), 4), "tiles_per_rank": int(tprb)}
            outs = [torch.zeros(tprb * 512, dtype=torch.int32, device="cuda") for _ in range(2)]
    dt, ms_int, launches = time_frames(tracer, outs, steps, warmup, rank, world, W, H, final, table_dev=table_dev, tiles_per_rank=tprb)
    ms_longest = _LONGEST[0] / max(_LONGEST[1], 1)   # the frame's longest traversal launch, averaged over the timed frames
    torch.cuda.synchronize()
    mem2 = torch.cuda.mem_get_info()[0]   # after two frames in flight: both frame contexts hold their work buffers
    # the pixels the timed loop's own render objects produced last (read back after the timed region): what main() compares with the oracle
    last_frame = outs[(steps - 1) % 2].cpu().numpy().view(np.uint32).copy() if (world == 1 and steps > 0) else None
    if world > 1 and rank == 0 and os.environ.get("XRT_BENCH_VERIFY"):   # rehearsals: the gathered, de-tiled frame is the unsharded one
        whole = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        tracer.RenderDevice(whole.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(whole, final), "gathered frame differs from the unsharded render"
        print("bench.py: gathered frame verified against the unsharded render", file=sys.stderr)
    rays = st0["rays_closest"] + st0["rays_shadow"]
    device_memory = {"scene_GB": round((mem0 - mem1) / 1e9, 3), "frame_work_buffers_GB": round((mem1b - mem2) / 1e9, 3),
                     "what": "hipMemGetInfo differences: scene arrays after the build; work buffers of BOTH frame contexts (two frames in flight) after the timed frames, output buffers excluded"}
    res = dict(ms_longest=ms_longest, tile_balance=balance, rays=rays, seconds=dt, ms_intersect=ms_int, launches=launches, stats=st0, build_s=build_s, width=W, height=H, last_frame=last_frame, device_memory=device_memory,
               tris=sum(m[0].ntri for m in spec.meshes), instances=len(spec.objects), overlapped=False)
    if world == 1 and dt / max(steps, 1) * 1e3 >= 0.04:
        # The frames of the timed region overlapped pairwise on the GPU (two streams), so a launch's duration includes time
        # it shared with the other frame's launches.  A short serialised pass (one explicit stream) gives the duration of a
        # launch that has the GPU to itself -- the figure a roofline of the kernel is about.
        side = torch.cuda.Stream()
        k = max(2, min(steps, 5))
        dt_s, ms_s, l_s = time_frames(tracer, outs, k, 1, rank, world, W, H, final, stream=side.cuda_stream)
        res.update(overlapped=True, serial_seconds=dt_s, serial_steps=k, serial_ms_intersect=ms_s, serial_launches=l_s, serial_ms_longest=_LONGEST[0] / max(_LONGEST[1], 1))
    return res, spec


def time_host_output(tracer, spec, steps, warmup):
    """The seam the north star names ends in the host's Color[] (CurrentTarget.SetData, RT:122-123): the same frames through
    xrt_render_begin / _end into two page-locked host buffers, the device-to-host copy of frame i under the rendering of
    frame i+1.  PCIe-inclusive; reported beside `value`, never as `value`."""
    n = spec.width * spec.height
    lib = xrt.abi.lib()
    bufs = [np.zeros(n, dtype=np.uint32) for _ in range(2)]
    for b in bufs:
        xrt.abi.check(lib.xrt_host_register(C.c_void_p(b.ctypes.data), b.nbytes))
    try:
        fr = [tracer.PrepareHost(b) for b in bufs]
        open_frames = []

        def frame(i):
            open_frames.append((fr[i % 2].begin(), i))
            while len(open_frames) >= 2:
                t, j = open_frames.pop(0)
                fr[j % 2].end(t)

        def drain():
            while open_frames:
                t, j = open_frames.pop(0)
                fr[j % 2].end(t)
        for i in range(warmup):
            frame(i)
        drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            frame(i)
        drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / max(steps, 1)
    finally:
        for b in bufs:
            lib.xrt_host_unregister(C.c_void_p(b.ctypes.data))


def cpu_baseline(spec, budget_s=12.0):
    """The CPU oracle (C++ restatement of the reference's C# path, kind "port") on the host cores, rank 0 at
    N=1 only, on a bounded sample of the same workload: about `budget_s` seconds of single-thread work on centre rows of
    the frame (`value`: the shipped reference renders on one thread, RayTracer.cs:99), then the WHOLE frame once on all cores
    (`value_all_cores`).  Returns the block for the JSON line and the oracle's pixels of that whole frame with the rows they
    cover -- main() compares them with a frame of the timed loop."""
    from oracle import oracle_py as orc
    o = orc.OracleScene(spec)
    H = spec.height
    t0 = time.perf_counter()
    o.render(nthreads=1, rows=(H // 2 - 2, H // 2 + 2), want_float=False)   # probe: 4 centre rows
    probe = max(time.perf_counter() - t0, 1e-4)
    rows = int(min(H, max(4, 4 * budget_s / probe)))
    r0 = max(0, H // 2 - rows // 2)
    reps, rays, dt = 0, 0, 0.0
    t0 = time.perf_counter()
    while True:
        _, _, st = o.render(nthreads=1, rows=(r0, r0 + rows), want_float=False)
        rays += st["rays_closest"] + st["rays_shadow"]
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s * 0.8 or rows < H:
            break
    nt = min(os.cpu_count() or 1, 16)
    # all cores: the whole frame when that fits about the same budget (a C5 frame takes ~4 s on 16 cores), else the same rows
    est_whole = probe * (H / 4.0) / max(nt * 0.8, 1.0)
    rows_m = (0, H) if est_whole <= 3.0 * budget_s else (r0, r0 + rows)
    t0 = time.perf_counter()
    rgba_m, _, stm = o.render(nthreads=nt, rows=rows_m, want_float=False)
    dtm = time.perf_counter() - t0
    rays_m = stm["rays_closest"] + stm["rays_shadow"]
    block = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
             "sample": "rows %d..%d of the %dx%d frame x%d (%d rays, %.1f s) of %s: oracle/ (C++ restatement of the C# path), single thread as the shipped reference (RayTracer.cs:99)"
                       % (r0, r0 + rows, spec.width, H, reps, rays, dt, spec.name),
             "value_all_cores": round(rays_m / dtm / 1e6, 4), "cores_all": nt,
             "sample_all_cores": "rows %d..%d (%d rays, %.1f s)" % (rows_m[0], rows_m[1], rays_m, dtm)}
    return block, rgba_m, rows_m


def parity_block(last_frame, oracle_rgba, rows, width):
    """The headline checks itself: rows `rows` of a frame the timed loop's own render objects produced (read back after the
    timed region) against the oracle's pixels of the same rows -- RGBA8, bit for bit."""
    a = last_frame.reshape(-1, width)[rows[0]:rows[1]]
    b = oracle_rgba.reshape(-1, width)[rows[0]:rows[1]]
    bad = int((a != b).sum())
    return {"parity_rows": [int(rows[0]), int(rows[1])], "parity_ok": bad == 0, "parity_mismatched_pixels": bad,
            "parity_what": "rows of the LAST frame of the timed region (pipelined, two in flight) vs the CPU oracle's render of the same rows, RGBA8 bit for bit"}


def dominant_launch_block(pmc, ms_longest):
    """The launch class that takes most of a frame's traversal time (on every configuration here: the launch of the primary rays), from the PMC passes
    grouped by the launches' position in the frame (tools/pmc_collect.py `launch_classes`) over the LIVE duration of the frame's longest traversal
    launch (xrt_stats.ms_intersect_longest, averaged over the timed frames)."""
    classes = pmc.get("launch_classes") if pmc else None
    if not classes or not ms_longest or ms_longest <= 0:
        return None
    need = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES")
    cand = [(c["SQ_BUSY_CYCLES"], i, c) for i, c in enumerate(classes) if all(k in c for k in need)]
    if not cand:
        return None
    _, idx, c = max(cand)
    t = ms_longest * 1e-3
    issue = c["SQ_INSTS_VALU"] * 64.0 / t / 1e12
    useful = c["SQ_THREAD_CYCLES_VALU"] / t / 1e12
    out = {"launch_position_in_frame": idx, "ms_per_launch": round(ms_longest, 5), "bound": "valu",
           "valu_issue_frac": round(issue / VALU_PEAK_TLANEOPS, 4), "valu_useful_frac": round(useful / VALU_PEAK_TLANEOPS, 4),
           "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0), 3), "valu_instructions": int(c["SQ_INSTS_VALU"])}
    cyc = t * CU_CYCLES_PER_S
    fr = {"valu": out["valu_issue_frac"]}
    for key, name, peak in (("salu", "SQ_INSTS_SALU", SALU_PEAK_IPC), ("branch", "SQ_INSTS_BRANCH", BRANCH_PEAK_IPC), ("smem", "SQ_INSTS_SMEM", SMEM_PEAK_IPC)):
        if name in c:
            out[key + "_frac"] = round(c[name] / cyc / peak, 4)
            fr[key] = out[key + "_frac"]
    if "FETCH_SIZE_KB" in c and "WRITE_SIZE_KB" in c:
        out["traffic"] = int((2.0 * c["FETCH_SIZE_KB"] + c["WRITE_SIZE_KB"]) * 1024)
        out["hbm_frac"] = round(out["traffic"] / t / 1e9 / HBM_PEAK_GBS, 4)
        fr["hbm"] = out["hbm_frac"]
    if c.get("SQ_WAVE_CYCLES"):
        out["waves_waiting_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    out["bound"] = max(fr, key=fr.get)
    out["frac"] = fr[out["bound"]]
    return out


def roofline_block(config, alg_bytes_per_launch, ms_per_launch, launches_per_frame, serial=None, ms_longest=None):
    """Three fractions for the dominant kernel over the same live launch time:
      algorithmic  the REFERENCE algorithm's bytes (un-pruned node / reference / triangle counts) / time / 8 TB/s -- what the
                   north star's target is quoted in; the GPU prunes and caches, so this is an equivalent rate, not a physical one (may exceed 1);
      hbm          HBM bytes the launch really moved (rocprofv3 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) / time / 8 TB/s;
      valu         VALU lane-operations the launch really executed (SQ_THREAD_CYCLES_VALU) / time / 78.6 T lane-op/s, and the
                   instruction-issue slots they occupied (SQ_INSTS_VALU x 64 lanes).
    `bound` names the physical limit the kernel is closest to; achieved / peak / unit / frac are that limit's."""
    t = ms_per_launch * 1e-3
    alg = alg_bytes_per_launch / t / 1e9 if t > 0 else 0.0
    pmc = load_pmc(config)
    fr = {"algorithmic": {"achieved": round(alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / HBM_PEAK_GBS, 4),
                          "bytes_per_launch": int(alg_bytes_per_launch), "what": "reference-algorithm bytes (SURVEY 8d), an equivalent rate: may exceed 1"}}
    traffic = None
    bound, top = "unmeasured (no PMC pass of this kernel build in profiles/pmc_traversal.json)", None
    if pmc and t > 0:
        traffic = int((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024)
        hbm = traffic / t / 1e9
        useful = pmc["SQ_THREAD_CYCLES_VALU"] / t / 1e12
        issue = pmc["SQ_INSTS_VALU"] * 64.0 / t / 1e12
        fr["hbm"] = {"achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4), "bytes_per_launch": traffic}
        fr["valu"] = {"achieved": round(useful, 2), "peak": round(VALU_PEAK_TLANEOPS, 2), "unit": "Tlane-op/s", "frac": round(useful / VALU_PEAK_TLANEOPS, 4),
                      "issue_frac": round(issue / VALU_PEAK_TLANEOPS, 4), "lane_utilisation": round(pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_INSTS_VALU"] * 64.0), 3),
                      "valu_instructions_per_launch": int(pmc["SQ_INSTS_VALU"]), "measured_peak": VALU_MEASURED_TLANEOPS,
                      "issue_frac_of_measured_peak": round(issue / VALU_MEASURED_TLANEOPS, 4),
                      "waves_waiting_frac": round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 3) if pmc.get("SQ_WAVE_CYCLES") else None}
        cands = {"valu": fr["valu"]["issue_frac"], "hbm": fr["hbm"]["frac"]}
        if pmc.get("SQ_INSTS_SALU") is not None:
            # The scalar unit is shared by a CU's four SIMDs.  Its three instruction classes issue side by side (measured, see the
            # constants above), so each is priced against its OWN measured rate; `mix` is the information round 2 reported (the
            # sum) over the rate measured for that very mix.
            cyc = t * CU_CYCLES_PER_S
            n_s, n_b, n_m = pmc["SQ_INSTS_SALU"], pmc.get("SQ_INSTS_BRANCH", 0.0), pmc.get("SQ_INSTS_SMEM", 0.0)
            for key, n, peak in (("salu", n_s, SALU_PEAK_IPC), ("branch", n_b, BRANCH_PEAK_IPC), ("smem", n_m, SMEM_PEAK_IPC)):
                fr[key] = {"achieved": round(n / cyc, 4), "peak": peak, "unit": "instructions per CU per cycle", "frac": round(n / cyc / peak, 4),
                           "issue_frac": round(n / cyc / peak, 4), "instructions_per_launch": int(n)}
                cands[key] = fr[key]["frac"]
            fr["scalar_mix"] = {"achieved": round((n_s + n_b + n_m) / cyc, 4), "peak": MIX_PK_IPC, "unit": "instructions per CU per cycle",
                                "frac": round((n_s + n_b + n_m) / cyc / MIX_PK_IPC, 4),
                                "what": "SALU + branch + SMEM together over the rate measured for k_packet's 80/16/4 mix (information; the classes issue side by side, "
                                        "so the bound is the largest single-class fraction)", "source": "profiles/r03/scalar_mix_peak.txt"}
        bound = max(cands, key=cands.get)
        top = fr[bound]
    out = {"bound": bound, "kernel": "traversal launches of a frame: k_packet (wave-packet form, coherent rays) + k_intersect (per-lane form)",
           "achieved": top["achieved"] if top else None, "peak": top["peak"] if top else None, "unit": top["unit"] if top else None,
           "frac": (top["issue_frac"] if bound in ("valu", "salu", "branch", "smem") else top["frac"]) if top else None,
           # (what a reader of this block alone must see beside `frac`: the share of the VALU peak that did useful lane work -- issue slots x lane
           # utilisation --, and, set by the caller, the rate of the queries that really reach the traversal kernels)
           "useful_frac": fr["valu"]["frac"] if "valu" in fr else None, "lane_utilisation": fr["valu"]["lane_utilisation"] if "valu" in fr else None,
           "traversed_value": None,
           "traffic": traffic, "ms_per_launch": round(ms_per_launch, 5), "launches_per_frame": launches_per_frame, "fractions": fr,
           "pmc_build_id": build_id() if pmc else None,
           "note": "branchy scalar fp32 traversal, scene resident in the 256 MiB Infinity Cache: not HBM-bound. `frac` is the largest PHYSICAL fraction -- "
                   "hbm (bytes moved / 8 TB/s), valu (issue slots used of 78.6 T lane-op/s; fractions.valu.frac of them did useful lane work), salu / branch / smem "
                   "(each class over its own measured issue rate: they issue side by side, profiles/r03/scalar_mix_peak.txt); fractions.algorithmic is the "
                   "SURVEY 8d equivalent rate of the un-pruned reference algorithm, not a physical fraction; durations = every launch of the timed region "
                   "times itself on the device clock, first wave's start to last wave's end (events on the dispatch packets cost ~5 us a launch; "
                   "rocprofv3's dispatch-level durations are ~4 us per launch longer); PMC = rocprofv3 passes of this build (profiles/), per launch"}
    # The traversal launches of a frame are of different kinds (round 4: the launch of a terrain's shadow rays and reflections mostly ends at the
    # mesh's normal box), so the averages above describe no launch in particular: `dominant_launch` is the class that takes most of the time.
    dom = dominant_launch_block(pmc, ms_longest)
    if dom:
        out["dominant_launch"] = dom
        out["all_launches"] = {"bound": out["bound"], "frac": out["frac"], "useful_frac": out["useful_frac"], "lane_utilisation": out["lane_utilisation"],
                               "ms_per_launch": out["ms_per_launch"], "what": "the same quantities averaged over every traversal launch of a frame"}
        out["bound"], out["frac"] = dom["bound"], dom["frac"]
        out["achieved"], out["peak"], out["unit"] = (round(dom["valu_issue_frac"] * VALU_PEAK_TLANEOPS, 2), round(VALU_PEAK_TLANEOPS, 2), "Tlane-op/s") if dom["bound"] == "valu" else (out["achieved"], out["peak"], out["unit"])
        out["useful_frac"], out["lane_utilisation"] = dom["valu_useful_frac"], dom["lane_utilisation"]
        out["traffic"] = dom.get("traffic", out["traffic"])
    if serial:
        if pmc and serial.get("ms_per_launch"):   # the same counters over the duration of a launch that has the GPU to itself
            ts = serial["ms_per_launch"] * 1e-3
            serial["fractions"] = {"hbm": round(traffic / ts / 1e9 / HBM_PEAK_GBS, 4),
                                   "valu_issue": round(pmc["SQ_INSTS_VALU"] * 64.0 / ts / 1e12 / VALU_PEAK_TLANEOPS, 4),
                                   "valu_useful": round(pmc["SQ_THREAD_CYCLES_VALU"] / ts / 1e12 / VALU_PEAK_TLANEOPS, 4)}
            for key, peak in (("salu", SALU_PEAK_IPC), ("branch", BRANCH_PEAK_IPC), ("smem", SMEM_PEAK_IPC)):
                if key in fr:
                    serial["fractions"][key] = round(fr[key]["instructions_per_launch"] / (ts * CU_CYCLES_PER_S) / peak, 4)
        if dom and serial.get("ms_longest"):
            d2 = dominant_launch_block(pmc, serial["ms_longest"])
            if d2:
                serial["dominant_launch"] = {k: d2[k] for k in ("ms_per_launch", "bound", "frac", "valu_issue_frac", "valu_useful_frac") if k in d2}
        out["serialised"] = serial
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)   # (60 frames of ~4 ms: the timed region is a quarter of a second; 20 left +-2 % of noise in it)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C5", choices=list(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the image (debug only; invalid as a benchmark)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements of the other configs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-host", action="store_true", help="skip the host-output (PCIe-inclusive) timing")
    ap.add_argument("--in-library", action="store_true",
                    help="N > 1 from ONE process: xrt_render_opts.n_gpus = N (replicas, per-device host threads, grouped RCCL send/recv and "
                         "de-tile inside libxrt) -- the path the C# host's single RenderInternal call would use; run as plain `python bench.py "
                         "--gpus N --in-library` (no torch.distributed launcher)")
    args = ap.parse_args()

    rank, local_rank, world = xrt.dist.env_rank_world()
    in_library = args.in_library and args.gpus > 1
    if in_library and world > 1:
        print("bench.py: --in-library is one process driving N GPUs; start it without torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and world == 1 and not in_library:
        print("bench.py: --gpus %d needs one process per GPU: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
              "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("XRT_DIST_BACKEND", "nccl")   # "gloo": rehearse the N>1 path when ranks share a GPU
    ndev = torch.cuda.device_count()
    if backend != "nccl" and ndev > 0:
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    res, spec = run_config(args.config, args.scale, args.steps, args.warmup, rank, local_rank, world, n_gpus_in_library=args.gpus if in_library else 1)
    # max over ranks of the timed region; total rays over ranks
    t = torch.tensor([res["seconds"]], dtype=torch.float64, device="cuda")
    r = torch.tensor([float(res["rays"]), float(intersect_bytes(res["stats"])), res["ms_intersect"], float(res["launches"]), float(res["stats"]["rays_traversed"])],
                     dtype=torch.float64, device="cuda")
    if world > 1:
        if dist.get_backend() != "nccl":
            t, r = t.cpu(), r.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rsum = r.clone()
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
    else:
        rsum = r
    seconds = float(t.item())
    rays_frame = float(rsum[0].item())
    value = rays_frame * args.steps / seconds / 1e6
    trav_frame = float(rsum[4].item())   # queries handed to the traversal kernels (xrt_stats.rays_traversed): the rest are primary rays that miss the scene's root box

    parity_failed = False
    if rank == 0:
        st = res["stats"]
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        # roofline of the dominant kernel (k_intersect), this rank: per launch over the timed region
        launches = max(res["launches"], 1)
        bytes_per_launch = intersect_bytes(st) * args.steps / launches
        ms_per_launch = res["ms_intersect"] / launches
        serial = None
        if res.get("overlapped"):
            l_s = max(res["serial_launches"], 1)
            ms_l = res["serial_ms_intersect"] / l_s
            ach_s = intersect_bytes(st) * res["serial_steps"] / l_s / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0
            serial = {"what": "the timed frames overlapped pairwise on two streams, so ms_per_launch includes time shared with the other frame; this is the same launch with the GPU to itself",
                      "ms_per_launch": round(ms_l, 5), "algorithmic_achieved": round(ach_s, 2), "ms_per_step": round(res["serial_seconds"] / res["serial_steps"] * 1e3, 4),
                      "ms_longest": res.get("serial_ms_longest")}
        line = {
            "metric": metric if args.scale == 1.0 else "Mrays/sec (scaled image, not a benchmark)",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": args.gpus if in_library else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(seconds / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config], "width": res["width"], "height": res["height"], "triangles": res["tris"],
                       "instances": res["instances"], "rays_per_frame": int(rays_frame),
                       "rays_closest_per_frame": int(st["rays_closest"]) if world == 1 else None, "rays_shadow_per_frame": int(st["rays_shadow"]) if world == 1 else None,
                       "rays_traversed_per_frame": int(trav_frame),
                       "rays_answered_by_raygen_per_frame": int(rays_frame - trav_frame),
                       # ... and of the traversed ones, the mesh queries the mesh's normal box answers (every triangle faces away from the ray, RE:48-51:
                       # the shadow rays and reflections of this terrain): counted by the untimed counting pass (xrt_stats.mesh_queries_facing_away)
                       "mesh_queries_answered_by_normal_box_per_frame": int(st.get("mesh_queries_facing_away", 0)) if world == 1 else None,
                       "parallelism": ("image tiles 64x8 x%d (xrt_render_opts.balance_tiles: dealt by the previous frame's tile costs), one process: xrt_render_opts.n_gpus (in-library RCCL send/recv gather)" % args.gpus) if in_library
                                      else ("image tiles 64x8 x%d, dealt longest-first by the previous frames' tile costs (xrt_balance_tiles; balance by cost %.3f, round-robin %.3f)"
                                            % (world, res["tile_balance"]["by_cost_table"], res["tile_balance"]["by_cost_round_robin"]) if res.get("tile_balance")
                                            else "image tiles 64x8 round-robin x%d" % world), "scene_build_s": round(res["build_s"], 3), "device_memory": res["device_memory"]},
            "Mrays_per_s_traversed": round(trav_frame * args.steps / seconds / 1e6, 3),
            "roofline": roofline_block(args.config if (world == 1 and not in_library) else "(N > 1: no PMC pass)", bytes_per_launch, ms_per_launch, launches // max(args.steps, 1), serial,
                                       ms_longest=res.get("ms_longest")),
        }
        line["roofline"]["traversed_value"] = line["Mrays_per_s_traversed"]   # Mrays/s of the queries that reach the traversal kernels (`value` counts the primary rays k_raygen answers too)
        solo = world == 1 and not in_library   # the side measurements below are single-GPU figures
        if solo and not args.no_host and args.scale == 1.0:
            try:
                _, tracer_h = xrt.configs.build_product(spec, device=local_rank)
                # (a fresh scene: six warm-up frames size both frame contexts, create their streams and let the library see a frame time)
                hs = time_host_output(tracer_h, spec, max(2, min(args.steps, 10)), 6)
                line["ms_per_step_host_output"] = round(hs * 1e3, 4)
                line["host_output"] = {"what": "xrt_render_begin/_end into two page-locked host Color[] buffers (RT:122-123), D2H of frame i under frame i+1; PCIe-inclusive, never `value`",
                                       "Mrays_per_s": round(rays_frame / hs / 1e6, 2)}
                del tracer_h
            except Exception as e:   # a side measurement must not take the headline down
                line["host_output"] = {"error": str(e)[:200]}
        if solo and args.scale == 1.0:
            try:   # what the C# host's blocking RenderInternal sees: one frame at a time (libxrt splits it over two streams)
                _, tracer_b = xrt.configs.build_product(spec, device=local_rank)
                outb = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
                frb = tracer_b.PrepareDevice(outb.data_ptr())
                for _ in range(3):
                    frb()
                torch.cuda.synchronize()
                kb = max(2, min(args.steps, 10))
                t0 = time.perf_counter()
                for _ in range(kb):
                    stb = frb()
                torch.cuda.synchronize()
                line["ms_per_step_blocking"] = round((time.perf_counter() - t0) / kb * 1e3, 4)
                line["blocking_pieces"] = int(stb["pieces"])
                del tracer_b, outb
            except Exception as e:
                line["ms_per_step_blocking"] = None
        if solo and args.scale == 1.0:
            try:   # THE seam, literally (RT:103-126): one blocking xrt_render per frame, ending in the host's Color[] (page-locked)
                _, tracer_x = xrt.configs.build_product(spec, device=local_rank)
                hostb = np.zeros(spec.width * spec.height, dtype=np.uint32)
                lib = xrt.abi.lib()
                xrt.abi.check(lib.xrt_host_register(C.c_void_p(hostb.ctypes.data), hostb.nbytes))
                try:
                    fx = tracer_x.PrepareHost(hostb)
                    for _ in range(3):
                        fx.end(fx.begin())
                    kb = max(2, min(args.steps, 10))
                    t0 = time.perf_counter()
                    for _ in range(kb):
                        fx.end(fx.begin())   # begin + end back to back = the blocking call: nothing else in flight
                    line["ms_per_step_blocking_host"] = round((time.perf_counter() - t0) / kb * 1e3, 4)
                    if res.get("last_frame") is not None:
                        line["blocking_host_equals_timed_frame"] = bool(np.array_equal(hostb, res["last_frame"]))
                finally:
                    lib.xrt_host_unregister(C.c_void_p(hostb.ctypes.data))
                del tracer_x
            except Exception as e:
                line["ms_per_step_blocking_host"] = None
                line["blocking_host_error"] = str(e)[:200]
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"], oracle_rgba, oracle_rows = cpu_baseline(spec)
            if res.get("last_frame") is not None:
                line.update(parity_block(res["last_frame"], oracle_rgba, oracle_rows, spec.width))
                parity_failed = not line["parity_ok"]
        if solo and not args.no_extra and args.scale == 1.0:
            other = {}
            for name, k in (("C2", 200), ("C3", 30), ("C4", 12), ("G1", 60)):   # (frames of 0.1-0.3 ms: enough of them for the clocks to have ramped up)   # G1: the scene the reference's own Stopwatch would time (Game1.cs)
                if name == args.config:
                    continue
                try:
                    r2, _ = run_config(name, 1.0, k, 3, 0, local_rank, 1)   # (enough frames for the fill and drain of two in flight not to show in the period)
                    # per-launch figures from the serialised pass when the timed frames overlapped
                    ms_i, l2, kk = (r2["serial_ms_intersect"], max(r2["serial_launches"], 1), r2["serial_steps"]) if r2["overlapped"] else (r2["ms_intersect"], max(r2["launches"], 1), k)
                    rb = roofline_block(name, intersect_bytes(r2["stats"]) * kk / l2, ms_i / l2, l2 // max(kk, 1),
                                        ms_longest=r2.get("serial_ms_longest") if r2["overlapped"] else r2.get("ms_longest"))
                    other[name] = {"workload": WORKLOADS[name], "Mrays_per_s": round(r2["rays"] * k / r2["seconds"] / 1e6, 2),
                                   "ms_per_step": round(r2["seconds"] / k * 1e3, 3), "rays_per_frame": int(r2["rays"]),
                                   "frames_overlap": bool(r2["overlapped"]), "ms_per_launch": round(ms_i / l2, 4),
                                   "bound": rb["bound"], "frac": rb["frac"], "dominant_launch": rb.get("dominant_launch"), "fractions": rb["fractions"], "scene_build_s": round(r2["build_s"], 2)}
                    if r2["overlapped"]:
                        other[name]["ms_per_step_serialised"] = round(r2["serial_seconds"] / kk * 1e3, 3)
                    # one blocking xrt_render_device at a time on the library's own streams (what RenderInternal, RT:103-126, sees: two-level scenes are rendered as two bands)
                    _, tr_b = xrt.configs.build_product(xrt.configs.config(name), device=local_rank)
                    sp_b = xrt.configs.config(name)
                    ob = torch.zeros(sp_b.width * sp_b.height, dtype=torch.int32, device="cuda")
                    fb = tr_b.PrepareDevice(ob.data_ptr())
                    for _ in range(4):
                        fb()
                    torch.cuda.synchronize()
                    tb0 = time.perf_counter()
                    for _ in range(10):
                        fb()
                    torch.cuda.synchronize()
                    other[name]["ms_per_step_blocking"] = round((time.perf_counter() - tb0) / 10 * 1e3, 3)
                    del tr_b, fb, ob
                except Exception as e:   # a side measurement must not take the headline down
                    other[name] = {"error": str(e)[:200]}
            line["other_configs"] = other
        print(json.dumps(line))
        sys.stdout.flush()
        if parity_failed:
            print("bench.py: the timed loop's frame differs from the oracle in %d pixels of rows %s: the measurement is void"
                  % (line["parity_mismatched_pixels"], line["parity_rows"]), file=sys.stderr)
        try:   # per-run metrics file: the line plus the frame's accounting
            mdir = os.environ.get("XRT_METRICS_DIR", os.path.join(ROOT, "gpurun_out"))
            os.makedirs(mdir, exist_ok=True)
            with open(os.path.join(mdir, "bench_metrics_%s_n%d.json" % (args.config, world)), "w") as f:
                json.dump({"line": line, "frame_stats": {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in st.items()},
                           "argv": sys.argv[1:], "kernel_build_id": build_id(), "time": time.strftime("%Y-%m-%dT%H:%M:%S")}, f, indent=1)
        except Exception:
            pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
