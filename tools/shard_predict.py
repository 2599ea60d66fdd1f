"""One-GPU PREDICTION of image-tile strong scaling (VERDICT r2 #4a) -- not a measurement of N GPUs.

    python tools/shard_predict.py [C4 C5 ...] [--out gpurun_out/shard_predict.json]

For N = 2, 4, 8 every one of the N tile shards of a frame (tile t -> rank t % N) is rendered on the ONE device, alone,
as blocking frames (xrt_render_device with shard_rank / shard_count): its GPU time (xrt_stats.ms_total, median of the timed
frames), its rays and the rays that reach the traversal kernels.  Reported per configuration and N:
    t_whole_ms                     the unsharded frame, same protocol
    t_shard_ms[r]                  shard r alone on the device
    predicted_strong_scaling       t_whole / max_r t_shard[r]           (what N devices would give if nothing else cost time)
    balance                        mean_r t_shard / max_r t_shard       (1 = perfectly even tiles)
    fixed_ms                       per-frame costs that do not shrink with N: the de-tile kernel over the gathered buffers (measured
                                   here) and the gather itself, ESTIMATED as one rank's tile buffer over one xGMI link at 153 GB/s (every rank
                                   has a link of its own to rank 0, the transfers run side by side) plus 10 us of launch latency
                                   (MI355X_MICROARCH.md has no measured RCCL figure)
    predicted_with_fixed           t_whole / (max_r t_shard[r] + fixed_ms)
    period_*, predicted_throughput_scaling[_with_fixed]   the same with the frame PERIOD of two frames in flight (what bench.py --gpus N times)
A shard of 1/N of the tiles is NOT 1/N of the time: launches of persistent waves have a floor (5-6 us each, ten per frame) and a
tail that does not shrink, which is exactly what this tool is for."""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

xrt = importlib.import_module("xna-ray-trace_amd")
XGMI_LINK_GBS = 153.0


def timed_deep(frs, reps):
    """Frame period with len(frs) frames in flight: the render objects belong to len(frs) / 2 scene objects (two tickets each), frame i + depth - 1 is
    enqueued before frame i is waited for."""
    depth = len(frs)
    for f in frs:
        f(); f()
    k = max(3 * depth, 2 * reps)
    open_t = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        open_t.append((i % depth, frs[i % depth].begin()))
        if len(open_t) >= depth:
            j, t = open_t.pop(0)
            frs[j].end(t)
    while open_t:
        j, t = open_t.pop(0)
        frs[j].end(t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def timed(frs, reps):
    """Median GPU time of a blocking frame (xrt_stats.ms_total: the latency one call sees) and the period of the same frame rendered the way
    bench.py times it: two tickets open on two render objects with an output buffer each, frame i+1 enqueued before frame i is waited for
    (launch tails of one frame fill with the other's work)."""
    fr = frs[0]
    for _ in range(3):
        st = fr()
    ms = []
    for _ in range(reps):
        st = fr()
        ms.append(st["ms_total"])
    k = max(6, 2 * reps)
    for f in frs:   # both frame contexts warm
        f()
    t_open = frs[0].begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, k + 1):
        t_next = frs[i % 2].begin()
        frs[(i - 1) % 2].end(t_open)
        t_open = t_next
    frs[k % 2].end(t_open)
    torch.cuda.synchronize()
    return statistics.median(ms), st, (time.perf_counter() - t0) / (k + 1) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C4", "C5"])
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--n", default="2,4,8", help="shard counts")
    ap.add_argument("--deep", type=int, default=0, help="also time every shard with this many frames in flight (2 per scene object: 4 = two scene objects per rank)")
    ap.add_argument("--layouts", default="balanced", help="comma list of round_robin, balanced (tiles dealt longest-first by the whole frame's tile costs: xrt_balance_tiles)")
    ap.add_argument("--out", default=os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "shard_predict.json"))
    args = ap.parse_args()
    out = {"what": "one-GPU PREDICTION of image-tile strong scaling: every shard rendered alone on one MI355X; not a multi-GPU measurement",
           "xgmi_link_GBs_assumed": XGMI_LINK_GBS, "configs": {}}
    for name in args.configs:
        spec = xrt.configs.config(name)
        scene, tracer = xrt.configs.build_product(spec)
        W, H = spec.width, spec.height
        whole = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        whole2 = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        t_whole, st_whole, p_whole = timed([tracer.PrepareDevice(whole.data_ptr()), tracer.PrepareDevice(whole2.data_ptr())], args.reps)
        cfg = {"width": W, "height": H, "t_whole_ms": round(t_whole, 4), "period_whole_ms": round(p_whole, 4), "rays_whole": int(st_whole["rays_closest"] + st_whole["rays_shadow"]),
               "rays_traversed_whole": int(st_whole["rays_traversed"]), "shards": {}}
        extra = []   # more scene objects of the same scene on this device (frames in flight beyond two)
        for _ in range(max(0, args.deep // 2 - 1)):
            extra.append(xrt.configs.build_product(spec))
        if args.deep:
            frs = []
            for sc, tr in [(scene, tracer)] + extra:
                frs += [tr.PrepareDevice(torch.zeros(W * H, dtype=torch.int32, device="cuda").data_ptr()) for _ in range(2)]
            cfg["period_whole_deep_ms"] = round(timed_deep(frs, args.reps), 4)
        tracer.TileCosts(reset=True)
        for _ in range(2):
            tracer.RenderDevice(whole.data_ptr())
        cost = tracer.TileCosts(reset=True)   # what every rank's xrt_scene_tile_costs add up to after two frames
        for n, layout in [(int(x), l) for x in args.n.split(",") for l in args.layouts.split(",")]:
            import ctypes as C
            c_tpr = C.c_int32()
            xrt.abi.check(xrt.abi.lib().xrt_shard_layout(W, H, n, None, None, C.byref(c_tpr)))   # (the library's own layout: variants may differ)
            tpr = c_tpr.value
            table_dev, by_cost = None, None
            if layout == "balanced":
                tpr, table = xrt.dist.balanced_table(W, H, n, cost)
                tracer.SetTileTable(n, tpr, table)
                for sc, tr in extra:
                    tr.SetTileTable(n, tpr, table)
                table_dev = torch.from_numpy(table).cuda()
                loads = [float(cost[r[r >= 0]].sum()) for r in table.reshape(n, tpr)]
                by_cost = sum(loads) / n / max(loads)
            else:
                tracer.SetTileTable(n, tpr, None)
                for sc, tr in extra:
                    tr.SetTileTable(n, tpr, None)
            count = tpr * 512
            gathered = torch.zeros(n * count, dtype=torch.int32, device="cuda")
            second = torch.zeros(count, dtype=torch.int32, device="cuda")   # output of the other frame in flight
            ts, ps, rays, trav, deep = [], [], [], [], []
            for r in range(n):
                if args.deep:
                    frs = []
                    for sc, tr in [(scene, tracer)] + extra:
                        frs += [tr.PrepareDevice(torch.zeros(count, dtype=torch.int32, device="cuda").data_ptr(), shard_rank=r, shard_count=n) for _ in range(2)]
                    deep.append(timed_deep(frs, args.reps))
                t, st, per = timed([tracer.PrepareDevice(gathered[r * count:(r + 1) * count].data_ptr(), shard_rank=r, shard_count=n),
                                    tracer.PrepareDevice(second.data_ptr(), shard_rank=r, shard_count=n)], args.reps)
                tracer.PrepareDevice(gathered[r * count:(r + 1) * count].data_ptr(), shard_rank=r, shard_count=n)()   # (the shard's pixels for the de-tile check below)
                ts.append(t); ps.append(per); rays.append(int(st["rays_closest"] + st["rays_shadow"])); trav.append(int(st["rays_traversed"]))
            final = torch.zeros(W * H, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dts = []
            for _ in range(5):
                e0.record()
                xrt.dist.detile_device(gathered, W, H, n, final, table_dev=table_dev, tiles_per_rank=tpr)
                e1.record()
                torch.cuda.synchronize()
                dts.append(e0.elapsed_time(e1))
            assert torch.equal(final, whole), "sharded frame differs from the whole frame"
            detile_ms = statistics.median(dts)
            gather_ms = (count * 4) / (XGMI_LINK_GBS * 1e9) * 1e3 + 0.010   # every rank sends its tile buffer over its OWN link to rank 0 (point-to-point xGMI): the n-1 transfers run side by side
            fixed = detile_ms + gather_ms
            cfg["shards"][str(n) if layout == "round_robin" else "%d_%s" % (n, layout)] = {"layout": layout, "tiles_per_rank": tpr, "balance_by_tile_cost": None if by_cost is None else round(by_cost, 4), "t_shard_ms": [round(t, 4) for t in ts], "period_shard_ms": [round(t, 4) for t in ps],
                                     "predicted_throughput_scaling": round(p_whole / max(ps), 3),
                                     "predicted_throughput_scaling_with_fixed": round(p_whole / (max(ps) + fixed), 3), "rays": rays, "rays_traversed": trav,
                                     "predicted_strong_scaling": round(t_whole / max(ts), 3), "balance": round(sum(ts) / n / max(ts), 3),
                                     "sum_of_shards_over_whole": round(sum(ts) / t_whole, 3),
                                     "fixed_ms": {"detile_measured": round(detile_ms, 4), "gather_estimated": round(gather_ms, 4)},
                                     "predicted_with_fixed": round(t_whole / (max(ts) + fixed), 3)}
            if deep:
                cfg["shards"][str(n) if layout == "round_robin" else "%d_%s" % (n, layout)].update({"frames_in_flight_deep": args.deep, "period_shard_deep_ms": [round(t, 4) for t in deep],
                    "predicted_throughput_scaling_deep": round(cfg["period_whole_deep_ms"] / max(deep), 3), "predicted_throughput_scaling_deep_with_fixed": round(cfg["period_whole_deep_ms"] / (max(deep) + fixed), 3),
                    "predicted_throughput_scaling_deep_vs_two_in_flight_whole": round(p_whole / (max(deep) + fixed), 3)})
                print("%s N=%d %s: %d frames in flight: whole %.3f ms, shards %s ms -> x%.2f (with fixed x%.2f; against the whole frame with two in flight x%.2f)" % (
                    name, n, layout, args.deep, cfg["period_whole_deep_ms"], " ".join("%.3f" % t for t in deep), cfg["period_whole_deep_ms"] / max(deep),
                    cfg["period_whole_deep_ms"] / (max(deep) + fixed), p_whole / (max(deep) + fixed)), flush=True)
            print("%s N=%d %s: blocking: whole %.3f ms, shards %s ms -> predicted x%.2f (with fixed costs x%.2f), balance %.2f | two in flight: whole %.3f ms, shards %s ms -> x%.2f (with fixed x%.2f)" % (
                name, n, layout, t_whole, " ".join("%.3f" % t for t in ts), t_whole / max(ts), t_whole / (max(ts) + fixed), sum(ts) / n / max(ts),
                p_whole, " ".join("%.3f" % t for t in ps), p_whole / max(ps), p_whole / (max(ps) + fixed)), flush=True)
        tracer.SetTileTable(2, 1, None)
        out["configs"][name] = cfg
        del tracer, scene
    out["time"] = time.strftime("%Y-%m-%dT%H:%M:%S")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
