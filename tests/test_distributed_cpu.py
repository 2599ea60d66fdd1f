"""world_size-2 (and 3) gloo test of the N>1 path on the CPU: image-tile sharding, the framebuffer gather
(the path's only exchange step) and the de-tile — with the oracle's frame standing in for the GPU render."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_table(rank, world, port, width, height, q):
    """The same exchange under a cost-aware tile table (xrt.h xrt_scene_set_tile_table): every rank holds the per-tile costs of ITS tiles,
    an all-reduce sums them, every rank computes the same longest-first table (xrt_balance_tiles, host arithmetic), packs its row of it."""
    sys.path.insert(0, ROOT)
    xrt = importlib.import_module("xna-ray-trace_amd")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx, ty, tpr = xrt.dist.shard_layout(width, height, world)
        rng = np.random.default_rng(5)
        frame = rng.integers(0, 2**32, size=width * height, dtype=np.uint32)
        cost_all = (rng.random(tx * ty) ** 3 * 1000 + 1).astype(np.float32)
        mine = torch.zeros(tx * ty)
        t = np.arange(tx * ty)
        mine[torch.from_numpy(t[t % world == rank])] = torch.from_numpy(cost_all[t % world == rank])   # what xrt_scene_tile_costs gives a round-robin rank
        dist.all_reduce(mine)
        tprb, table = xrt.dist.balanced_table(width, height, world, mine.numpy())
        ok = not np.array_equal(table, np.pad(xrt.dist.round_robin_table(width, height, world)[1].reshape(world, -1), ((0, 0), (0, tprb - tpr)), constant_values=-1).reshape(-1))
        local = torch.from_numpy(xrt.dist.pack_shard(frame, width, height, rank, world, table, tprb).view(np.int32).copy())
        gathered = xrt.dist.gather_frame(local, width, height)
        tables = [None] * world
        dist.all_gather_object(tables, table.tolist())
        ok = ok and all(tb == tables[0] for tb in tables)   # every rank computed the same table
        if rank == 0:
            out = xrt.dist.detile_host(gathered.numpy().view(np.uint32), width, height, world, table=table, tiles_per_rank=tprb)
            loads = np.array([cost_all[r[r >= 0]].sum() for r in table.reshape(world, tprb)])
            rr = np.array([cost_all[t % world == r].sum() for r in range(world)])
            q.put(bool(ok and np.array_equal(out, frame) and loads.mean() / loads.max() >= rr.mean() / rr.max() and loads.mean() / loads.max() > 0.97))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, width, height, q):
    sys.path.insert(0, ROOT)
    xrt = importlib.import_module("xna-ray-trace_amd")
    from oracle import oracle_py as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = xrt.configs.crate_grid_scene(width, height)
        frame = orc.OracleScene(spec).render(nthreads=1, want_float=False)[0]   # every rank holds the replicated scene
        local = torch.from_numpy(xrt.dist.pack_shard(frame, width, height, rank, world).view(np.int32).copy())
        gathered = xrt.dist.gather_frame(local, width, height)
        ok = True
        if rank == 0:
            out = xrt.dist.detile_host(gathered.numpy().view(np.uint32), width, height, world)
            ok = bool(np.array_equal(out, frame))
        # one gather carrying the tiles of M frames (bench.py, N > 1): frame m of rank r sits at r * M * n + m * n
        M, n = 3, local.numel()
        frames = [frame, frame[::-1].copy(), (frame ^ np.uint32(0x00ff00ff)).astype(np.uint32)]
        group = torch.cat([torch.from_numpy(xrt.dist.pack_shard(f, width, height, rank, world).view(np.int32).copy()) for f in frames])
        got = xrt.dist.gather_frame_async(group, recv=torch.empty(world * M * n, dtype=torch.int32) if rank == 0 else None)()
        if rank == 0:
            for m, f in enumerate(frames):
                out = xrt.dist.detile_host(got.numpy().view(np.uint32), width, height, world, rank_stride=M * n, offset=m * n)
                ok = ok and bool(np.array_equal(out, f))
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height", [(2, 128, 32), (3, 100, 37)])
def test_tile_shard_gather_detile_gloo(world, width, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world,width,height", [(2, 640, 160), (3, 500, 133)])
def test_balanced_tile_table_gloo(world, width, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_table, args=(r, world, port, width, height, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_balance_tiles_properties():
    """xrt_balance_tiles (host arithmetic, no device): every tile exactly once, no rank over its slots, a rank's slots in descending order of cost, a
    better balance than round-robin on skewed costs, round-robin without costs, deterministic, and the refusals."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    xrt = importlib.import_module("xna-ray-trace_amd")
    rng = np.random.default_rng(3)
    for (w, h, n) in ((1920, 1080, 8), (3840, 2160, 8), (1920, 1080, 2), (100, 37, 3), (64, 8, 4)):
        tx, ty, tpr = xrt.dist.shard_layout(w, h, n)
        cost = (rng.random(tx * ty) ** 4 * 1e6).astype(np.float32)
        cost[: tx * ty // 5] *= 30          # an expensive band (the horizon rows)
        cost[rng.integers(0, tx * ty, size=3)] = 0.0
        tprb, table = xrt.dist.balanced_table(w, h, n, cost)
        rows = table.reshape(n, tprb)
        assert np.array_equal(np.sort(table[table >= 0]), np.arange(tx * ty))
        floor = cost[cost > 0].min() if (cost > 0).any() else 0.0
        for r in rows:
            used = r[r >= 0]
            assert np.all(r[len(used):] == -1)
            assert np.all(np.diff(np.maximum(cost[used], floor)) <= 0)   # a rank's slots in descending order of cost: its launches start with their longest packets
        if tx * ty >= 4 * n:
            loads = np.array([cost[r[r >= 0]].sum() for r in rows])
            rr = np.array([cost[np.arange(tx * ty) % n == k].sum() for k in range(n)])
            assert loads.mean() / loads.max() >= rr.mean() / rr.max()
            if tx * ty >= 100 * n:
                assert loads.mean() / loads.max() > 0.99
        assert np.array_equal(xrt.dist.balanced_table(w, h, n, cost)[1], table)
        assert np.array_equal(xrt.dist.balanced_table(w, h, n, np.zeros(tx * ty, np.float32), slack=0.0)[1], xrt.dist.round_robin_table(w, h, n)[1])
    out = (C.c_int32 * 8)()
    assert xrt.abi.lib().xrt_balance_tiles(1920, 1080, 8, None, 10, out) == xrt.abi.XRT_E_INVALID_ARG   # 80 slots for 4050 tiles
    assert b"do not hold" in xrt.abi.lib().xrt_last_error()


def test_shard_layout_matches_library():
    import ctypes as C
    sys.path.insert(0, ROOT)
    xrt = importlib.import_module("xna-ray-trace_amd")
    for (w, h, n) in ((1920, 1080, 8), (3840, 2160, 8), (100, 37, 3), (64, 8, 4), (65, 9, 2)):
        tx, ty, tpr = C.c_int32(), C.c_int32(), C.c_int32()
        assert xrt.abi.lib().xrt_shard_layout(w, h, n, C.byref(tx), C.byref(ty), C.byref(tpr)) == 0
        assert (tx.value, ty.value, tpr.value) == xrt.dist.shard_layout(w, h, n)
    # pack/detile round trip covers every pixel exactly once
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 2**32, size=100 * 37, dtype=np.uint32)
    parts = np.concatenate([xrt.dist.pack_shard(frame, 100, 37, r, 3) for r in range(3)])
    assert np.array_equal(xrt.dist.detile_host(parts, 100, 37, 3), frame)
