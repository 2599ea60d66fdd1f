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
