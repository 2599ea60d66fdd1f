"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards by image tiles: the scene is replicated, tile t (64x8 pixels, row-major
numbering) belongs to rank t % world and lands in slot t // world of that rank's contiguous buffer.  The
only exchange step of a frame is the gather of those buffers on rank 0, followed by a de-tile kernel.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as abi


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_layout(width, height, world):
    """(tiles_x, tiles_y, tiles_per_rank) — pure arithmetic, mirrors xrt_shard_layout."""
    tx = (width + abi.TILE_W - 1) // abi.TILE_W
    ty = (height + abi.TILE_H - 1) // abi.TILE_H
    return tx, ty, (tx * ty + world - 1) // world


def _slot_xy():
    """x, y of the 512 path slots of a tile: eight 8x8-pixel blocks side by side, Z-order inside a block (xrt_core.h tile_slot_xy)."""
    w = np.arange(512)
    blk, i = w >> 6, w & 63
    x = blk * 8 + ((i & 1) | ((i >> 1) & 2) | ((i >> 2) & 4))
    y = ((i >> 1) & 1) | ((i >> 2) & 2) | ((i >> 3) & 4)
    return x, y


_SLOT_X, _SLOT_Y = _slot_xy()


def _tile_to_slots(tile):
    """(TILE_H, TILE_W) pixels -> the 512 path slots of a tile."""
    return tile[_SLOT_Y, _SLOT_X]


def _slots_to_tile(slots):
    tile = np.zeros((abi.TILE_H, abi.TILE_W), dtype=slots.dtype)
    tile[_SLOT_Y, _SLOT_X] = slots
    return tile


def balanced_table(width, height, world, tile_cost, slack=0.25):
    """xrt_balance_tiles: (tiles_per_rank, table[world * tiles_per_rank]) -- the frame's tiles dealt longest-first by `tile_cost` (one float
    per tile, row-major; None or all zeros: round-robin), with `slack` more slots per rank than an even deal needs.  Deterministic:
    every rank computes the same table from the same costs (bench.py all-reduces them first)."""
    tx, ty, tpr = shard_layout(width, height, world)
    tprb = tpr + int(np.ceil(tpr * slack))
    table = np.full(world * tprb, -1, dtype=np.int32)
    cost = None if tile_cost is None else np.ascontiguousarray(tile_cost, dtype=np.float32)
    if cost is not None and cost.size != tx * ty:
        raise ValueError("tile_cost has %d entries, the frame has %d tiles" % (cost.size, tx * ty))
    abi.check(abi.lib().xrt_balance_tiles(width, height, world, None if cost is None else cost.ctypes.data_as(C.POINTER(C.c_float)), tprb,
                                          table.ctypes.data_as(C.POINTER(C.c_int32))))
    return tprb, table


def round_robin_table(width, height, world):
    """The default layout (tile t -> rank t % world, slot t // world) written as a table."""
    tx, ty, tpr = shard_layout(width, height, world)
    table = np.full(world * tpr, -1, dtype=np.int32)
    t = np.arange(tx * ty)
    table[(t % world) * tpr + t // world] = t
    return tpr, table


def pack_shard(frame, width, height, rank, world, table=None, tiles_per_rank=None):
    """Host mirror of what xrt_render_device writes for a shard: frame (H*W uint32) -> this rank's
    tile-contiguous buffer (tiles_per_rank*512).  Used by the CPU (gloo) tests.  table / tiles_per_rank: an installed tile table."""
    tx, ty, tpr = shard_layout(width, height, world)
    if table is not None:
        tpr = int(tiles_per_rank)
    img = np.asarray(frame, dtype=np.uint32).reshape(height, width)
    out = np.zeros(tpr * 512, dtype=np.uint32)
    for slot in range(tpr):
        t = slot * world + rank if table is None else int(table[rank * tpr + slot])
        if t < 0:
            continue
        if t >= tx * ty:
            break
        x0, y0 = (t % tx) * abi.TILE_W, (t // tx) * abi.TILE_H
        tile = np.zeros((abi.TILE_H, abi.TILE_W), dtype=np.uint32)
        h, w = min(abi.TILE_H, height - y0), min(abi.TILE_W, width - x0)
        tile[:h, :w] = img[y0:y0 + h, x0:x0 + w]
        out[slot * 512:(slot + 1) * 512] = _tile_to_slots(tile)
    return out


def detile_host(gathered, width, height, world, rank_stride=0, offset=0, table=None, tiles_per_rank=None):
    """Host mirror of xrt_detile_device / xrt_detile_table_device (k_detile): rank-major gathered buffers -> H*W frame."""
    tx, ty, tpr = shard_layout(width, height, world)
    if table is not None:
        tpr = int(tiles_per_rank)
    flat = np.asarray(gathered, dtype=np.uint32).reshape(-1)
    stride = rank_stride or tpr * 512
    g = np.stack([flat[offset + r * stride: offset + r * stride + tpr * 512] for r in range(world)]).reshape(world, tpr, 512)
    img = np.zeros((height, width), dtype=np.uint32)
    for rank in range(world):
        for slot in range(tpr):
            t = slot * world + rank if table is None else int(table[rank * tpr + slot])
            if t < 0:
                continue
            if t >= tx * ty:
                break
            x0, y0 = (t % tx) * abi.TILE_W, (t // tx) * abi.TILE_H
            h, w = min(abi.TILE_H, height - y0), min(abi.TILE_W, width - x0)
            img[y0:y0 + h, x0:x0 + w] = _slots_to_tile(g[rank, slot])[:h, :w]
    return img.reshape(-1)


def gather_frame(local, width, height, group=None, dst=0):
    """torch.distributed.gather of the per-rank tile buffers (int32 tensors, CPU/gloo or GPU/RCCL).
    Returns the rank-major gathered tensor on `dst`, None elsewhere."""
    return gather_frame_async(local, group, dst)()


def gather_frame_async(local, group=None, dst=0, recv=None):
    """Start the gather and return a function that waits for it and yields the rank-major tensor on `dst`
    (None elsewhere).  `recv` (dst only): a preallocated (world * len(local)) tensor whose row views receive
    the shards directly, so no concatenation copy is needed."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bufs = None
    if rank == dst:
        if recv is None:
            recv = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        bufs = list(recv.view(world, -1).unbind(0))
    work = dist.gather(local, bufs, dst=dst, group=group, async_op=True)

    def wait():
        work.wait()
        return recv if rank == dst else None
    return wait


def detile_device(gathered, width, height, world, out, stream=None, rank_stride=0, offset=0, table_dev=None, tiles_per_rank=None):
    """xrt_detile_device on HBM-resident tensors.  rank_stride / offset (pixels): rank r's tiles start at
    offset + r * rank_stride of `gathered` (0 = contiguous) -- one gather may carry the tiles of several frames.
    table_dev (an int32 tensor on the device) / tiles_per_rank: the buffers were rendered under that tile table (xrt_detile_table_device)."""
    if table_dev is not None:
        abi.check(abi.lib().xrt_detile_table_device(width, height, world, int(tiles_per_rank), C.c_void_p(table_dev.data_ptr()),
                                                    C.c_void_p(gathered.data_ptr() + 4 * int(offset)), int(rank_stride), C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(stream or 0)))
        return out
    abi.check(abi.lib().xrt_detile_device(width, height, world, C.c_void_p(gathered.data_ptr() + 4 * int(offset)), int(rank_stride),
                                          C.c_void_p(out.data_ptr()), C.c_void_p(stream or 0)))
    return out
