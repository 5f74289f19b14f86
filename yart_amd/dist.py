"""Multi-GPU tile sharding: one process per GPU, ``torch.distributed`` (backend ``nccl`` =
RCCL over xGMI on ROCm, ``gloo`` on CPU for the tests).

The reference's unit of parallel work is the 64x64 tile (reference
``src/cpu/tile-renderer.hpp:126-144, 161-184``: worker threads pop tiles from a shared
queue and merge finished tiles into the frame under a mutex, ``:225-241``).  Every
(pixel, sample) is independent and the sampler is a pure function of them, so the path
shards with no sample-level exchange: each rank renders ALL samples of its own tiles and
finishes GMoN locally; the only exchange step is the merge of the per-rank framebuffers,
one ``reduce(SUM)`` of W*H*4 fp32 to rank 0 per render (33 MB at 1080p; non-owned pixels
are exactly 0 so the sum is the merged frame, bit for bit).

Tiles are dealt round-robin in Morton order (neighbouring tiles — similar cost — go to
different ranks, which balances sky against geometry).
"""
from __future__ import annotations

import numpy as np


def _part1by1(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint64) & 0xFFFFFFFF
    v = (v ^ (v << 16)) & 0x0000FFFF0000FFFF
    v = (v ^ (v << 8)) & 0x00FF00FF00FF00FF
    v = (v ^ (v << 4)) & 0x0F0F0F0F0F0F0F0F
    v = (v ^ (v << 2)) & 0x3333333333333333
    v = (v ^ (v << 1)) & 0x5555555555555555
    return v


def tile_owners(width: int, height: int, tile: int, world_size: int) -> np.ndarray:
    """(tiles_y, tiles_x) array of owning ranks — the host mirror of buildPixelList() in
    yart_amd/csrc/yart_hip.hip (tiles sorted by Morton code, dealt round-robin)."""
    tx, ty = -(-width // tile), -(-height // tile)
    ys, xs = np.mgrid[0:ty, 0:tx]
    morton = (_part1by1(ys.ravel()) << np.uint64(1)) | _part1by1(xs.ravel())
    linear = (ys.ravel() * tx + xs.ravel()).astype(np.uint64)
    order = np.lexsort((linear, morton))           # sort by (morton, linear) like std::sort on pairs
    owners = np.empty(tx * ty, np.int64)
    owners[order] = np.arange(tx * ty) % world_size
    return owners.reshape(ty, tx)


def pixel_mask(width: int, height: int, tile: int, rank: int, world_size: int) -> np.ndarray:
    """(H, W) bool mask of the pixels `rank` renders."""
    own = tile_owners(width, height, tile, world_size) == rank
    return np.kron(own, np.ones((tile, tile), bool))[:height, :width]


def merge(framebuffer, dst: int = 0, group=None):
    """Merge per-rank framebuffers (disjoint tile sets, zeros elsewhere) onto `dst`.
    `framebuffer` is a torch tensor (device memory with nccl/RCCL, host memory with gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(framebuffer, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return framebuffer


def render_sharded(device_scene, params: dict, framebuffer, rank: int, world_size: int, stream=None, group=None):
    """Render this rank's tiles into `framebuffer` (torch CUDA tensor, H x W x 4 fp32) and
    merge on rank 0. Returns the library's stats dict for this rank."""
    st = device_scene.render_into(framebuffer, params, rank=rank, world_size=world_size, stream=stream)
    merge(framebuffer, 0, group)
    return st
