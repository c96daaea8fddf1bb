"""Row sharding of the dense-correlation level grids across ranks (one process per GPU).

Every (level, direction) search pass is embarrassingly parallel over rows of the searched
image (reference: each closure of the rayon loop writes only its own cell,
correlation/mod.rs:288-304), so rank r searches rows [r*rps, (r+1)*rps) and the bands are
all-gathered in place; cross-checks then run redundantly on the full grid.  There is no
reduction across ranks, so the N-rank result is bit-identical to the 1-rank result.

The collective is torch.distributed's all-gather: backend "nccl" (= RCCL over xGMI) for device
memory, "gloo" for the CPU tests.  The library hands us a raw pointer; we alias it as a tensor.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def rows_per_shard(lh: int, den: int) -> int:
    return (lh + den - 1) // den


def shard_rows(lh: int, num: int, den: int):
    """Same rule as cvhip_ctx_set_row_shard (include/cvhip.h)."""
    rps = rows_per_shard(lh, den)
    r0 = min(lh, num * rps)
    return r0, min(lh, r0 + rps)


def level_is_sharded(h1: int, h2: int, den: int) -> bool:
    """cvhip_correlate_level shards a level only if every rank gets >= 64 rows."""
    return den > 1 and min(h1, h2) // den >= 64


class _DeviceBytes:
    """Expose [ptr, ptr + nbytes) of device memory through __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def alias_bytes(ptr: int, nbytes: int, device: bool):
    """A uint8 torch tensor aliasing raw memory (device or host)."""
    import torch

    if device:
        return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


CELL_BYTES = 4  # bytes per level pixel of EACH plane of a level grid (u32 match word / f32 score)


def copy_rows(dst_grid, src_grid, r0: int, r1: int, scores: bool = True):
    """Rows [r0, r1) of src_grid's planes -> the same rows of dst_grid (both from PointCorrelations.level_grid, on the
    same GPU): what a gather of one band does.  scores=False: the match plane only."""
    lw = src_grid["lw"]
    nbytes = (r1 - r0) * lw * CELL_BYTES
    if nbytes <= 0:
        return
    for key in ("cells", "scores") if scores else ("cells",):
        alias_bytes(dst_grid[key] + r0 * lw * CELL_BYTES, nbytes, True).copy_(alias_bytes(src_grid[key] + r0 * lw * CELL_BYTES, nbytes, True))


def make_allgather(rank: int, world: int, group=None, device: bool = True):
    """gather(cells_ptr, shard_bytes, n_shards, direction) for PointCorrelations.set_row_shard."""
    import torch.distributed as dist

    def gather(cells_ptr: int, shard_bytes: int, n_shards: int, direction: int):
        assert n_shards == world
        full = alias_bytes(cells_ptr, shard_bytes * n_shards, device)
        mine = full[rank * shard_bytes:(rank + 1) * shard_bytes]
        if dist.get_backend(group) == "nccl":
            # the send chunk is a private copy (16.8 MB at 4096^2 / 8 ranks: microseconds) so that the
            # collective never sees overlapping send/receive buffers
            dist.all_gather_into_tensor(full, mine.clone(), group=group)
        elif device:
            # gloo cannot all-gather device memory: stage through the host (test / fallback path only)
            import torch

            torch.cuda.current_stream().synchronize()
            outs = [torch.empty(shard_bytes, dtype=torch.uint8) for _ in range(n_shards)]
            dist.all_gather(outs, mine.cpu(), group=group)
            full.copy_(torch.cat(outs))
        else:
            outs = [full[r * shard_bytes:(r + 1) * shard_bytes] for r in range(n_shards)]
            dist.all_gather(outs, mine.clone(), group=group)

    return gather


class RcclCommunicator:
    """One RCCL communicator per device handle, created inside libcvhip (cvhip_rccl_*): no torch in the data path.
    `unique_id()` is called on rank 0 and the 128 bytes distributed by the launcher (e.g. torch.distributed's
    broadcast_object_list over the rendezvous store)."""

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib

        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().cvhip_rccl_unique_id(buf), "cvhip_rccl_unique_id")
        return bytes(buf)

    def __init__(self, device, unique_id: bytes, rank: int, world: int):
        from . import _lib

        assert len(unique_id) == 128
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().cvhip_rccl_create(device.handle, buf, rank, world, C.byref(self._h)), "cvhip_rccl_create")

    @property
    def handle(self):
        return self._h

    def allgather(self, ptr: int, shard_bytes: int):
        from . import _lib

        _lib.check(_lib.lib().cvhip_rccl_allgather(self._h, C.c_void_p(ptr), shard_bytes), "cvhip_rccl_allgather")

    def gather(self, ptr: int, shard_bytes: int, root: int = 0):
        from . import _lib

        _lib.check(_lib.lib().cvhip_rccl_gather(self._h, C.c_void_p(ptr), shard_bytes, root), "cvhip_rccl_gather")

    def close(self):
        from . import _lib

        if getattr(self, "_h", None):
            _lib.lib().cvhip_rccl_destroy(self._h)
            self._h = None
