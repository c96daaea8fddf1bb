"""Worker for tests/test_sharding_cpu.py: one gloo rank.  Emulates what every rank's device
does around the all-gather hook: it owns a host buffer laid out like the library's level grid
(8-byte cells, den * rows_per_shard rows), fills ONLY its own row band with the true level
result, calls the same gather function the GPU path uses, and checks it ends with the full grid."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch.distributed as dist

    from cybervision_amd import sharding

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = sharding.make_allgather(rank, world, device=False)
    rng = np.random.default_rng(1234)  # same stream on every rank -> same "true" grids
    for (lw, lh) in [(64, 64), (130, 97), (257, 511), (33, 7)]:
        truth = rng.integers(0, 2 ** 32, size=(lh, lw, 2), dtype=np.uint64).astype(np.uint32)
        rps = sharding.rows_per_shard(lh, world)
        buf = np.full((world * rps, lw, 2), 0xDEADBEEF, dtype=np.uint32)  # padded to equal chunks
        r0, r1 = sharding.shard_rows(lh, rank, world)
        buf[r0:r1] = truth[r0:r1]
        gather(buf.ctypes.data, rps * lw * 8, world, 0)
        assert (buf[:lh] == truth).all(), f"rank {rank}: gathered grid differs for {lw}x{lh}"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
