"""Worker for test_corr_gpu.py::test_library_rccl_path_world2: one process per GPU (rank r on cuda:r), the library's own RCCL
path (cvhip_rccl_*: a communicator on the device handle, collectives on the handle's stream) at world size > 1.  gloo only
carries the 128-byte communicator id.  Two golden cases per run: a perspective pair through cvhip_ctx_set_row_shard_rccl (an
all-gather after every sharded search pass: EVERY rank must hold the golden grid) and a row-local tilted pair through
independent bands + cvhip_ctx_gather_bands_rccl (the single gather to rank 0: rank 0 must hold the golden grid)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def check(rank, what, got, g):
    xy, corr = got
    want_xy, want_corr = g["fwd_xy"].astype(np.int32), g["fwd_corr"]
    bad = np.nonzero((xy != want_xy).any(axis=-1))
    assert bad[0].size == 0, f"rank {rank} {what}: {bad[0].size} cells differ from the golden grid, rows {np.unique(bad[0])[:20]}"
    valid = want_xy[..., 0] >= 0
    assert (corr.view(np.uint32)[valid] == want_corr.view(np.uint32)[valid]).all(), f"rank {rank} {what}: scores differ"


def main():
    import torch
    import torch.distributed as dist

    from cybervision_amd import correlation, sharding, synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    dev = correlation.create_gpu_context(ordinal=rank)  # a private stream: the collectives are ordered on it by the library
    uid = [sharding.RcclCommunicator.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = sharding.RcclCommunicator(dev, uid[0], rank, world)
    try:
        for case, mode in (("persp_240x180", "gather"), ("tilt3_200x150", "band")):
            g = np.load(ROOT / "tests" / "golden" / f"corr_{case}.npz")
            img1, img2, steps = g["img1"], g["img2"], int(g["steps"])
            p1, p2 = synth.box_pyramid(img1, steps), synth.box_pyramid(img2, steps)
            (h1, w1), (h2, w2) = img1.shape, img2.shape
            pc = correlation.PointCorrelations(dev, (w1, h1), (w2, h2), g["F"], correlation.ProjectionMode(int(g["projection"])))
            try:
                band = pc.set_row_band(rank, world)
                assert band == (mode == "band"), f"{case}: expected mode {mode}"
                if not band:
                    pc.set_row_shard_rccl(comm)
                for i in range(steps + 1):
                    k = steps - i
                    pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
                if band:
                    pc.gather_bands_rccl(comm, 0)
                dev.synchronize()
                if not band or rank == 0:
                    check(rank, f"{case} ({mode})", pc.complete(), g)
            finally:
                pc.close()
            dist.barrier()
    finally:
        dev.close()   # deferred by the library until the communicator is gone
        comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
