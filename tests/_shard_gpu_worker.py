"""Worker for test_corr_gpu.py::test_two_rank_sharded_level_calls: one of two processes sharing GPU 0.
Each rank row-shards the search passes, the library calls the all-gather hook from inside
cvhip_correlate_level, and the final grid must equal the committed golden fixture on every rank."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    import torch.distributed as dist

    from cybervision_amd import correlation, sharding, synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    g = np.load(ROOT / "tests" / "golden" / f"corr_{sys.argv[1]}.npz")
    img1, img2 = g["img1"], g["img2"]
    steps = int(g["steps"])
    p1, p2 = synth.box_pyramid(img1, steps), synth.box_pyramid(img2, steps)
    d1 = [torch.from_numpy(p).cuda() for p in p1]
    d2 = [torch.from_numpy(p).cuda() for p in p2]
    dev = correlation.create_gpu_context(ordinal=0, stream=torch.cuda.current_stream().cuda_stream)
    h1, w1 = img1.shape
    h2, w2 = img2.shape
    pc = correlation.PointCorrelations(dev, (w1, h1), (w2, h2), g["F"], correlation.ProjectionMode(int(g["projection"])))
    calls = []
    inner = sharding.make_allgather(rank, world, device=True)

    def gather(ptr, nbytes, n, direction):
        calls.append((nbytes, n, direction))
        inner(ptr, nbytes, n, direction)

    band = pc.set_row_band(rank, world)  # independent bands when the geometry is row-local ...
    assert band == (sys.argv[2] == "band"), f"expected mode {sys.argv[2]}"
    if not band:
        pc.set_row_shard(rank, world, gather)  # ... else bands + all-gather after every sharded pass
    sharded_levels = 0
    for i in range(steps + 1):
        k = steps - i
        before = len(calls)
        pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
        expect = (not band) and sharding.level_is_sharded(d1[k].shape[0], d2[k].shape[0], world)
        # a sharded level: the two match planes, and at scale 1 the forward score plane too
        assert (len(calls) - before) == ((3 if k == 0 else 2) if expect else 0), (k, calls[before:])
        sharded_levels += int(expect)
    if band:  # the single final gather of the forward bands
        lg = pc.level_grid(correlation.CorrelationDirection.Forward)
        inner(lg["cells"], lg["rows_per_shard"] * lg["lw"] * 4, world, 0)   # match plane
        inner(lg["scores"], lg["rows_per_shard"] * lg["lw"] * 4, world, 2)  # score plane
    else:
        assert sharded_levels >= 1, "test case too small to exercise the collective"
    xy, corr = pc.complete()
    want_xy, want_corr = g["fwd_xy"].astype(np.int32), g["fwd_corr"]
    bad = np.nonzero((xy != want_xy).any(axis=-1))
    assert bad[0].size == 0, (f"rank {rank}: sharded result differs from the golden grid in {bad[0].size} cells, "
                              f"rows {np.unique(bad[0])[:20]} cols {np.unique(bad[1])[:20]}")
    valid = want_xy[..., 0] >= 0
    assert (corr.view(np.uint32)[valid] == want_corr.view(np.uint32)[valid]).all()
    pc.close()
    dev.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
