#!/usr/bin/env python3
"""RCCL smoke test on ONE GPU (world size 1): torch.distributed's nccl backend initialises, and the all-gather
hook used by bench.py (cybervision_amd.sharding.make_allgather) runs on a raw device pointer of a level grid,
ordered with the library's kernels on the same stream.  The multi-rank behaviour itself is covered by the
gloo tests; this only proves the RCCL code path is callable on this software stack."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist

from cybervision_amd import correlation, sharding, synth

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a, b, _ = synth.make_pair(512, 512)
steps = synth.optimal_scale_steps(512, 512)
p1, p2 = synth.box_pyramid(a, steps), synth.box_pyramid(b, steps)
d1 = [torch.from_numpy(p).cuda() for p in p1]
d2 = [torch.from_numpy(p).cuda() for p in p2]
dev = correlation.create_gpu_context(ordinal=0, stream=torch.cuda.current_stream().cuda_stream)
pc = correlation.PointCorrelations(dev, (512, 512), (512, 512), synth.F_HORIZONTAL)
assert pc.set_row_band(0, 1)
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
g = pc.level_grid(correlation.CorrelationDirection.Forward)
before = sharding.alias_bytes(g["cells"], g["rows_per_shard"] * g["lw"] * 4, device=True).clone()
gather = sharding.make_allgather(0, 1)
gather(g["cells"], g["rows_per_shard"] * g["lw"] * 4, 1, 0)
dist.barrier()
torch.cuda.synchronize()
after = sharding.alias_bytes(g["cells"], g["rows_per_shard"] * g["lw"] * 4, device=True)
assert torch.equal(before, after)
xy, corr = pc.complete()
print("rccl smoke ok:", int((xy[..., 0] >= 0).sum()), "matches")
pc.close()
dist.destroy_process_group()

# the library's own RCCL path (cvhip_rccl_*), world size 1, on a device handle that owns a PRIVATE stream: the
# communicator, the in-place all-gather / gather-to-root on a level grid, and the sharded-context wiring
dev2 = correlation.create_gpu_context(ordinal=0)
comm = sharding.RcclCommunicator(dev2, sharding.RcclCommunicator.unique_id(), 0, 1)
pc = correlation.PointCorrelations(dev2, (512, 512), (512, 512), synth.f_tilt(30.0))
pc.set_row_shard_rccl(comm)
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
g = pc.level_grid(correlation.CorrelationDirection.Forward)
nbytes = g["rows_per_shard"] * g["lw"] * 4
dev2.synchronize()
before = sharding.alias_bytes(g["cells"], nbytes, device=True).clone()
comm.allgather(g["cells"], nbytes)
comm.gather(g["cells"], nbytes, 0)
pc.gather_bands_rccl(comm, 0)
dev2.synchronize()
assert torch.equal(before, sharding.alias_bytes(g["cells"], nbytes, device=True))
xy2, _ = pc.complete()
print("library rccl smoke ok:", int((xy2[..., 0] >= 0).sum()), "matches")
pc.close()
comm.close()
dev2.close()
