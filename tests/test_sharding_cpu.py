"""Host logic of the multi-GPU path, on CPU: the row partition rule and the in-place band
all-gather (world_size 2 and 3 over gloo, ragged level heights)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

from cybervision_amd import sharding

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("lh", [1, 7, 64, 97, 511, 512, 4096])
@pytest.mark.parametrize("den", [1, 2, 3, 4, 8])
def test_shard_rows_partition(lh, den):
    rows = []
    for num in range(den):
        r0, r1 = sharding.shard_rows(lh, num, den)
        assert 0 <= r0 <= r1 <= lh and r1 - r0 <= sharding.rows_per_shard(lh, den)
        rows.extend(range(r0, r1))
    assert rows == list(range(lh)), "bands must tile the level exactly once, in order"
    assert den * sharding.rows_per_shard(lh, den) >= lh


def test_small_levels_are_not_sharded():
    assert not sharding.level_is_sharded(64, 64, 2)       # coarsest level: every rank computes it whole
    assert not sharding.level_is_sharded(256, 256, 8)
    assert sharding.level_is_sharded(512, 512, 8) and sharding.level_is_sharded(4096, 4096, 8)
    assert not sharding.level_is_sharded(4096, 4096, 1)


@pytest.mark.parametrize("world", [2, 3])
def test_band_allgather_over_gloo(world):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "_shard_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, out


def test_torch_generator_equals_numpy_generator():
    """bench.py builds its pairs with synth.make_pair_torch (on the GPU): same integer arithmetic, same bytes."""
    import numpy as np

    from cybervision_amd import synth

    for (w, h, tilt, seed) in [(300, 200, 0.0, 1234), (257, 311, 30.0, 7), (128, 128, 90.0, 1234)]:
        a, b, d = synth.make_pair(w, h, seed=seed, tilt_deg=tilt)
        ta, tb, td = synth.make_pair_torch(w, h, seed=seed, tilt_deg=tilt, device="cpu")
        assert (ta.numpy() == a).all() and (tb.numpy() == b).all() and (td.numpy() == d).all()
        for p, q in zip(synth.box_pyramid(a, 2), synth.box_pyramid_torch(ta, 2)):
            assert (q.numpy() == p).all()
