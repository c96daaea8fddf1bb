"""Seeded slices of the randomised parity sweeps (tests/tools/fuzz_box.py, fuzz_ransac.py) under the suite: the sweeps are what
found the round-4 cross-stream race (fresh contexts, host images through the upload ring, the reference's four calls per level,
result bands 0-4, packed cells), so a small slice of them runs with every `pytest -m gpu` - the full sweeps stay tools."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).parent / "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_box_slice(gpu_device, oracle):
    """24 random affine pairs (sizes 90-360 px, 13 tilts around both axes, padded second images, SEM noise) and 6 perspective
    pairs, each through search versions 3, 4 and 5 in a fresh context, alternately through the level call and the reference's
    four calls on host images, with 0-4 result bands and packed / plain cells: every grid == the oracle's, bit for bit."""
    import fuzz_box

    msgs = []
    bad = fuzz_box.run(24, seed=2026, maxdim=360, dev=gpu_device, log=msgs.append)
    bad += fuzz_box.run(6, seed=2027, maxdim=300, perspective=True, dev=gpu_device, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if "MISMATCH" in m)


def test_fuzz_ransac_slice(gpu_device):
    """10 random match sets (600-30 000 matches, 10-60 % outliers, three image sizes, both 7-point pencils): the polled, the
    in-order and the listener-attached schedule of cvhip_find_ransac return the same matrix and the same inlier mask."""
    import fuzz_ransac

    msgs = []
    bad = fuzz_ransac.run(10, seed=77, dev=gpu_device, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if "MISMATCH" in m)
