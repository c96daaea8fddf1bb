"""Pyramid levels: SourceImage::resize = the `image` crate's Lanczos3
(reconstruction.rs:146-162).  The crate is not vendored with the reference, so nothing pins either side to it
("parity unpinned"); but the device (weights from glibc sinf on the host, f32 accumulation in tap order, no
contraction) and the oracle's numpy restatement (the same glibc sinf through ctypes, the same single IEEE f32
operations in the same order) must agree BIT FOR BIT: byte output, MAX_DIFF = 0."""
import numpy as np
import pytest

from cybervision_amd import synth

MAX_DIFF = 0            # grey levels: byte output is bit-exact against the oracle
MAX_FRACTION = 0.0      # of the pixels


@pytest.fixture(scope="module")
def lz():
    from oracle import cvref_resize

    return cvref_resize


def test_lanczos3_known_answers(lz):
    """Weights are normalised per output sample: a constant image stays constant; equal dimensions are a copy; the
    kernel is the Lanczos window (1 at 0, 0 at the other integers and beyond +-3); downsampling a smooth ramp by 2
    reproduces the ramp sampled at the new pixel centres; dims follow (w as f32 * scale) as u32."""
    c = np.full((90, 130), 201, dtype=np.uint8)
    assert (lz.resize_lanczos3(c, 41, 33) == 201).all()
    a, _, _ = synth.make_pair(96, 80, seed=5)
    assert (lz.resize_lanczos3(a, 96, 80) == a).all()
    k = lz.lanczos3_kernel(np.array([0.0, 1.0, 2.0, 3.0, 3.5, -1.0, 0.5], dtype=np.float32))
    assert k[0] == 1.0 and np.abs(k[1:3]).max() < 1e-6 and k[3] == 0.0 and k[4] == 0.0 and abs(k[5]) < 1e-6
    assert abs(k[6] - (np.sin(np.pi / 2) / (np.pi / 2)) * (np.sin(np.pi / 6) / (np.pi / 6))) < 1e-6
    ramp = np.tile(np.arange(40, 200, dtype=np.uint8)[None, :], (64, 1))
    half = lz.resize_lanczos3(ramp, 80, 32)
    want = 40 + 2 * np.arange(80) + 0.5
    assert np.abs(half[16, 4:-4] - want[4:-4]).max() <= 1.0
    assert lz.resize_scale(np.zeros((75, 101), dtype=np.uint8), 0.25).shape == (18, 25)
    # overshoot is clamped, not wrapped: a hard edge stays within [0, 255]
    edge = np.zeros((64, 64), dtype=np.uint8)
    edge[:, 32:] = 255
    out = lz.resize_lanczos3(edge, 32, 32)
    assert out.min() == 0 and out.max() == 255


@pytest.mark.gpu
@pytest.mark.parametrize("dims,scale", [((512, 384), 0.5), ((1000, 700), 0.25), ((333, 517), 0.125), ((256, 256), 1.0),
                                        ((2048, 2048), 1.0 / 32)])
def test_device_lanczos3_matches_numpy_restatement(gpu_device, lz, dims, scale):
    from cybervision_amd import correlation

    a, _, _ = synth.make_pair(dims[0], dims[1], seed=dims[0] % 97, sem_style=True)
    want = lz.resize_scale(a, scale)
    got = correlation.resize_lanczos3(gpu_device, a, scale)
    assert got.shape == want.shape and got.dtype == np.uint8
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= MAX_DIFF, d.max()
    assert (d > 0).mean() <= MAX_FRACTION, (d > 0).mean()


@pytest.mark.gpu
def test_device_lanczos_pyramid_resident(gpu_device, lz):
    """The level loop's pyramid, built on the device from a device-resident image and kept there."""
    import torch

    from cybervision_amd import correlation

    a, _, _ = synth.make_pair(640, 480, seed=3)
    pyr = correlation.lanczos_pyramid(gpu_device, torch.from_numpy(a).cuda(), 2)
    assert [tuple(p.shape) for p in pyr] == [(480, 640), (240, 320), (120, 160)] and all(p.is_cuda for p in pyr)
    assert (pyr[0].cpu().numpy() == a).all()
    for k in (1, 2):
        d = np.abs(pyr[k].cpu().numpy().astype(np.int32) - lz.resize_scale(a, 1.0 / (1 << k)).astype(np.int32))
        assert d.max() <= MAX_DIFF and (d > 0).mean() <= MAX_FRACTION
