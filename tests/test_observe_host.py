"""Observation resampling, host side: the Pillow restatement that builds dtsim_observe's
tables is pinned bit-exact against PIL.Image.resize(BILINEAR) -- what the reference's learners call
through scipy imresize (learning/utils/wrappers.py:38-54)."""
import numpy as np
import pytest

from dtsim import resample

PIL = pytest.importorskip("PIL.Image")

CASES = [((480, 640), (120, 160)), ((480, 640), (80, 80)), ((480, 640), (84, 84)), ((120, 160), (150, 200)),
         ((84, 84), (42, 64)), ((480, 640), (480, 160)), ((480, 640), (60, 640)), ((480, 640), (224, 224))]


def _image(rng, H, W):
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    a[: H // 2, : W // 2] = rng.integers(0, 2, (H // 2, W // 2, 3), dtype=np.uint8) * 255   # hard edges: exercises clipping
    return a


@pytest.mark.parametrize("src,dst", CASES)
def test_resize_matches_pil_bit_exact(src, dst):
    rng = np.random.default_rng(hash((src, dst)) & 0xFFFF)
    a = _image(rng, *src)
    ref = np.asarray(PIL.fromarray(a).resize((dst[1], dst[0]), PIL.BILINEAR))
    assert np.array_equal(resample.resize_bilinear(a, dst[0], dst[1]), ref)


def test_tables_are_normalised_fixed_point():
    b, k = resample.coeffs(640, 160)
    assert k.shape == (160, 9) and b.shape == (160, 2)
    s = k.sum(axis=1)
    assert np.all(np.abs(s - (1 << resample.PRECISION_BITS)) <= 4)        # taps sum to 1.0 up to rounding
    assert np.all(b[:, 0] >= 0) and np.all(b[:, 0] + b[:, 1] <= 640) and np.all(np.diff(b[:, 0]) >= 0)


def test_observation_layouts():
    rng = np.random.default_rng(1)
    fr = np.stack([_image(rng, 120, 160) for _ in range(3)])
    o = resample.observation(fr, 60, 80)
    assert o.shape == (3, 60, 80, 3) and o.dtype == np.uint8
    c = resample.observation(fr, 60, 80, chw=True, normalize=True)
    assert c.shape == (3, 3, 60, 80) and c.dtype == np.float32
    assert np.array_equal(c, o.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255))


def test_resize_matches_pil_on_random_sizes():
    """Fuzz: random source / target sizes (up- and down-scaling, extreme aspect changes, 1-pixel axes) against Pillow."""
    from PIL import Image
    rng = np.random.default_rng(123)
    for _ in range(60):
        H, W = int(rng.integers(1, 97)), int(rng.integers(1, 131))
        oh, ow = int(rng.integers(1, 80)), int(rng.integers(1, 90))
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        got = resample.resize_bilinear(img, oh, ow)
        assert np.array_equal(got, want), ((H, W), (oh, ow))


@pytest.mark.parametrize("n_in,n_out,S", [(640, 160, 4), (640, 80, 8), (480, 240, 2), (480, 120, 4), (480, 60, 8), (320, 80, 4)])
def test_power_of_two_scales_have_uniform_small_integer_taps(n_in, n_out, S):
    """What dtsim_observe's fast kernels (k_observe_pow2, the dot4 / two-lane paths of k_observe) rely on, and detect from the tables at
    run time: for a power-of-two scale S every interior output coordinate has the SAME 2S taps, starting at S*o - S/2, and they are the
    triangle weights (1, 3, .., 2S-1, 2S-1, .., 3, 1) times 2^(22 - log2(2 S^2)) exactly; only the first and last coordinate are clipped."""
    bounds, kk = resample.coeffs(n_in, n_out)
    tri = [2 * t + 1 for t in range(S)] + [2 * t + 1 for t in range(S - 1, -1, -1)]
    shift = 22 - int(np.log2(2 * S * S))
    assert sum(tri) == 2 * S * S and kk.shape[1] >= 2 * S
    for o in range(1, n_out - 1):
        assert bounds[o, 0] == S * o - S // 2 and bounds[o, 1] == 2 * S
        assert kk[o, :2 * S].tolist() == [w << shift for w in tri]
    assert bounds[0, 0] == 0 and bounds[0, 1] < 2 * S and bounds[-1, 0] + bounds[-1, 1] == n_in
