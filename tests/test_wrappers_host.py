"""Adapters above the env boundary (reference wrappers.py / learning/utils/wrappers.py), host side."""
import numpy as np
import pytest

from dtsim import resample
from gym_duckietown import wrappers as W


class _Env:
    """Minimal env with the attributes the wrappers touch."""
    distortion = True
    undistort = False
    wheel_dist = 0.102
    reward_range = (-1000, 1000)

    def __init__(self, h=120, w=160):
        self.observation_space = W._box(0, 255, (h, w, 3), np.uint8)
        self.action_space = W._box(-1, 1, (2,), np.float32)
        self.rng = np.random.default_rng(0)
        self.last_action = None
        self.h, self.w = h, w

    @property
    def unwrapped(self):
        return self

    def _obs(self):
        return self.rng.integers(0, 256, (self.h, self.w, 3), dtype=np.uint8)

    def reset(self):
        return self._obs()

    def step(self, action):
        self.last_action = np.asarray(action, dtype=float)
        return self._obs(), 1.5, False, {}


def test_action_wrappers():
    e = _Env()
    d = W.DiscreteWrapper(e)
    d.step(0); assert np.allclose(e.last_action, [0.6, 1.0])
    d.step(2); assert np.allclose(e.last_action, [0.7, 0.0])
    with pytest.raises(AssertionError):
        d.step(5)
    s = W.SteeringToWheelVelWrapper(e)
    s.step([0.5, 1.0])
    # envs/duckietown_env.py:36-61 arithmetic: omega = (v +- 0.5 a b) / r, u = omega (gain +- trim) / k, clamp
    om_r, om_l = (0.5 + 0.5 * 0.102) / 0.0318, (0.5 - 0.5 * 0.102) / 0.0318
    assert np.allclose(e.last_action, [min(om_l / 27.0, 1.0), min(om_r / 27.0, 1.0)])
    a = W.ActionWrapper(e)
    a.step([1.0, -0.5]); assert np.allclose(e.last_action, [0.8, -0.5])
    r = W.DtRewardWrapper(e)
    assert r.step([0, 0])[1] == 11.5 and r.reward(-1000) == -10 and r.reward(-2.0) == 2.0


def test_observation_wrappers_shapes_and_values():
    PIL = pytest.importorskip("PIL.Image")
    e = _Env(120, 160)
    o = W.PILResizeWrapper(e, shape=(60, 80, 3))
    x = o.reset()
    assert x.shape == (60, 80, 3) and x.dtype == np.uint8
    raw = _Env(120, 160).reset()                                   # same RNG stream as e's first frame
    assert np.array_equal(x, np.asarray(PIL.fromarray(raw).resize((80, 60), PIL.BILINEAR)))
    n = W.NormalizeWrapper(W.ImgWrapper(o))
    y, rew, done, info = n.step([0, 0])
    assert y.shape == (3, 60, 80) and y.dtype == np.float32 and 0.0 <= y.min() and y.max() <= 1.0
    t = W.PyTorchObsWrapper(_Env(120, 160))
    assert t.reset().shape == (3, 160, 120)


def test_cubic_resize_properties():
    # restated OpenCV INTER_CUBIC (parity unpinned): exact on constants, identity at scale 1, interpolates ramps
    img = np.full((40, 60, 3), 77, np.uint8)
    assert np.all(W.resize_cubic(img, 20, 30) == 77)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
    assert np.array_equal(W.resize_cubic(img, 33, 47), img)
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (10, 1, 3))
    up = W.resize_cubic(ramp, 10, 200).astype(int)
    assert np.all(np.abs(np.diff(up[5, 4:-4, 0]) - 1) <= 1)        # monotone ramp away from the replicated borders
    rw = W.ResizeWrapper(_Env(120, 160), resize_w=80, resize_h=80)
    assert rw.reset().shape == (120, 80, 80)                       # the reference resizes the axis-swapped image


def test_cubic_tables_known_answers():
    """OpenCV's interpolateCubic (A = -0.75) at fx = 0.5 and 0.25, 11-bit fixed point: (-3/32, 19/32, 19/32, -3/32) and
    (-27/256, 225/256, 67/256, -9/256) -- the halving and doubling cases of cv::resize."""
    from dtsim import resample
    first, taps = resample.cubic_coeffs(640, 320)
    assert first.tolist()[:3] == [-1, 1, 3] and all(t == [-192, 1216, 1216, -192] for t in taps.tolist())
    first, taps = resample.cubic_coeffs(10, 20)
    assert first[1] == -1 and taps[1].tolist() == [-216, 1800, 536, -72]
    assert first[2] == -1 and taps[2].tolist() == [-72, 536, 1800, -216] and first[0] == -2
    for a, b in ((640, 80), (480, 84), (120, 150), (84, 42)):
        f, t = resample.cubic_coeffs(a, b)
        assert np.all(np.abs(t.sum(1) - 2048) <= 1) and f.min() >= -2 and f.max() + 3 <= a + 1


def test_undistort_wrapper_sets_flag_and_remaps():
    e = _Env(480, 640)
    u = W.UndistortWrapper(e)
    assert e.undistort is True
    x = u.reset()
    assert x.shape == (480, 640, 3)
    mx, my = u.mapx, u.mapy
    assert abs(mx[240, 320] - 320) < 40 and abs(my[240, 320] - 240) < 40   # centre maps near the centre
