"""GPU: dtsim_observe (device-side ResizeWrapper / ImgWrapper / NormalizeWrapper) is bit-identical to
PIL.Image.resize(BILINEAR) applied to the rendered frames (and to the host restatement)."""
import numpy as np
import pytest

from dtsim import BatchedSimulator, resample

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("W,H,ow,oh", [(640, 480, 160, 120), (640, 480, 320, 240), (160, 120, 40, 30), (640, 480, 320, 120), (640, 480, 80, 80), (640, 480, 84, 84), (160, 120, 200, 150),
                                       (84, 84, 64, 42), (640, 480, 160, 480), (640, 480, 640, 60), (640, 480, 640, 480),
                                       # power-of-two scales on both axes (k_observe_pow2 + k_observe_border): every (x, y) scale pair
                                       (640, 480, 160, 240), (640, 480, 160, 60), (640, 480, 80, 240), (640, 480, 80, 120), (640, 480, 80, 60),
                                       (320, 240, 80, 60), (64, 48, 16, 12), (32, 16, 4, 4)])
def test_observe_matches_pil(W, H, ow, oh):
    import torch
    N = 5
    sim = BatchedSimulator("small_loop_only_duckies", N, camera_width=W, camera_height=H, distortion=False,
                           domain_rand=True, seed=11)
    sim.render()
    frames = sim.frames_host()
    o = sim.observe(oh, ow)
    sim.sync()                                     # observe is asynchronous on the handle's stream
    obs = torch.as_tensor(o, device="cuda:0").cpu().numpy()
    assert obs.shape == (N, oh, ow, 3) and obs.dtype == np.uint8
    for e in range(N):
        ref = np.asarray(PIL.fromarray(frames[e]).resize((ow, oh), PIL.BILINEAR))
        assert np.array_equal(obs[e], ref), (e, np.abs(obs[e].astype(int) - ref.astype(int)).max())
    o = sim.observe(oh, ow, chw=True, normalize=True)
    sim.sync()
    chw = torch.as_tensor(o, device="cuda:0").cpu().numpy()
    assert chw.shape == (N, 3, oh, ow) and chw.dtype == np.float32
    assert np.array_equal(chw, resample.observation(frames, oh, ow, chw=True, normalize=True))
    sim.close()


def test_observe_into_caller_buffer_and_errors():
    import torch
    sim = BatchedSimulator("small_loop", 3, camera_width=160, camera_height=120, seed=2)
    sim.render()
    buf = torch.zeros((3, 3, 60, 80), dtype=torch.float32, device="cuda:0")
    out = sim.observe(60, 80, chw=True, normalize=True, out=buf)
    sim.sync()
    assert out is buf and float(buf.max()) <= 1.0 and float(buf.mean()) > 0.0
    nor = BatchedSimulator("small_loop", 2, render=False, seed=2)
    with pytest.raises(Exception):
        nor.observe(60, 80)
    nor.close(); sim.close()


@pytest.mark.parametrize("W,H,ow,oh", [(640, 480, 80, 80), (640, 480, 160, 120), (640, 480, 84, 84), (160, 120, 200, 150), (84, 84, 64, 42),
                                       (640, 480, 640, 480), (126, 94, 37, 53)])
def test_observe_cubic_matches_the_opencv_statement(W, H, ow, oh):
    """The reference's own ResizeWrapper (src/gym_duckietown/wrappers.py:129-138, cv2 INTER_CUBIC) on the device: bit-identical to
    dtsim/resample.py resize_cubic (OpenCV's published 8-bit fixed-point path) for down- and up-scaling, odd sizes (byte-wise row
    path: 126 * 3 is not a multiple of 4) and the identity."""
    import torch
    N = 4
    sim = BatchedSimulator("small_loop_only_duckies", N, camera_width=W, camera_height=H, distortion=False, domain_rand=True, seed=5)
    sim.render()
    frames = sim.frames_host()
    o = sim.observe(oh, ow, interpolation="cv_cubic")
    sim.sync()
    obs = torch.as_tensor(o, device="cuda:0").cpu().numpy()
    assert obs.shape == (N, oh, ow, 3) and obs.dtype == np.uint8
    for e in range(N):
        ref = resample.resize_cubic(frames[e], oh, ow)
        assert np.array_equal(obs[e], ref), (e, np.abs(obs[e].astype(int) - ref.astype(int)).max())
    o = sim.observe(oh, ow, chw=True, normalize=True, interpolation="cv_cubic")
    sim.sync()
    chw = torch.as_tensor(o, device="cuda:0").cpu().numpy()
    ref = np.stack([resample.resize_cubic(frames[e], oh, ow) for e in range(N)]).transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)
    assert chw.shape == (N, 3, oh, ow) and chw.dtype == np.float32 and np.array_equal(chw, ref)
    with pytest.raises(ValueError):
        sim.observe(oh, ow, interpolation="lanczos")
    sim.close()
