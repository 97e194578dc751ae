"""GPU: DuckietownVecEnv -- the batched gym contract on device tensors (step -> reward/done -> device-side reset of
the finished episodes -> render -> PIL-exact observation), no host synchronisation inside the loop."""
import numpy as np
import pytest

from dtsim import DuckietownVecEnv, _ffi, resample

pytestmark = pytest.mark.gpu


def test_vecenv_contract_and_auto_restart():
    import torch
    N = 256
    env = DuckietownVecEnv("small_loop_only_duckies", N, obs_shape=(60, 80), seed=3, max_steps=30, domain_rand=False,
                           camera_width=160, camera_height=120)
    obs = env.reset()
    assert obs.shape == (N, 3, 60, 80) and obs.dtype == torch.float32 and obs.is_cuda
    assert 0.0 <= float(obs.min()) and float(obs.max()) <= 1.0
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    n_done = torch.zeros(N, device="cuda")
    for t in range(70):
        a = torch.rand((N, 2), device="cuda", generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        assert obs.shape == (N, 3, 60, 80) and rew.shape == (N,) and done.dtype == torch.bool
        n_done += done.float()
        # a finished episode is restarted before the observation is made: its step counter is back to 0
        torch.cuda.synchronize()
        steps_now = torch.as_tensor(env.sim.field_device(_ffi.FIELD_STEP_COUNT), device="cuda")
        assert bool((steps_now[done] == 0).all())
        assert bool((info["episode_steps"][done] > 0).all())
    assert float(n_done.min()) >= 2                      # max_steps = 30: every env finished at least twice in 70 steps
    # the observation is the PIL-exact resize of the frames
    torch.cuda.synchronize()
    frames = torch.as_tensor(env.sim.frames_device(), device="cuda").cpu().numpy()
    want = resample.observation(frames[:4], 60, 80, chw=True, normalize=True)
    assert np.array_equal(obs[:4].cpu().numpy(), want)
    env.close()


def test_vecenv_raw_frames_and_numpy_actions():
    import torch
    env = DuckietownVecEnv("small_loop", 8, obs_shape=None, seed=1, camera_width=160, camera_height=120)
    obs = env.reset()
    assert obs.shape == (8, 120, 160, 3) and obs.dtype == torch.uint8
    o2, r, d, info = env.step(np.full((8, 2), 0.3, np.float32))
    torch.cuda.synchronize()
    assert o2.shape == obs.shape and float(r.abs().sum()) > 0
    env.close()
