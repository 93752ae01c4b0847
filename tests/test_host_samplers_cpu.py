"""Host-side sampler loops (cmtts_amd.host.sample_*, get_sigmas_karras, stochastic_iterative_sampler — the reference's
karras_diffusion.py:580-854 restated on torch tensors) against the numpy oracle, on the CPU with a synthetic denoiser:
no GPU and no HIP library needed (the loops never touch the extension)."""
import numpy as np
import pytest
import torch

from cmtts_amd import host
from oracle import cmtts_oracle as O

SIGMA_MIN, SIGMA_MAX, RHO = 0.002, 80.0, 7.0


def _denoiser_np(x, sigma):
    """A smooth stand-in for KarrasDenoiser.denoise: shrinks x by a sigma-dependent factor."""
    s = np.asarray(sigma, np.float32).reshape(-1, 1, 1, 1)
    return (x * np.float32(0.25) / (np.float32(0.25) + s * s) + np.float32(0.1) * np.tanh(x)).astype(np.float32)


def _denoiser_t(x, sigma):
    s = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
    return x * 0.25 / (0.25 + s * s) + 0.1 * torch.tanh(x)


class _Gen:
    def __init__(self, noise):
        self.noise, self.i = noise, 0

    def randn_like(self, x):
        t = torch.from_numpy(self.noise[self.i])
        self.i += 1
        return t


def _noise(n, shape, seed=0):
    rs = np.random.RandomState(seed)
    return [rs.standard_normal(size=shape).astype(np.float32) for _ in range(n)]


def test_sigmas_match_oracle():
    for n in (2, 5, 18):
        s = host.get_sigmas_karras(n, SIGMA_MIN, SIGMA_MAX, RHO).numpy()
        np.testing.assert_allclose(s, O.get_sigmas_karras(n, SIGMA_MIN, SIGMA_MAX, RHO), rtol=2e-6, atol=0)
        assert s[-1] == 0 and s.dtype == np.float32


@pytest.mark.parametrize("sampler,kw", [("euler", {}), ("heun", {}), ("dpm", {}), ("ancestral", {}),
                                        ("heun", dict(s_churn=4.0, s_tmin=0.05, s_tmax=50.0, s_noise=1.003))])
def test_sampler_loops_match_oracle(sampler, kw):
    shape = (2, 1, 7, 5)
    noise = _noise(8, shape, seed=3)
    x0 = (_noise(1, shape, seed=9)[0] * SIGMA_MAX).astype(np.float32)
    steps = 5
    sig_np = O.get_sigmas_karras(steps, SIGMA_MIN, SIGMA_MAX, RHO)
    ref = O.ode_samplers(_denoiser_np, x0, sig_np, noise, sampler, **kw)
    fn = {"euler": host.sample_euler, "heun": host.sample_heun, "dpm": host.sample_dpm,
          "ancestral": host.sample_euler_ancestral}[sampler]
    gen = _Gen(noise)
    got = fn(_denoiser_t, torch.from_numpy(x0), host.get_sigmas_karras(steps, SIGMA_MIN, SIGMA_MAX, RHO), gen, **kw)
    np.testing.assert_allclose(got.numpy(), ref, rtol=2e-4, atol=2e-4)
    # the reference draws one eps per iteration for heun/dpm (even without churn) and one per step for ancestral
    assert gen.i == {"euler": 0, "heun": steps, "dpm": steps, "ancestral": steps}[sampler]


def test_progdist_is_euler_without_the_trailing_zero():
    """karras_diffusion.py:856-888 (sampler "progdist", steps + 1 sigmas requested for it :529-530): Euler steps that stop at the last
    non-zero sigma, no noise draws."""
    shape = (2, 1, 7, 5)
    x0 = (_noise(1, shape, seed=11)[0] * SIGMA_MAX).astype(np.float32)
    steps = 4
    sig_np = O.get_sigmas_karras(steps + 1, SIGMA_MIN, SIGMA_MAX, RHO)
    ref = O.ode_samplers(_denoiser_np, x0, sig_np[:-1], [], "euler")
    gen = _Gen([])
    got = host.sample_progdist(_denoiser_t, torch.from_numpy(x0), host.get_sigmas_karras(steps + 1, SIGMA_MIN, SIGMA_MAX, RHO), gen)
    np.testing.assert_allclose(got.numpy(), ref, rtol=2e-4, atol=2e-4)
    assert gen.i == 0


def test_stochastic_iterative_sampler_schedule():
    """karras_diffusion.py:830-854 with a general ts: evaluation sigmas and re-noising follow the oracle's schedule."""
    shape = (1, 1, 4, 3)
    noise = _noise(4, shape, seed=5)
    x = torch.from_numpy(_noise(1, shape, seed=6)[0] * SIGMA_MAX)
    seen = []

    def den(xx, sigma):
        seen.append(float(sigma[0]))
        return _denoiser_t(xx, sigma)

    ts, steps = (0, 1, 3), 4
    out = host.stochastic_iterative_sampler(den, x, None, _Gen(noise), ts, t_min=SIGMA_MIN, t_max=SIGMA_MAX, rho=RHO, steps=steps)

    class Cfg:
        sigma_min, sigma_max, rho = SIGMA_MIN, SIGMA_MAX, RHO
    sig, std = O.multistep_schedule(None, Cfg, ts, steps)
    np.testing.assert_allclose(seen, sig, rtol=1e-5)
    xr = x.numpy()
    for i, s in enumerate(sig):
        xr = _denoiser_np(xr, np.full((1,), s, np.float32)) + noise[i] * np.float32(std[i])
    np.testing.assert_allclose(out.numpy(), xr, rtol=1e-4, atol=1e-4)
