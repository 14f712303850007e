"""End-to-end through the ComfyUI-facing glue on the GPU with ComfyUI stubbed (the reference's own test
technique): override_sample_function -> KSAMPLER.sample -> KSamplerX0Inpaint -> HIP engine, driven by a
k-diffusion-style Euler sampler function, compared with the CPU oracle walking the same schedule."""
import importlib
import sys
import types

import numpy as np
import pytest

from oracle import lanpaint_oracle as orc
from tests import golden_cases as gc
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None):
    """k-diffusion sample_euler, the shape ComfyUI's sampler functions have."""
    extra_args = extra_args or {}
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "denoised": denoised})
        x = x + (x - denoised) / sigmas[i] * (sigmas[i + 1] - sigmas[i])
    return x


@pytest.fixture()
def glue(monkeypatch, hip_lib):
    comfy_mod = types.ModuleType("comfy")
    comfy_mod.__path__ = []
    samplers = types.ModuleType("comfy.samplers")

    class KSAMPLER:
        def __init__(self, sampler_function, extra_options={}, inpaint_options={}):
            self.sampler_function, self.extra_options, self.inpaint_options = sampler_function, extra_options, inpaint_options

        def max_denoise(self, model_wrap, sigmas):
            return False

        def sample(self, *a, **k):
            raise AssertionError("stock KSAMPLER.sample must be replaced inside override_sample_function")

    class CFGGuider:
        def outer_sample(self, *a, **k):
            return "orig"

        def predict_noise(self, *a, **k):
            return "orig"

    samplers.KSAMPLER, samplers.CFGGuider = KSAMPLER, CFGGuider
    samplers.KSampler = type("KSampler", (), {"SCHEDULERS": ["karras"]})
    model_base = types.ModuleType("comfy.model_base")
    model_base.ModelType = types.SimpleNamespace(FLUX="FLUX", FLOW="FLOW", EPS="EPS")
    model_base.WAN22 = type("WAN22", (), {})
    helpers = types.ModuleType("comfy.sampler_helpers")
    helpers.prepare_mask = lambda noise_mask, shape, device: noise_mask
    ver = types.ModuleType("comfyui_version")
    ver.__version__ = "0.6.0"
    comfy_mod.samplers, comfy_mod.model_base, comfy_mod.sampler_helpers = samplers, model_base, helpers
    for name, mod in (("comfy", comfy_mod), ("comfy.samplers", samplers), ("comfy.model_base", model_base),
                      ("comfy.sampler_helpers", helpers), ("comfyui_version", ver)):
        monkeypatch.setitem(sys.modules, name, mod)
    sys.modules.pop("lanpaint_amd.nodes", None)
    nodes = importlib.import_module("lanpaint_amd.nodes")
    yield nodes, samplers
    sys.modules.pop("lanpaint_amd.nodes", None)


@pytest.mark.parametrize("flow", [False, True])
def test_ksampler_sample_through_override_matches_oracle(glue, flow):
    import torch
    nodes, samplers = glue
    dev = "cuda"
    shape, n_think = (1, 4, 16, 16), 3
    sig = gc.flow_sigmas(6) if flow else gc.karras_sigmas(6, 0.05, 12.0)
    rng = np.random.default_rng(13)
    latent = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    denoise_mask = np.zeros(shape, dtype=np.float32)
    denoise_mask[..., 8:] = 1.0
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(64)]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)   # noqa: E731

    class Sampling:
        def noise_scaling(self, sigma, n, l, max_denoise=False):
            return sigma * n + (1.0 - sigma) * l if flow else l + n * sigma

        def inverse_noise_scaling(self, sigma, latent):
            return latent / (1.0 - sigma) if flow else latent

    calls = {"n": 0}

    class Guider:        # the `model_wrap` KSAMPLER.sample receives (CFGGuider in ComfyUI)
        def __init__(self):
            self.inner_model = types.SimpleNamespace(model_type="FLOW" if flow else "EPS", model_sampling=Sampling())
            self.model_patcher = types.SimpleNamespace(
                LanPaint_NumSteps=n_think, LanPaint_Friction=15.0, LanPaint_Lambda=5.0, LanPaint_Beta=1.0,
                LanPaint_StepSize=0.2, LanPaint_cfg_BIG=-0.5, LanPaint_EarlyStop=1, LanPaint_MinStepFrac=1.0,
                LanPaint_InnerThreshold=0.0, LanPaint_InnerPatience=1)

        def __call__(self, x, sigma, model_options=None, seed=None):      # predict_noise-style dual head
            calls["n"] += 1
            return 0.9 * x, 0.8 * x

    guider = Guider()
    it = iter([tt(d) for d in draws])
    import lanpaint_amd.lanpaint as lp_mod
    real_init = lp_mod.LanPaint.__init__

    def init_with_recorded_xi(self, *a, **k):
        k.setdefault("rng", lambda like: next(it))
        real_init(self, *a, **k)

    lp_mod.LanPaint.__init__ = init_with_recorded_xi
    try:
        with nodes.override_sample_function():
            ks = samplers.KSAMPLER(sample_euler)
            assert samplers.KSAMPLER.sample is nodes.KSAMPLER.sample
            out = ks.sample(guider, tt(sig), {"model_options": {}, "seed": 5}, None, tt(noise), latent_image=tt(latent),
                            denoise_mask=tt(denoise_mask))
        torch.cuda.synchronize()
    finally:
        lp_mod.LanPaint.__init__ = real_init
    assert guider.cfg_BIG == -0.5

    # oracle: the same walk on the CPU
    it_o = iter(draws)
    gcalls = {"n": 0}

    def np_model(x, t, model_options=None, seed=None):
        gcalls["n"] += 1
        return 0.9 * x, 0.8 * x

    o = orc.OracleLanPaint(np_model, n_think, 15.0, 5.0, 1.0, 0.2, is_flow=flow, min_step_frac=1.0,
                           randn=lambda like: next(it_o))
    latent_mask = orc.binarize_and_invert(denoise_mask)
    s0 = np.float32(sig[0])
    x = ((s0 * noise + (1 - s0) * latent) if flow else (latent + noise * s0)).astype(np.float32)
    for i in range(len(sig) - 1):
        s = np.float32([sig[i]])
        times = orc.times_from_sigma(s, flow)
        n_eff = orc.effective_inner_steps(n_think, sig, float(s[0]), float(times[1].mean()), 1, 1.0)
        den = o(x, latent, noise, s, latent_mask, times, None, 5, n_steps=n_eff)
        x = (x + (x - den) / np.float32(sig[i]) * np.float32(sig[i + 1] - sig[i])).astype(np.float32)
    want = x / (1.0 - sig[-1]) if flow else x
    assert calls["n"] == gcalls["n"] > 0
    assert_close(out.cpu().numpy(), want, "KSAMPLER.sample output", rel=1e-4, mse=1e-8)
