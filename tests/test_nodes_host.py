"""Host-side logic of the ComfyUI-facing layer, with ComfyUI stubbed the way the reference's
own tests stub it (tests/test_reshape_mask.py:18-54).  Mirrors reference tests
test_min_step_frac.py, test_node_params.py and the override restore semantics."""
import importlib
import sys
import types

import pytest

RETIRED = ["LanPaint_Beta", "LanPaint_Friction", "LanPaint_EarlyStop", "LanPaint_InnerThreshold",
           "LanPaint_InnerPatience", "LanPaint_MinStepFrac"]


@pytest.fixture()
def nodes(monkeypatch, hip_lib):
    comfy_mod = types.ModuleType("comfy")
    comfy_mod.__path__ = []
    samplers = types.ModuleType("comfy.samplers")

    class _KSAMPLER:
        def sample(self, *a, **k):
            return "orig_sample"

    class _CFGGuider:
        def outer_sample(self, *a, **k):
            return "orig_outer"

        def predict_noise(self, *a, **k):
            return "orig_predict"

    class _KSampler:
        SCHEDULERS = ["normal", "karras"]

    samplers.KSAMPLER, samplers.CFGGuider, samplers.KSampler = _KSAMPLER, _CFGGuider, _KSampler
    model_base = types.ModuleType("comfy.model_base")
    model_base.ModelType = types.SimpleNamespace(FLUX="FLUX", FLOW="FLOW")
    model_base.WAN22 = type("WAN22", (), {})
    helpers = types.ModuleType("comfy.sampler_helpers")
    helpers.prepare_mask = lambda noise_mask, shape, device: "orig_prepare_mask"
    ver = types.ModuleType("comfyui_version")
    ver.__version__ = "0.6.0"
    comfy_mod.samplers, comfy_mod.model_base, comfy_mod.sampler_helpers = samplers, model_base, helpers
    for name, mod in (("comfy", comfy_mod), ("comfy.samplers", samplers), ("comfy.model_base", model_base),
                      ("comfy.sampler_helpers", helpers), ("comfyui_version", ver)):
        monkeypatch.setitem(sys.modules, name, mod)
    sys.modules.pop("lanpaint_amd.nodes", None)
    mod = importlib.import_module("lanpaint_amd.nodes")
    yield mod
    sys.modules.pop("lanpaint_amd.nodes", None)


def test_min_step_frac_effective_steps(nodes):
    f = nodes.min_step_frac_effective_steps
    assert f(5, 0.1, 0.0) == 5 and f(5, 0.01, 0.0) == 5
    assert f(5, 0.2, 0.05) == 5 and f(5, 0.05, 0.05) == 5
    assert f(5, 0.04, 0.05) == 4 and f(5, 0.025, 0.05) == 2 and f(5, 0.005, 0.05) == 0 and f(5, 0.0, 0.05) == 0
    assert f(0, 0.01, 0.05) == 0


def test_retired_params_hidden_and_widgets_kept(nodes):
    for cls in (nodes.LanPaint_KSampler, nodes.LanPaint_KSamplerAdvanced, nodes.LanPaint_SamplerCustom,
                nodes.LanPaint_SamplerCustomAdvanced):
        req = cls.INPUT_TYPES().get("required", {})
        for name in RETIRED:
            assert name not in req
        assert cls.FUNCTION == "sample" and "LATENT" in cls.RETURN_TYPES
    assert set(nodes.LanPaint_KSamplerAdvanced.INPUT_TYPES()["hidden"]) >= set(RETIRED)
    assert set(nodes.LanPaint_SamplerCustomAdvanced.INPUT_TYPES()["hidden"]) >= set(RETIRED)
    assert "LanPaint_MinStepFrac" in nodes.LanPaint_KSampler.INPUT_TYPES()["hidden"]
    req = nodes.LanPaint_KSamplerAdvanced.INPUT_TYPES()["required"]
    for name in ("LanPaint_NumSteps", "LanPaint_Lambda", "LanPaint_StepSize", "LanPaint_PromptMode", "LanPaint_Info",
                 "Inpainting_mode"):
        assert name in req
    assert set(nodes.NODE_CLASS_MAPPINGS) == {"LanPaint_KSampler", "LanPaint_KSamplerAdvanced", "LanPaint_SamplerCustom",
                                              "LanPaint_SamplerCustomAdvanced", "LanPaint_MaskBlend",
                                              "LanPaint_ImageEncode", "LanPaint_ImageDecode"}
    assert set(nodes.NODE_DISPLAY_NAME_MAPPINGS) == set(nodes.NODE_CLASS_MAPPINGS)
    enc, dec = nodes.NODE_CLASS_MAPPINGS["LanPaint_ImageEncode"], nodes.NODE_CLASS_MAPPINGS["LanPaint_ImageDecode"]
    assert enc.FUNCTION == "encode" and enc.RETURN_TYPES == ("LATENT",) and set(enc.INPUT_TYPES()["optional"]) == {"mask"}
    assert dec.FUNCTION == "decode" and dec.RETURN_TYPES == ("IMAGE",)
    assert dec.INPUT_TYPES()["optional"]["blend_overlap"][1] == dict(
        dec.INPUT_TYPES()["optional"]["blend_overlap"][1], default=9, min=1, max=51, step=2)
    mb = nodes.NODE_CLASS_MAPPINGS["LanPaint_MaskBlend"]
    assert mb.FUNCTION == "blend_images" and mb.INPUT_TYPES()["required"]["blend_overlap"][1]["max"] == 51
    assert nodes.LanPaint_KSampler.INPUT_TYPES()["required"]["LanPaint_NumSteps"][1]["default"] == 5


def test_sanitize_param(nodes):
    s = nodes._sanitize_param
    allowed = ("Image First", "Prompt First")
    assert s("Image First", "Image First", allowed=allowed) == "Image First"
    assert s("Prompt First", "Image First", allowed=allowed) == "Prompt First"
    assert s(1.0, "Image First", allowed=allowed) == "Image First"
    assert s("bogus", "Image First", allowed=allowed) == "Image First"
    assert s(None, "Image First", allowed=allowed) == "Image First"
    assert s(5, 5) == 5 and s(3.7, 0.2) == 3.7 and s("abc", 0.2) == 0.2 and s(None, 0.2) == 0.2 and s(True, 5) == 5


def test_pinned_hyperparams_match_reference_defaults(nodes):
    m = types.SimpleNamespace()
    nodes._pin_hyperparams(m, 7.5, 5, "Image First")
    assert (m.LanPaint_StepSize, m.LanPaint_Lambda, m.LanPaint_Beta, m.LanPaint_NumSteps) == (0.2, 5.0, 1.0, 5)
    assert (m.LanPaint_MinStepFrac, m.LanPaint_Friction, m.LanPaint_EarlyStop) == (1.0, 15.0, 1)
    assert (m.LanPaint_InnerThreshold, m.LanPaint_InnerPatience, m.LanPaint_cfg_BIG) == (0.0, 1, 7.5)
    nodes._pin_hyperparams(m, 7.5, 3, "Prompt First", lamb=8.0, step_size=0.15)
    assert m.LanPaint_cfg_BIG == -0.5 and m.LanPaint_Lambda == 8.0 and m.LanPaint_StepSize == 0.15


def test_override_sample_function_patches_and_restores(nodes):
    import comfy
    g, k, h = comfy.samplers.CFGGuider, comfy.samplers.KSAMPLER, comfy.sampler_helpers
    orig = (g.outer_sample, g.predict_noise, k.sample, h.prepare_mask)
    with nodes.override_sample_function():
        assert g.outer_sample is nodes.CFGGuider_LanPaint.outer_sample
        assert g.predict_noise is nodes.CFGGuider_LanPaint.predict_noise
        assert k.sample is nodes.KSAMPLER.sample
        assert h.prepare_mask is not orig[3]
        with nodes.override_sample_function():           # nested entry must not capture the patches as originals
            assert k.sample is nodes.KSAMPLER.sample
        assert k.sample is nodes.KSAMPLER.sample
    assert (g.outer_sample, g.predict_noise, k.sample, h.prepare_mask) == orig
    with pytest.raises(ValueError):
        with nodes.override_sample_function():
            raise ValueError("boom")
    assert (g.outer_sample, g.predict_noise, k.sample, h.prepare_mask) == orig     # restored in `finally`


def test_detect_minimax_h3_audio(nodes):
    class DM:
        sigma_shift_video, sigma_shift_audio = 12.0, 3.0
    patcher = types.SimpleNamespace(model=types.SimpleNamespace(diffusion_model=DM()))
    shapes = [(1, 24, 37, 30, 54), (1, 32, 2, 207)]
    assert nodes._detect_minimax_h3_audio(patcher, {}, shapes[:1]) is None
    assert nodes._detect_minimax_h3_audio(patcher, {}, None) is None
    assert nodes._detect_minimax_h3_audio(types.SimpleNamespace(model=object()), {}, shapes) is None
    assert nodes._detect_minimax_h3_audio(patcher, {}, shapes) == (shapes, 12.0, 3.0)
    opts = {"transformer_options": {"minimax_h3_sigma_shift_video": 10.0, "minimax_h3_sigma_shift_audio": 2.5}}
    assert nodes._detect_minimax_h3_audio(patcher, opts, shapes) == (shapes, 10.0, 2.5)
    assert nodes.time_shift_sigma is None


def test_reshape_mask_refuses_without_hip(nodes):
    import torch
    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nodes.reshape_mask(torch.zeros(4, 4), (1, 4, 8, 8))


def test_sampling_function_returns_fused_heads_or_eager_tuple(nodes, monkeypatch):
    import torch
    import comfy
    calls = {"cfg": 0}

    def calc_cond_batch(model, conds, x, timestep, model_options):
        return [x * 2.0, x * 0.5 if conds[1] is not None else torch.zeros_like(x)]

    def cfg_function(model, cond_pred, uncond_pred, cond_scale, x, timestep, model_options={}, cond=None, uncond=None):
        calls["cfg"] += 1
        return uncond_pred + (cond_pred - uncond_pred) * cond_scale

    monkeypatch.setattr(comfy.samplers, "calc_cond_batch", calc_cond_batch, raising=False)
    monkeypatch.setattr(comfy.samplers, "cfg_function", cfg_function, raising=False)
    x = torch.arange(6.0).reshape(1, 1, 2, 3)
    out = nodes.sampling_function_LanPaint(None, x, torch.tensor([1.0]), "neg", "pos", 5.0, -0.5, model_options={})
    assert isinstance(out, nodes.FusedCFGHeads) and calls["cfg"] == 0
    x0, x0_big = out                                    # behaves like the reference's tuple
    assert torch.allclose(x0, x * 0.5 + (x * 2.0 - x * 0.5) * 5.0)
    assert torch.allclose(x0_big, x * 0.5 + (x * 2.0 - x * 0.5) * -0.5)
    assert len(out) == 2 and out[0] is x0
    # custom cfg hooks present -> the eager reference path
    out2 = nodes.sampling_function_LanPaint(None, x, torch.tensor([1.0]), "neg", "pos", 5.0, -0.5,
                                            model_options={"sampler_post_cfg_function": [lambda a: a["denoised"]]})
    assert isinstance(out2, tuple) and calls["cfg"] == 2
    # cond_scale == 1 drops the uncond pass exactly like the reference (nodes.py:162-165)
    seen = {}
    monkeypatch.setattr(comfy.samplers, "calc_cond_batch",
                        lambda model, conds, x, t, mo: seen.setdefault("conds", conds) and [x, torch.zeros_like(x)], raising=False)
    nodes.sampling_function_LanPaint(None, x, torch.tensor([1.0]), "neg", "pos", 1.0, 1.0, model_options={})
    assert seen["conds"] == ["pos", None]
    seen.clear()
    nodes.sampling_function_LanPaint(None, x, torch.tensor([1.0]), "neg", "pos", 1.0, 1.0,
                                     model_options={"disable_cfg1_optimization": True})
    assert seen["conds"] == ["pos", "neg"]


def test_image_encode_decode_nodes_plain_paths_need_no_gpu(nodes):
    """Without a mask both nodes are a plain VAE call (reference nodes.py:1261-1266, 1323-1330); wrong latent rank
    raises like the reference."""
    import pytest
    import torch

    class VAE:
        def __init__(self, rank=4):
            self.rank = rank

        def encode(self, image):
            b, h, w, _c = image.shape
            z = torch.zeros((b, 4, h // 8, w // 8))
            return z if self.rank == 4 else (z.unsqueeze(2) if self.rank == 5 else z[0])

        def decode(self, z):
            return torch.ones((z.shape[0], z.shape[-2] * 8, z.shape[-1] * 8, 3))

    img = torch.rand(1, 64, 48, 3)
    latent, = nodes.LanPaint_ImageEncode().encode(img, VAE())
    assert set(latent) == {"samples"} and tuple(latent["samples"].shape) == (1, 4, 8, 6)
    with pytest.raises(ValueError, match="4D or 5D"):
        nodes.LanPaint_ImageEncode().encode(img, VAE(rank=3))
    same, = nodes.LanPaint_ImageEncode().encode(img, VAE(), mask=torch.ones(8, 6))      # already at the latent size
    assert tuple(same["noise_mask"].shape) == (1, 1, 8, 6)
    vid, = nodes.LanPaint_ImageEncode().encode(img, VAE(rank=5), mask=torch.ones(1, 8, 6))
    assert tuple(vid["noise_mask"].shape) == (1, 1, 1, 8, 6)
    out, = nodes.LanPaint_ImageDecode().decode(latent, VAE())
    assert tuple(out.shape) == (1, 64, 48, 3)
    resized, = nodes.LanPaint_ImageDecode().decode(latent, VAE(), image=torch.rand(1, 60, 50, 3))
    assert tuple(resized.shape) == (1, 60, 50, 3)


def test_aten_randn_policy_matches_what_torch_consumed_on_the_mi355x(hip_lib):
    """(block * grid, generator-offset increment) of one torch.randn call, as measured on the MI355X (256 CUs, 2048
    threads per CU) while pinning LP_RNG_TORCH against torch.randn itself (tests/test_gpu_kernels.py)."""
    from lanpaint_amd.lanpaint import aten_randn_policy
    mi355x = (256, 2048)
    assert aten_randn_policy(1, *mi355x) == (256, 4)
    assert aten_randn_policy(257, *mi355x) == (512, 4)
    assert aten_randn_policy(65536, *mi355x) == (65536, 4)                  # SDXL latent: one value per thread
    assert aten_randn_policy(2096640, *mi355x) == (524288, 4)               # Wan latent: the grid cap, 4 values per thread
    assert aten_randn_policy(4 * 2096640 + 3, *mi355x) == (524288, 16)
    assert aten_randn_policy(16 * 2096640, *mi355x) == (524288, 64)


def test_per_tensor_caches_work_on_inference_tensors(nodes):
    """ComfyUI runs its nodes under torch.inference_mode(): inference tensors raise on `._version`.  The caches keyed on
    a tensor (the run's noise verdict lanpaint.py:51, the binarised mask nodes.py:281-283) must identify such a tensor
    by object + address instead of crashing on the first sigma call."""
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd.lanpaint import tensor_version
    with torch.inference_mode():
        noise = torch.randn(1, 4, 8, 8)
        zero = torch.zeros(1, 4, 8, 8)
        dm = (torch.rand(1, 4, 8, 8) > 0.5).float()
        assert noise.is_inference() and tensor_version(noise) == -1
        with pytest.raises(RuntimeError):
            noise._version
        eng = LanPaint(lambda *a, **k: None, 5, 15.0, 5.0, 1.0, 0.2)
        assert eng._noise_is_zero(noise) is False and eng._noise_is_zero(noise) is False     # second call: cache hit
        assert eng._noise_is_zero(zero) is True
        k = nodes.KSamplerX0Inpaint(None, torch.linspace(10, 0, 5))
        m1 = k._latent_mask(dm)
        assert k._latent_mask(dm) is m1 and torch.equal(m1, 1 - (dm > 0.5).float())
        assert k._latent_mask(dm.clone()) is not m1
    assert tensor_version(torch.zeros(2)) == 0


def test_native_inner_step_rule_equals_the_python_rule(nodes, hip_lib):
    """lp_effective_inner_steps (what lp_node_call applies between the device's answer and the graph launch) against
    nodes.py:286-299 + min_step_frac_effective_steps (nodes.py:134-144) as Python evaluates them: the reference's own KAT
    table, every half-way case of round() (ties to even), and a seeded sweep; float32 scalars as the mailbox delivers them."""
    import random
    import numpy as np
    f = nodes.min_step_frac_effective_steps

    def py(n, step_f, frac, total, early, msf):
        if total - int(step_f) <= early:
            return 0
        return f(n, frac, msf)

    c = hip_lib.lp_effective_inner_steps
    for n, frac, msf, want in [(5, 0.1, 0.0, 5), (5, 0.2, 0.05, 5), (5, 0.05, 0.05, 5), (5, 0.04, 0.05, 4), (5, 0.025, 0.05, 2),
                               (5, 0.005, 0.05, 0), (5, 0.0, 0.05, 0), (0, 0.01, 0.05, 0)]:
        assert c(n, 0.0, frac, 100, 1, msf) == want == f(n, frac, msf)
    for n in range(0, 13):                      # exact ties: n * frac / msf = k + 0.5
        for k in range(0, 12):
            frac = (k + 0.5) / max(n, 1) * 0.5
            assert c(n, 3.0, frac, 30, 1, 0.5) == py(n, 3.0, frac, 30, 1, 0.5)
    rnd = random.Random(3)
    for _ in range(20000):
        n, total, early = rnd.randint(0, 12), rnd.randint(1, 60), rnd.choice([0, 1, 2, 5])
        step_f = float(rnd.randint(0, total))
        frac = float(np.float32(rnd.random() * 1.2))
        msf = rnd.choice([0.0, 0.05, 0.3, 0.5, 1.0, rnd.random()])
        assert c(n, step_f, frac, total, early, msf) == py(n, step_f, frac, total, early, msf)
