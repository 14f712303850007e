"""Differential test of the oracle against the UNMODIFIED reference, live, where the reference tree is present (the build
container; skipped on the GPU box and in any checkout without /root/reference -- the committed fixtures pin the oracle there).
hypothesis draws engine calls the fixtures do not enumerate: shapes (4-D / 5-D, odd sizes, batch rows on their own sigmas),
VE and flow, hard / soft / all-known / all-inpaint masks, inner-step counts 0..4, Lambda / Beta / StepSize / MinStepFrac, every
model-output form, the inner early stop with its trace.  Both sides consume ONE recorded xi stream (the reference's
torch.randn_like is wrapped, never its source copied); outputs must agree to fp32 rounding."""
import os
import sys
import warnings

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "LanPaint")), reason="the reference tree is not present")


def _reference():
    import importlib
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore", message=".*autocast.*")
    return importlib.import_module("src.LanPaint.lanpaint").LanPaint


class _Sampling:
    def __init__(self, flow):
        self.flow, self.noise_scale = flow, 1.0

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return sigma * noise + (1.0 - sigma) * latent_image if self.flow else latent_image + noise * sigma


class _Model:
    """x -> heads in one of the forms lanpaint.py:34-43 accepts; arithmetic that works on numpy arrays and torch tensors alike."""

    def __init__(self, flow, form):
        self.inner_model, self.model_sampling, self.form, self.calls = self, _Sampling(flow), form, 0

    def __call__(self, x, t, model_options=None, seed=None):
        self.calls += 1
        tb = t.reshape((-1,) + (1,) * (x.ndim - 1))
        a, b = x * 0.9 - 0.05 * tb, x * 0.8 + 0.1
        return {"tuple": (a, b), "list1": [a], "single": a, "triple": (a, b, b)}[self.form]


shapes = st.sampled_from([(1, 4, 6, 6), (2, 4, 5, 7), (3, 2, 4, 4), (1, 3, 3, 4, 5), (2, 2, 2, 3, 3), (1, 1, 9)])


@settings(max_examples=int(__import__("os").environ.get("LP_FUZZ_EXAMPLES", "80")), deadline=None, suppress_health_check=list(HealthCheck))
@given(shape=shapes, flow=st.booleans(), n_steps=st.integers(0, 4), mask_kind=st.sampled_from(["box", "random", "soft", "ones", "zeros"]),
       form=st.sampled_from(["tuple", "list1", "single", "triple"]), lamb=st.sampled_from([5.0, 0.5, 12.0]), beta=st.sampled_from([1.0, 0.5, 2.0]),
       step=st.sampled_from([0.2, 0.05, 0.6]), msf=st.sampled_from([0.0, 0.3, 1.0]), row_sigmas=st.booleans(),
       stop=st.sampled_from([None, (3.0, 1), (0.4, 2), (1e-9, 1)]), seed=st.integers(0, 10_000))
def test_oracle_equals_the_reference_on_random_calls(shape, flow, n_steps, mask_kind, form, lamb, beta, step, msf, row_sigmas, stop, seed):
    import torch
    from oracle.lanpaint_oracle import OracleLanPaint, times_from_sigma
    Ref = _reference()
    rng = np.random.default_rng(seed)
    b = shape[0]
    base = float(rng.uniform(0.05, 0.95)) if flow else float(np.exp(rng.uniform(np.log(0.03), np.log(14.0))))
    scale = rng.uniform(0.6, 1.0, size=b).astype(np.float32) if (row_sigmas and b > 1) else np.ones(b, np.float32)
    sigma = (np.float32(base) * scale).astype(np.float32)
    y, noise = rng.standard_normal(shape, dtype=np.float32), rng.standard_normal(shape, dtype=np.float32)
    sb = sigma.reshape((-1,) + (1,) * (len(shape) - 1))
    x = (sb * noise + (1 - sb) * y).astype(np.float32) if flow else (y + noise * sb).astype(np.float32)
    mask = {"box": lambda: np.concatenate([np.ones(shape[:-1] + (shape[-1] // 2,), np.float32), np.zeros(shape[:-1] + (shape[-1] - shape[-1] // 2,), np.float32)], -1),
            "random": lambda: (rng.random(shape) > 0.5).astype(np.float32), "soft": lambda: rng.random(shape, dtype=np.float32),
            "ones": lambda: np.ones(shape, np.float32), "zeros": lambda: np.zeros(shape, np.float32)}[mask_kind]()
    times = tuple(np.asarray(t, dtype=np.float32) for t in times_from_sigma(sigma, flow))
    mo = None
    if stop is not None:
        mo = {"lanpaint_semantic_stop": {"threshold": stop[0], "patience": stop[1]}, "lanpaint_semantic_trace": []}
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731

    # the reference, its randn_like recorded
    draws, orig = [], torch.randn_like

    def rec(t, *a, **k):
        d = orig(t, *a, **k)
        draws.append(d.numpy().copy())
        return d
    ref_model = _Model(flow, form)
    ref = Ref(ref_model, 5, 15.0, lamb, beta, step, IS_FLUX=False, IS_FLOW=flow, MinStepFrac=msf)
    xr = tt(x.copy())
    mo_r = None if mo is None else {"lanpaint_semantic_stop": dict(mo["lanpaint_semantic_stop"]), "lanpaint_semantic_trace": []}
    torch.manual_seed(seed)
    torch.randn_like = rec
    try:
        out_r = ref(xr, tt(y), tt(noise), tt(sigma), tt(mask), tuple(tt(t) for t in times), mo_r, 0, n_steps=n_steps)
    finally:
        torch.randn_like = orig

    it = iter(draws)
    o_model = _Model(flow, form)
    o = OracleLanPaint(o_model, 5, 15.0, lamb, beta, step, is_flow=flow, min_step_frac=msf, randn=lambda like: next(it))
    xo = x.copy()
    out_o = o(xo, y, noise, sigma, mask, times, mo, 0, n_steps=n_steps)
    assert sum(1 for _ in it) == 0 and o_model.calls == ref_model.calls
    scale_x = max(1.0, float(np.abs(xr.numpy()).max()))
    np.testing.assert_allclose(xo, xr.numpy(), atol=2e-5 * scale_x, rtol=0)
    np.testing.assert_allclose(np.asarray(out_o), out_r.numpy(), atol=2e-5 * max(1.0, float(np.abs(out_r.numpy()).max())), rtol=0)
    if mo is not None:
        tr_o, tr_r = mo["lanpaint_semantic_trace"], mo_r["lanpaint_semantic_trace"]
        assert len(tr_o) == len(tr_r)
        for a, c in zip(tr_o, tr_r):
            assert (a["inner_step"], a["patience_counter"], a["stopped"], a["patience_eff"]) == (c["inner_step"], c["patience_counter"], c["stopped"], c["patience_eff"])
            assert sorted(a) == sorted(c)
            for key in ("dist", "dist_inpaint", "dist_ring", "dist_drift", "threshold_eff", "abt"):
                assert (a[key] is None) == (c[key] is None), key
                if a[key] is not None:
                    assert a[key] == pytest.approx(c[key], rel=5e-4, abs=1e-9), key


def _reference_nodes(version="0.6.0"):
    """The reference's nodes.py with ComfyUI stubbed the way its own tests stub it (tests/test_reshape_mask.py:18-54)."""
    import importlib
    import types
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    comfy_mod = types.ModuleType("comfy")
    comfy_mod.__path__ = []
    utils = types.ModuleType("comfy.utils")

    def repeat_to_batch_size(t, b):                       # comfy.utils semantics: narrow when larger, tile + narrow when smaller
        if t.shape[0] > b:
            return t[:b]
        if t.shape[0] < b:
            import math
            return t.repeat([math.ceil(b / t.shape[0])] + [1] * (t.ndim - 1))[:b]
        return t
    utils.repeat_to_batch_size = repeat_to_batch_size
    samplers = types.ModuleType("comfy.samplers")
    samplers.KSAMPLER = type("KSAMPLER", (), {})
    samplers.KSampler = type("KSampler", (), {"SCHEDULERS": ["karras"]})
    mb = types.ModuleType("comfy.model_base")
    mb.ModelType = types.SimpleNamespace(FLUX="FLUX", FLOW="FLOW")
    mb.WAN22 = type("WAN22", (), {})
    ver = types.ModuleType("comfyui_version")
    ver.__version__ = version
    comfy_mod.utils, comfy_mod.samplers, comfy_mod.model_base = utils, samplers, mb
    saved = {k: sys.modules.get(k) for k in ("comfy", "comfy.utils", "comfy.samplers", "comfy.model_base", "nodes", "latent_preview",
                                             "comfyui_version", "src.LanPaint.nodes")}
    for name, mod in (("comfy", comfy_mod), ("comfy.utils", utils), ("comfy.samplers", samplers), ("comfy.model_base", mb),
                      ("nodes", types.ModuleType("nodes")), ("latent_preview", types.ModuleType("latent_preview")), ("comfyui_version", ver)):
        sys.modules[name] = mod
    sys.modules.pop("src.LanPaint.nodes", None)
    try:
        return importlib.import_module("src.LanPaint.nodes")
    finally:
        for k, v in saved.items():                       # leave no stub behind for the other tests of the session
            if k == "src.LanPaint.nodes":
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.parametrize("version", ["0.6.0", "0.5.0"])
def test_oracle_reshape_mask_equals_the_reference_on_random_masks(version):
    """reshape_mask (nodes.py:59-133) on the mask layouts ComfyUI produces -- [H, W], [B, H, W], [F, H, W] / [F, 1, H, W] video
    masks, the audio forms [F] and [1, 1, F, 1] -- to random latent shapes, both sides of the 0.6.0 gate: BIT-equal (the oracle
    carries the CPU index rules of the torch kernel the reference's F.interpolate runs on a host mask)."""
    import torch
    from oracle import lanpaint_oracle as orc
    ref = _reference_nodes(version)
    new = version == "0.6.0"
    rng = np.random.default_rng(7 if new else 8)
    checked = 0
    for _ in range(400):
        video = bool(rng.integers(0, 2))
        five_d = video or bool(rng.integers(0, 3) == 0)
        b, c = int(rng.integers(1, 3)), int(rng.integers(1, 5))
        h, w = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        H, W = int(rng.integers(2, 150)), int(rng.integers(2, 150))
        if video:
            f, F = int(rng.integers(1, 8)), int(rng.integers(1, 30))
            out_shape = (b, c, f, h, w)
            form = int(rng.integers(0, 3))
            m = rng.random((F, H, W)) if form == 0 else (rng.random((F, 1, H, W)) if form == 1 else rng.random((H, W)))
        elif five_d:
            f = int(rng.integers(1, 6))
            out_shape = (b, c, f, h, w)
            m = rng.random((H, W)) if rng.integers(0, 2) else rng.random((int(rng.integers(1, 3)), H, W))
        else:
            out_shape = (b, c, h, w)
            form = int(rng.integers(0, 4))
            if form == 0:
                m = rng.random((H, W))
            elif form == 1:
                m = rng.random((int(rng.integers(1, 3)), H, W))
            elif form == 2:
                m = rng.random((int(rng.integers(2, 200)),))                          # audio [F] at video frame rate
            else:
                m = rng.random((1, 1, int(rng.integers(2, 200)), 1))                  # ... as SetLatentNoiseMask reshapes it
        m = (m > 0.5).astype(np.float32)
        try:
            want = ref.reshape_mask(torch.from_numpy(m), out_shape, video_inpainting=video).numpy()
        except Exception:
            continue                                       # (a layout the reference itself rejects: nothing to mirror)
        got = orc.reshape_mask(m, out_shape, video_inpainting=video, comfy_060_or_newer=new, mask_on="cpu")
        assert got.shape == want.shape, (m.shape, out_shape, video, got.shape, want.shape)
        assert np.array_equal(got, want), (m.shape, out_shape, video)
        checked += 1
    assert checked >= 250
    # the inner-step rule (nodes.py:134-144) on a dense grid, banker's rounding included
    for n in range(0, 12):
        for frac in np.linspace(0.0, 1.0, 41):
            for mf in (0.0, 0.05, 0.3, 1.0):
                assert orc.min_step_frac_effective_steps(n, float(frac), mf) == ref.min_step_frac_effective_steps(n, float(frac), mf)
