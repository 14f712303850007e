"""The reference's own unit tests, re-run against lanpaint_amd on the GPU (same stubs, same
assertions): tests/test_sho_regression.py, test_lanpaint_semantic_stop.py, test_av_schedule.py,
test_reshape_mask.py, test_videomask.py:475-713 -- plus the KSamplerX0Inpaint sampler callable."""
import os

import numpy as np
import pytest

from oracle import lanpaint_oracle as orc
from tests import golden_cases as gc
from tests.helpers import assert_close
from tests.stubs import MODELS

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch
    assert torch.cuda.is_available()


class _DummySampling:
    noise_scale = 1.0

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return latent_image + noise * sigma


class _FlowSampling:
    noise_scale = 1.0

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        assert sigma.numel() == 1, "noise_scaling requires a scalar sigma"
        return sigma * (self.noise_scale * noise) + (1.0 - sigma) * latent_image


class _DummyModel:
    def __init__(self, sampling=None):
        self.inner_model = self
        self.model_sampling = sampling or _DummySampling()
        self.last_input = None
        self.model_type = "EPS"
        self.calls = 0

    def __call__(self, x, sigma, model_options=None, seed=None):
        self.last_input = x
        self.calls += 1
        return x, x


def _engine(model=None, n_steps=1, **kw):
    from lanpaint_amd import LanPaint
    return LanPaint(model or _DummyModel(), NSteps=n_steps, Friction=15.0, Lambda=1.0, Beta=1.0, StepSize=0.2, **kw)


def T(*a, **k):
    import torch
    return torch.tensor(*a, device=DEV, **k)


# ---- reference tests/test_sho_regression.py:6-33 -------------------------------------------
def test_langevin_dynamics_uses_first_order_scheme():
    import torch
    from unittest.mock import MagicMock
    from lanpaint_amd import LanPaint
    torch.manual_seed(0)
    lp = LanPaint(Model=MagicMock(), NSteps=10, Friction=1.0, Lambda=1.0, Beta=1.0, StepSize=0.1)
    x_t = torch.randn(1, 4, 8, 8, device=DEV)
    lp.img_dim_size = 4
    mask = torch.zeros_like(x_t)
    score = lambda x: torch.zeros_like(x)   # noqa: E731
    current_times = (T([0.5]), T([0.5]), T([0.5]))
    x_out, args_out = lp.langevin_dynamics(x_t, score, mask, T([0.1]), current_times, sigma_y=1.0)
    assert hasattr(args_out, "v") and hasattr(args_out, "C") and hasattr(args_out, "x0")
    assert args_out.v is None
    assert args_out[1] is args_out.C and args_out[2] is args_out.x0
    assert torch.isfinite(x_out).all()
    # second call consumes the state (steady scheme) and legacy 2-/3-tuples are accepted
    x2, st2 = lp.langevin_dynamics(x_out, score, mask, T([0.1]), current_times, sigma_y=1.0, args=args_out)
    x3, st3 = lp.langevin_dynamics(x_out, score, mask, T([0.1]), current_times, sigma_y=1.0, args=(None, args_out.C))
    assert torch.isfinite(x2).all() and torch.isfinite(x3).all() and st3.v is None


def test_langevin_dynamics_matches_oracle_iteration():
    """Public per-iteration entry vs the oracle's think_iteration with an arbitrary score."""
    import torch
    rng = np.random.default_rng(0)
    shape = (2, 4, 6, 6)
    xt = rng.standard_normal(shape, dtype=np.float32)
    mask = gc.box_mask(shape)
    abt = np.float32([0.3, 0.7])
    ve = np.sqrt((1 - abt) / abt).astype(np.float32)
    step = (0.2 * (1 - abt)).astype(np.float32)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(3)]
    score_np = lambda x: (0.5 - x) * 0.7   # noqa: E731

    it = iter(draws)
    o = orc.OracleLanPaint(None, 1, 15.0, 5.0, 1.0, 0.2, randn=lambda like: next(it))
    o.ndim = 4
    b = lambda a: a.reshape(-1, 1, 1, 1)   # noqa: E731
    one = np.ones_like(b(abt))
    x1, s1 = o.think_iteration(xt, score_np, mask, b(step), (ve, abt, ve), one, one, None)
    x2, s2 = o.think_iteration(x1, score_np, mask, b(step), (ve, abt, ve), one, one, s1)

    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
    it2 = iter([tt(d) for d in draws])
    from lanpaint_amd import LanPaint
    eng = LanPaint(None, 1, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it2))
    eng.img_dim_size = 4
    times = (tt(ve), tt(abt), tt(ve))
    g1, st1 = eng.langevin_dynamics(tt(xt), lambda x: (0.5 - x) * 0.7, tt(mask), tt(b(step)), times,
                                    sigma_x=tt(one), sigma_y=tt(one), args=None)
    g2, st2 = eng.langevin_dynamics(g1, lambda x: (0.5 - x) * 0.7, tt(mask), tt(b(step)), times,
                                    sigma_x=tt(one), sigma_y=tt(one), args=st1)
    assert_close(g1.cpu().numpy(), x1, "iteration 0 x_t")
    assert_close(st1.C.cpu().numpy(), s1.C, "iteration 0 C", rel=5e-5)
    assert_close(g2.cpu().numpy(), x2, "iteration 1 x_t")
    assert_close(st2.x0.cpu().numpy(), s2.x0, "iteration 1 x0")


# ---- reference tests/test_lanpaint_semantic_stop.py:31-104 ---------------------------------
def _stop_inputs():
    import torch
    x = torch.zeros((1, 4, 8, 8), device=DEV)
    sigma = T([1.0])
    return x, torch.zeros_like(x), torch.ones_like(x), sigma, torch.zeros_like(x), (sigma, T([0.5]), T([0.0]))


@pytest.mark.parametrize("mask_all_known,patience,expected", [(False, 2, 3), (True, 1, 10)])
def test_semantic_stop_with_overridden_langevin(mask_all_known, patience, expected):
    import torch
    engine = _engine(n_steps=10)
    calls = {"langevin": 0, "with_score": 0}

    def fake_langevin(x_t, score, mask, step_size, current_times, sigma_x=1, sigma_y=0, args=None):
        calls["langevin"] += 1
        calls["with_score"] += score is not None
        return x_t, args

    engine.langevin_dynamics = fake_langevin
    mo = {"lanpaint_semantic_stop": {"threshold": 1e-6, "patience": patience}}
    x, latent, noise, sigma, mask, times = _stop_inputs()
    if mask_all_known:
        mask = torch.ones_like(mask)
    engine(x, latent, noise, sigma, mask, times, model_options=mo, seed=0, n_steps=10)
    assert calls["langevin"] == expected and calls["with_score"] == expected


# ---- reference tests/test_av_schedule.py:157-324 ---------------------------------------------
def _flat_pack():
    import torch
    x = torch.zeros(1, 1, 8, device=DEV)
    ai = torch.zeros(1, 1, 8, device=DEV)
    ai[..., 5:] = 1.0
    return (x, torch.zeros_like(x), torch.ones_like(x), T([0.5]), torch.zeros_like(x), (T([1.0]), T([0.5]), T([0.5])),
            (T([0.25]), T([0.9]), T([0.2])), ai)


def test_audio_rows_get_audio_schedule_parameters():
    import torch
    engine = _engine(_DummyModel(_FlowSampling()))
    cap = {}

    def fake_prepare_step_size(current_times, step_size, sigma_x, sigma_y):
        cap["abt"], cap["step_size"] = current_times[1], step_size
        abt = current_times[1]
        ones, z = torch.ones_like(abt), torch.zeros_like(abt)
        return (current_times[0], abt, ones, ones, ones, ones, z, z, z, z)

    engine.prepare_step_size = fake_prepare_step_size
    x, latent, noise, sigma, mask, times, times_a, ai = _flat_pack()
    engine(x, latent, noise, sigma, mask, times, model_options=None, seed=0, n_steps=1, current_times_audio=times_a,
           audio_indicator=ai)
    abt, step = cap["abt"].flatten(), cap["step_size"].flatten()
    assert float(abt[0]) == 0.5 and float(abt[-1]) == pytest.approx(0.9)
    assert float(step[0]) == pytest.approx(0.2 * 0.5) and float(step[-1]) == pytest.approx(0.2 * 0.1)
    cap.clear()
    engine(x, latent, noise, sigma, mask, times, model_options=None, seed=0, n_steps=1)
    assert float(cap["abt"].flatten()[0]) == 0.5 and float(cap["abt"].flatten()[-1]) == 0.5


def test_replace_step_uses_audio_sigma_for_audio_rows():
    import torch
    engine = _engine(_DummyModel(_FlowSampling()), n_steps=0)
    x, latent, noise, sigma, _, times, times_a, ai = _flat_pack()
    engine(x, latent, noise, sigma, torch.ones_like(x), times, model_options=None, seed=0, n_steps=0,
           current_times_audio=times_a, audio_indicator=ai)
    inp = engine.inner_model.last_input.flatten()
    assert float(inp[0]) == pytest.approx(0.5) and float(inp[-1]) == pytest.approx(0.2)


def test_score_model_audio_correction_and_flat_target():
    import torch
    from lanpaint_amd import LanPaint

    class Offset(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            return x + 2.0, x + 2.0

    engine = LanPaint(Offset(_FlowSampling()), NSteps=1, Friction=15.0, Lambda=1.0, Beta=1.0, StepSize=0.2, IS_FLOW=True)
    engine.img_dim_size = 3
    ai = torch.zeros(1, 1, 8, device=DEV)
    ai[..., 5:] = 1.0
    z = torch.zeros(1, 1, 8, device=DEV)
    abt, sig = torch.full((1, 1, 8), 0.5, device=DEV), torch.ones(1, 1, 8, device=DEV)
    tflow = engine.add_none_dims(T([0.5]))
    s = engine.score_model(z, z, z, abt, sig, tflow, model_options=None, seed=0).flatten()
    assert float(s[0]) == pytest.approx(2.0) and float(s[-1]) == pytest.approx(2.0)
    engine.audio_indicator, engine.audio_correction = ai, (1.0 - ai) + 0.625 * ai
    s = engine.score_model(z, z, z, abt, sig, tflow, model_options=None, seed=0).flatten()
    assert float(s[0]) == pytest.approx(2.0) and float(s[-1]) == pytest.approx(1.25)


def test_add_none_dims_and_prepare_step_size_per_row():
    import torch
    engine = _engine()
    for nd in (3, 4, 5):
        engine.img_dim_size = nd
        assert tuple(engine.add_none_dims(torch.zeros(1)).shape) == (1,) + (1,) * (nd - 1)
        assert tuple(engine.add_none_dims(torch.zeros(())).shape) == (1,) * nd
    engine.img_dim_size = 3
    assert tuple(engine.add_none_dims(torch.zeros(1, 1, 8)).shape) == (1, 1, 8)
    abt = torch.full((1, 1, 8), 0.5)
    abt[..., 5:] = 0.9
    step = torch.full((1, 1, 8), 0.1)
    step[..., 5:] = 0.02
    one = torch.ones(1, 1, 8)
    out = engine.prepare_step_size((one, abt, one), step, one, one)
    assert all(t.ndim == 3 for t in out) and len(out) == 10
    adt = (out[6] * out[2]).flatten()
    assert float(adt[0]) == pytest.approx(0.2) and float(adt[-1]) == pytest.approx(0.2)


def test_model_output_forms_and_empty_output_error():
    engine = _engine()
    a, b = object(), object()
    assert engine.unpack_model_output((a, b, 3)) == (a, b)
    assert engine.unpack_model_output([a]) == (a, a)
    assert engine.unpack_model_output(a) == (a, a)
    with pytest.raises(ValueError, match="Model output is empty"):
        engine.unpack_model_output(())


# ---- reference tests/test_reshape_mask.py + tests/test_videomask.py:475-713, on the HIP kernel --------
def test_reshape_mask_reference_kats():
    import torch
    from lanpaint_amd.nodes import prepare_mask, reshape_mask
    z = lambda *s: torch.zeros(*s)   # noqa: E731   (CPU inputs, like ComfyUI hands them over)
    assert tuple(reshape_mask(z(1, 4, 4), (1, 16, 1, 8, 8)).shape) == (1, 16, 1, 8, 8)
    out = prepare_mask(z(4, 4), (2, 3, 8, 8), device=torch.device("cuda"))
    assert tuple(out.shape) == (2, 3, 8, 8) and out.device.type == "cuda"
    m = z(8, 8, 8)
    m[2, 5, 5], m[6, 7, 7] = 1.0, 1.0
    out = reshape_mask(m, (1, 16, 2, 4, 4), video_inpainting=True)
    assert tuple(out.shape) == (1, 16, 2, 4, 4)
    assert out[0, 0, 0].max() == 1.0 and out[0, 0, 1].max() == 1.0
    assert out[0, 0, 0, 2, 2] == 1.0 and out[0, 0, 1, 3, 3] == 1.0
    m = z(8, 8, 8)
    m[3, 5, 5] = 1.0
    assert reshape_mask(m, (1, 16, 2, 4, 4), video_inpainting=True).max() == 0.0
    m = z(3, 8, 8)
    m[1, 5, 5] = 1.0
    assert reshape_mask(m, (1, 16, 1, 4, 4), video_inpainting=True).max() == 1.0
    m = z(16, 4, 4)
    m[6, 1, 1] = 1.0
    out = reshape_mask(m, (1, 24, 4, 4, 4), video_inpainting=True)
    assert out[0, 0, 0].max() == 1.0 and out[0, 0, 3].max() == 1.0
    m = z(124, 864, 480)
    m[62, 100:140, 100:140] = 1.0
    out = reshape_mask(m, (1, 24, 37, 30, 54), video_inpainting=True)
    assert tuple(out.shape) == (1, 24, 37, 30, 54) and out.max() == 1.0 and (out == 0.0).float().mean() > 0.8
    m4 = z(124, 1, 864, 480)
    m4[62, 0, 100:140, 100:140] = 1.0
    assert torch.equal(reshape_mask(m4, (1, 24, 37, 30, 54), video_inpainting=True), out)
    assert reshape_mask(torch.ones(1, 6, 8), (1, 24, 37, 3, 4), video_inpainting=True).min() == 1.0
    assert reshape_mask(torch.ones(4, 6, 8), (1, 4, 6, 8)).max() == 1.0
    a = z(100)
    a[50:60] = 1.0
    out = reshape_mask(a, (1, 32, 2, 40))
    assert tuple(out.shape) == (1, 32, 2, 40) and (out == 0.0).float().mean() > 0.7
    assert out[0, 0, 0, 20] == 1.0 and out[0, 0, 1, 20] == 1.0
    out = reshape_mask(a.reshape(1, 1, 100, 1), (1, 32, 2, 40))
    assert out[0, 0, 0, 20] == 1.0 and out[0, 0, 1, 20] == 1.0
    m = z(8, 1, 6, 8)
    m[2, 0, 2, 3], m[6, 0, 4, 5] = 1.0, 1.0
    out = reshape_mask(m, (1, 24, 2, 6, 8), video_inpainting=True)
    assert out[0, 0, 0, 2, 3] == 1.0 and out[0, 0, 1, 4, 5] == 1.0


def test_reshape_mask_index_math_equals_torch_gpu_interpolate_exhaustively():
    """lp_reshape_mask vs torch's own GPU nearest-exact (1-D / 2-D / 3-D kernels) over every (in, out) pair of a
    grid that includes up-sampling and the pairs where torch's CPU kernels deviate from ATen's formula: bit-exact."""
    import torch
    from lanpaint_amd.nodes import reshape_mask
    for n_in in list(range(1, 16)) + [54, 124]:
        ramp = torch.arange(n_in, dtype=torch.float32, device=DEV)
        for n_out in range(1, 260, 3 if n_in > 4 else 1):
            got = reshape_mask(ramp.reshape(1, n_in), (1, 1, 1, n_out)).reshape(-1)
            want = torch.nn.functional.interpolate(ramp.reshape(1, 1, 1, n_in), size=(1, n_out), mode="nearest-exact").reshape(-1)
            assert torch.equal(got, want), (n_in, n_out)
    got = reshape_mask(torch.arange(2, dtype=torch.float32, device=DEV).reshape(1, 1, 2, 1, 1), (1, 1, 47, 1, 1), video_inpainting=False)
    want = torch.nn.functional.interpolate(torch.arange(2, dtype=torch.float32, device=DEV).reshape(1, 1, 2, 1, 1), size=(47, 1, 1),
                                           mode="nearest-exact")
    assert torch.equal(got.reshape(-1), want.reshape(-1))


def test_reshape_mask_equals_torch_interpolate_pipeline():
    """Property check vs the exact torch ops the reference calls, on random sizes (all below the smallest
    output size, 41, at which torch's CPU kernels deviate from ATen's formula)."""
    import torch
    from lanpaint_amd.nodes import reshape_mask
    rng = np.random.default_rng(5)
    for _ in range(12):
        f, h, w = (int(v) for v in rng.integers(1, 40, 3))
        tf, th, tw = (int(v) for v in rng.integers(1, 24, 3))
        m = torch.from_numpy((rng.random((f, h, w)) > 0.6).astype(np.float32))
        got = reshape_mask(m, (2, 3, tf, th, tw), video_inpainting=True).cpu()
        t = torch.nn.functional.interpolate(m[None, None], size=(tf, th, tw), mode="nearest-exact")
        t = torch.nn.functional.max_pool3d(t, kernel_size=(5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0))
        assert torch.equal(got, t.repeat(2, 3, 1, 1, 1)), (f, h, w, tf, th, tw)
        got2 = reshape_mask(m, (1, 2, th, tw)).cpu()                     # 3-D mask = batch of images
        t2 = torch.nn.functional.interpolate(m[:, None], size=(th, tw), mode="nearest-exact").repeat(1, 2, 1, 1)[:1]
        assert torch.equal(got2, t2)


# ---- KSamplerX0Inpaint: the sampler-facing callable ----------------------------------------------
@pytest.mark.parametrize("flow", [False, True])
def test_ksampler_x0_inpaint_matches_oracle(flow):
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes
    shape = (1, 4, 8, 8)
    sig = gc.flow_sigmas(5) if flow else gc.karras_sigmas(5, 0.1, 10.0)
    rng = np.random.default_rng(9)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    denoise_mask = (rng.random(shape) > 0.5).astype(np.float32) * 0.9      # non-binary input, thresholded at 0.5
    latent_mask = orc.binarize_and_invert(denoise_mask)
    x0 = (sig[0] * noise + (1 - sig[0]) * y) if flow else (y + noise * sig[0])
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(64)]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731

    class M(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            self.calls += 1
            return 0.9 * x, 0.8 * x

    model = M(_FlowSampling() if flow else _DummySampling())
    model.model_type = nodes.ModelType.FLOW if flow else "EPS"
    it = iter([tt(d) for d in draws])
    k = nodes.KSamplerX0Inpaint(model, tt(sig))
    k.latent_image, k.noise = tt(y), tt(noise)
    k.PaintMethod = LanPaint(model, 4, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, MinStepFrac=0.5, rng=lambda like: next(it))
    k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 0.5

    it_o = iter(draws)
    omodel = M()
    o = orc.OracleLanPaint(omodel, 4, 15.0, 5.0, 1.0, 0.2, is_flow=flow, min_step_frac=0.5, randn=lambda like: next(it_o))
    x, xo = tt(x0), x0.astype(np.float32).copy()
    n_effs = []
    for i in range(len(sig) - 1):
        s = np.float32([sig[i]])
        den = k(x, tt(s), tt(denoise_mask), model_options={}, seed=0)
        times = orc.times_from_sigma(s, flow)
        n_eff = orc.effective_inner_steps(4, sig, float(s[0]), float(times[1].mean()), 1, 0.5)
        n_effs.append(n_eff)
        assert k.PaintMethod.last_inner_steps == n_eff
        den_o = o(xo, y, noise, s, latent_mask, times, None, 0, n_steps=n_eff)
        assert_close(den.cpu().numpy(), den_o, f"sigma[{i}] denoised", rel=5e-5)
        r = float((sig[i + 1] - sig[i]) / sig[i])
        x = x + (x - den) * r
        xo = (xo + (xo - den_o) * np.float32(r)).astype(np.float32)
    assert n_effs[-1] == 0 and max(n_effs) == 4          # EarlyStop=1 tail and the un-ramped head both exercised
    assert_close(x.cpu().numpy(), xo, "final x", rel=5e-5)
    assert model.calls == omodel.calls
    # no-mask path: plain model call (nodes.py:302)
    out = k(x, tt(np.float32([sig[0]])), None, model_options={}, seed=0)
    assert torch.equal(out, 0.9 * x)


@pytest.mark.parametrize("name", sorted(gc.NODE_SCHEDULES))
def test_ksampler_x0_inpaint_matches_the_reference_sampler_callable(name):
    """a1 against a reference RUN (no oracle in between): tests/golden/node_*.npz holds what the reference's own
    KSamplerX0Inpaint.__call__ (nodes.py:229-315) + engine did over a 12-sigma schedule with the node defaults (MinStepFrac
    1.0, EarlyStop 1): the inner-step count per sigma, every denoised, the final x.  The HIP callable, fed the same xi stream,
    must choose the same counts (lp_sigma_times + the host rule) and reproduce the trajectory."""
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes
    from tests.helpers import load_golden
    sc = gc.build_node_schedule(name)
    g = load_golden(name)
    h, flow, sig = sc["hyper"], sc["flow"], sc["sigmas"]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    it = iter([tt(d) for d in gc.seeded_xi(int(g["xi_seed"]), sc["shape"], int(g["n_draws"]))])

    class M(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            self.calls += 1
            return 0.9 * x, 0.8 * x

    model = M(_FlowSampling() if flow else _DummySampling())
    model.model_type = nodes.ModelType.FLOW if flow else "EPS"
    k = nodes.KSamplerX0Inpaint(model, tt(sig))
    k.latent_image, k.noise = tt(sc["y"]), tt(sc["noise"])
    k.PaintMethod = LanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=flow,
                             MinStepFrac=h["MinStepFrac"], rng=lambda like: next(it))
    k.LanPaint_early_stop, k.LanPaint_min_step_frac = h["EarlyStop"], h["MinStepFrac"]
    x, dm, mo = tt(sc["x"]), tt(sc["denoise_mask"]), {}
    n_eff = []
    for i in range(len(sig) - 1):
        s = torch.full((sc["shape"][0],), float(sig[i]), dtype=torch.float32, device=DEV)
        calls = model.calls
        den = k(x, s, dm, model_options=mo, seed=0)
        n_eff.append(model.calls - calls - 1)
        assert k.PaintMethod.last_inner_steps == n_eff[-1]
        assert_close(den.cpu().numpy(), g["denoised"][i], f"{name}: denoised[{i}]", rel=5e-5)
        x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
    assert n_eff == list(g["n_eff"])
    assert sum(1 for _ in it) == 0 and model.calls == int(g["model_calls"])
    assert_close(x.cpu().numpy(), g["x_final"], f"{name}: final x", rel=5e-5)


@pytest.mark.parametrize("flow,inference,shape", [(False, False, (2, 4, 16, 16)), (True, False, (2, 4, 16, 16)), (False, True, (2, 4, 16, 16)),
                                                  (True, False, (2, 16, 3, 8, 10))],       # 5-D video latent (BASELINE configs[4] in small; two rows: a per-row sigma takes the fused flow-form replace step, lanpaint.py:89-92)
                         ids=["ve", "flow", "ve_inference_mode", "flow_video5d"])
def test_ksampler_x0_inpaint_split_phase_graph_path_equals_eager(flow, inference, shape, monkeypatch):
    """(`inference`: the whole run inside torch.inference_mode(), as ComfyUI executes its nodes -- inference tensors
    do not track `_version`, which the per-tensor caches of the engine and the sampler callable used to read.)
    The replayed node path (replace step enqueued before the host knows n_eff, mailbox read, graph picked
    afterwards: engine.begin_call / finish_call) against eager launches: same n_eff sequence as the reference's rule
    (nodes.py:286-299) and bitwise equal trajectories under a torch seed, over repeated passes of a schedule whose
    n_eff ramps through every value."""
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes
    n_think = 5
    sig = gc.flow_sigmas(12) if flow else gc.karras_sigmas(12, 0.05, 12.0)
    rng = np.random.default_rng(21)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    denoise_mask = (rng.random(shape) > 0.4).astype(np.float32)
    if len(shape) == 5:          # the job's mask as a video workflow builds it: pixel-resolution frames through reshape_mask's video path
        pix = np.zeros((4 * (shape[2] - 1) + 1, shape[3] * 8, shape[4] * 8), np.float32)
        pix[pix.shape[0] // 2:, :, : pix.shape[2] // 2] = 1.0
        dm5 = nodes.reshape_mask(torch.from_numpy(pix).to(DEV), shape, video_inpainting=True)
        assert tuple(dm5.shape) == shape and 0 < float(dm5.mean()) < 1
        denoise_mask = dm5.cpu().numpy()
    x0 = (sig[0] * noise + (1 - sig[0]) * y) if flow else (y + noise * sig[0])
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731

    class M(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            self.calls += 1
            return 0.9 * x, 0.8 * x

    import contextlib
    monkeypatch.delenv("LANPAINT_AMD_SPECULATE", raising=False)      # (this test is about the default: guessing on)
    res = {}
    for graph in (False, True):
      with (torch.inference_mode() if inference else contextlib.nullcontext()):
        model = M(_FlowSampling() if flow else _DummySampling())
        model.model_type = nodes.ModelType.FLOW if flow else "EPS"
        k = nodes.KSamplerX0Inpaint(model, tt(sig))
        k.latent_image, k.noise = tt(y), tt(noise)
        if inference:
            assert k.noise.is_inference()
        k.PaintMethod = LanPaint(model, n_think, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, MinStepFrac=1.0, rng="torch", graph=graph)
        k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
        one_call = {"n": 0}
        node_call = k.PaintMethod.node_call

        def counted(*a, **kw):                       # the one-FFI-trip steady state (lp_node_call) must be what replays take
            r = node_call(*a, **kw)
            one_call["n"] += r is not None
            if r is not None:
                one_call["spec"] = one_call.get("spec", 0) + int(a[-1].speculated)
                one_call["hit"] = one_call.get("hit", 0) + int(a[-1].hit)
                one_call["one"] = one_call.get("one", 0) + int(a[-1].one_launch)
            return r
        k.PaintMethod.node_call = counted
        dm, mo = tt(denoise_mask), {}
        torch.manual_seed(77)
        outs, n_effs, split = [], [], 0
        for rep in range(3):                            # pass 0 captures, passes 1-2 replay through begin/finish
            x = tt(x0)
            for i in range(len(sig) - 1):
                s = torch.full((shape[0],), float(sig[i]), dtype=torch.float32, device=DEV)
                before = k.PaintMethod.iterations_run
                tok_cap = k.PaintMethod._last_cap
                den = k(x, s, dm, model_options=mo, seed=0)
                n_effs.append(k.PaintMethod.iterations_run - before)
                split += int(graph and tok_cap is not None and tok_cap.tail is not None)
                outs.append(den)
                x = torch.lerp(den, x, float(sig[i + 1] / sig[i]))
        torch.cuda.synchronize()
        res[graph] = ([o.cpu().numpy() for o in outs], x.cpu().numpy(), n_effs, model.calls, split,
                      torch.cuda.get_rng_state(DEV).clone(), one_call["n"], one_call.get("spec", 0), one_call.get("hit", 0),
                      one_call.get("one", 0))
    expect = []
    for i in range(len(sig) - 1):
        s = np.full((shape[0],), sig[i], dtype=np.float32)
        expect.append(orc.effective_inner_steps(n_think, sig, float(sig[i]), float(orc.times_from_sigma(s, flow)[1].mean()), 1, 1.0))
    assert res[False][2] == res[True][2] == expect * 3
    assert len(set(expect)) >= 5 and 0 in expect and n_think in expect   # the ramp exercises (nearly) every graph variant
    assert res[True][4] >= 2 * (len(sig) - 1)                            # the replays went through the split-phase path
    assert res[True][6] >= 2 * (len(sig) - 1) - 6 and res[False][6] == 0    # ... in one trip through the FFI each (lp_node_call)
    # ... most of them queued for a SPECULATED count before the device answered (the schedule is walked in order: hits), and
    # the wrap-around from the last sigma of a pass to the first of the next is a miss the device voided -- with no trace in
    # the results (bitwise equality below) or in the generator state
    assert res[True][7] >= len(sig) and res[True][8] >= res[True][7] - 3 and res[True][8] < res[True][7]
    # round 6: a speculated call is ONE hipGraphLaunch -- the replace launch with the sigma algebra folded in is node 0 of a copy
    # of the captured call (lp_graph_clone_sigma_root), its arguments refreshed per call; LANPAINT_AMD_NODE_ONE_LAUNCH=0 (the
    # round-5 form: eager replace launch in front of the tail graph) is covered by the parametrised run below
    want_one = os.environ.get("LANPAINT_AMD_NODE_ONE_LAUNCH", "1") != "0"
    assert (res[True][9] == res[True][7]) if want_one else (res[True][9] == 0), (res[True][9], res[True][7])
    assert res[True][3] < res[False][3]          # the Python backbone only ran while capturing
    for a, b in zip(res[False][0], res[True][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[False][1], res[True][1])
    assert torch.equal(res[False][5], res[True][5])                      # the generator ends where eager leaves it


def test_node_path_with_the_eager_replace_launch_in_front_of_the_tail_graph(monkeypatch):
    """The round-5 form of a speculated call (LANPAINT_AMD_NODE_ONE_LAUNCH=0) stays available and equal."""
    monkeypatch.setenv("LANPAINT_AMD_NODE_ONE_LAUNCH", "0")
    test_ksampler_x0_inpaint_split_phase_graph_path_equals_eager(False, False, (2, 4, 16, 16), monkeypatch)


# ---- argument forms the reference accepts through plain torch broadcasting -------------------------------
def test_broadcastable_mask_scalar_sigma_and_half_latents():
    import torch
    from lanpaint_amd import LanPaint
    from tests.helpers import load_golden, xi_list
    case = gc.build_case("ve_basic")
    g = load_golden("ve_basic")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731

    def run(x, mask, sigma, times, y=None, noise=None):
        it = iter([tt(d) for d in xi_list(g)])
        eng = LanPaint(_DummyLinear(), 5, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
        out = eng(x, tt(case["y"]) if y is None else y, tt(case["noise"]) if noise is None else noise, sigma, mask, times, None, 0)
        return x, out

    # (1) mask given as [1, 1, H, W] (broadcast over channels), sigma / times as 0-d tensors
    m11 = tt(case["mask"][:, :1])
    s0 = torch.tensor(float(case["sigma"][0]), device=DEV)
    t0 = tuple(torch.tensor(float(t[0]), device=DEV) for t in case["times"])
    x, out = run(tt(case["x"].copy()), m11, s0, t0)
    assert_close(x.cpu().numpy(), g["x_out"], "broadcast mask + 0-d sigma: x")
    assert_close(out.cpu().numpy(), g["out"], "broadcast mask + 0-d sigma: out")
    # (2) times given already broadcast as [B, 1, 1, 1]
    t4 = tuple(tt(t).reshape(1, 1, 1, 1) for t in case["times"])
    x, out = run(tt(case["x"].copy()), tt(case["mask"]), tt(case["sigma"]), t4)
    assert_close(out.cpu().numpy(), g["out"], "[B,1,1,1] times: out")
    # (3) fp16 sampler latent: computed in fp32, written back / returned in fp16
    xh = tt(case["x"].copy()).half()
    x, out = run(xh, tt(case["mask"]), tt(case["sigma"]), tuple(tt(t) for t in case["times"]))
    assert x.dtype == torch.float16 and out.dtype == torch.float16 and x.data_ptr() == xh.data_ptr()
    assert float((out.float().cpu() - torch.from_numpy(g["out"])).abs().max()) < 0.06
    # (4) known latent / noise given as double precision
    x, out = run(tt(case["x"].copy()), tt(case["mask"]), tt(case["sigma"]), tuple(tt(t) for t in case["times"]),
                 y=tt(case["y"]).double(), noise=tt(case["noise"]).double())
    assert_close(out.cpu().numpy(), g["out"], "float64 y / noise: out")


class _DummyLinear(_DummyModel):
    def __call__(self, x, sigma, model_options=None, seed=None):
        self.calls += 1
        return 0.9 * x, 0.8 * x


def test_rows_whose_step_size_is_zero_are_identity_updates_and_do_not_disturb_the_other_rows():
    """Documented deviation: the reference tests `mean(dtx) <= 0` over the whole batch
    (lanpaint.py:205) -- all rows skip or none does, and a row with step 0 inside a batch that goes on divides by
    zero (A = 1 / (1 - abt) with abt = 1).  Here the test is per row (LP_C_VALID): the row with sigma = 0 passes
    through the loop unchanged, finite, and the other row gets exactly what it gets when run alone."""
    import torch
    from lanpaint_amd import LanPaint
    shape = (2, 4, 8, 8)
    rng = np.random.default_rng(17)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    mask = gc.box_mask(shape)
    sig = np.float32([0.0, 1.3])
    x = (y + noise * sig.reshape(2, 1, 1, 1)).astype(np.float32)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731

    def run(sig_rows):
        it = iter([tt(d) for d in xi])
        eng = LanPaint(MODELS["linear_tuple"](), 4, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
        s = tt(sig_rows)
        xx = tt(x)
        out = eng(xx, tt(y), tt(noise), s, tt(mask), gc.times_from_sigma(s, False), None, 0)
        torch.cuda.synchronize()
        return xx.cpu().numpy(), out.cpu().numpy(), eng

    xi = [rng.standard_normal(shape, dtype=np.float32) for _ in range(7)]
    xb, ob, eng = run(sig)
    x1, o1, _ = run(np.float32([1.3, 1.3]))      # (two rows again: per-row sigma takes the reference's flow-form replace step)
    assert eng.iterations_run == 4                                      # the backbone is still called (per-row rule)
    assert np.isfinite(xb).all() and np.isfinite(ob).all()
    np.testing.assert_array_equal(xb[1], x1[1])                         # the live row: untouched by its dead neighbour
    np.testing.assert_array_equal(ob[1], o1[1])
    # the dead row: replace step (sigma = 0 -> known region = y, inpaint region = x) and the x_t <-> x round trip only
    m = mask[0] == 1
    np.testing.assert_allclose(xb[0][m], y[0][m], rtol=0, atol=1e-6)
    np.testing.assert_allclose(xb[0][~m], x[0][~m], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("bits", [False, True])
def test_nan_in_the_head_an_element_does_not_read_stays_out_of_it(bits):
    """Documented deviation: the reference blends the two score branches with the mask as a
    weight (lanpaint.py:182-184), so a NaN in the head an element does NOT use still reaches it (NaN * 0 = NaN) --
    head 0 at a known element, head 1 at an inpaint element.  With a hard mask the kernels SELECT the branch: such a
    NaN never enters, and the run equals the one whose backbone output is clean -- pinned here for the fp32 and the
    bit-packed mask.  (A NaN in the head the element does use propagates in both, of course.)"""
    import torch
    from lanpaint_amd import LanPaint, pack_mask
    shape = (1, 4, 8, 8)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    mask = gc.box_mask(shape)
    sig = np.float32([1.1])
    x = (y + noise * sig).astype(np.float32)
    xi = [rng.standard_normal(shape, dtype=np.float32) for _ in range(5)]
    known = torch.from_numpy(mask == 1).to(DEV)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731

    class Poisoned(MODELS["linear_tuple"]):
        poison = False

        def __call__(self, xin, t, model_options=None, seed=None):
            h0, h1 = super().__call__(xin, t, model_options=model_options, seed=seed)
            if self.poison and self.calls <= 3:      # the think iterations; the final call's head 0 is blended with
                #                                      the mask as a weight by reference and kernel alike
                h0 = torch.where(known, torch.full_like(h0, float("nan")), h0)       # head 0 is unused where known
                h1 = torch.where(~known, torch.full_like(h1, float("nan")), h1)      # head 1 is unused where inpainted
            return h0, h1

    def run(poison):
        it = iter([tt(d) for d in xi])
        model = Poisoned()
        model.poison = poison
        eng = LanPaint(model, 3, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
        m = tt(mask)
        xx = tt(x)
        s = tt(sig)
        out = eng(xx, tt(y), tt(noise), s, pack_mask(m) if bits else m, gc.times_from_sigma(s, False), None, 0)
        torch.cuda.synchronize()
        return xx.cpu().numpy(), out.cpu().numpy()

    xc, oc = run(False)
    xp, op = run(True)
    assert np.isfinite(xp).all()
    np.testing.assert_array_equal(xp, xc)
    # `out` is the final head-0 prediction re-projected: known elements take y, inpaint elements the clean part of head 0
    np.testing.assert_array_equal(op, oc)


@pytest.mark.parametrize("rng,shape", [("torch", (1, 4, 16, 16)), ("philox", (1, 4, 16, 16)), ("torch", (1, 4, 368, 368))],
                         ids=["torch", "philox", "torch_16B_lanes"])
def test_speculated_inner_step_counts_that_miss_leave_no_trace(rng, shape):
    """lp_node_call queues a sigma call for a GUESSED inner-step count -- the sigma algebra riding in the replace launch
    (LP_PH_SIGMA; one element and four per lane) -- the device checks the guess and voids the run on a
    miss, then the call is queued again.  A Heun-like order (every sigma evaluated twice) makes half the guesses wrong until
    guessing turns itself off: the results must equal a run that never guesses (LANPAINT_AMD_SPECULATE=0, read when the callable is built) bit for bit -- x,
    every denoised, the torch generator (rng="torch") and the replayed Philox counter (rng="philox": same noise in both runs)."""
    import os
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes
    n_think = 5
    sig = gc.karras_sigmas(10, 0.05, 12.0)
    rs = np.random.default_rng(4)
    y = rs.standard_normal(shape, dtype=np.float32)
    noise = rs.standard_normal(shape, dtype=np.float32)
    denoise_mask = (rs.random(shape) > 0.4).astype(np.float32)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    order = [0, 1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9, 0, 1, 2, 3, 4, 5, 6, 7, 8]       # doubled, then in order, then a wrap

    class M(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            self.calls += 1
            return 0.9 * x, 0.8 * x

    res = {}
    for spec in ("0", "1"):
        os.environ["LANPAINT_AMD_SPECULATE"] = spec
        try:
            from tests.stubs import VESampling
            model = M(VESampling())               # (declares its noise_scaling form: the replace step is fusable, the call replayable)
            model.model_type = "EPS"
            k = nodes.KSamplerX0Inpaint(model, tt(sig))
            k.latent_image, k.noise = tt(y), tt(noise)
            k.PaintMethod = LanPaint(model, n_think, 15.0, 5.0, 1.0, 0.2, MinStepFrac=1.0, rng=rng, philox_seed=11, graph=True)
            k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
            stats = {"spec": 0, "hit": 0}
            node_call = k.PaintMethod.node_call

            def counted(*a, **kw):
                r = node_call(*a, **kw)
                if r is not None:
                    stats["spec"] += int(a[-1].speculated)
                    stats["hit"] += int(a[-1].hit)
                return r
            k.PaintMethod.node_call = counted
            dm, mo = tt(denoise_mask), {}
            torch.manual_seed(5)
            outs = []
            for rep in range(2):                 # pass 0 captures every count, pass 1 replays (and guesses)
                x = tt(y + noise * sig[0])
                for j in order:
                    s = torch.full((1,), float(sig[j]), dtype=torch.float32, device=DEV)
                    den = k(x, s, dm, model_options=mo, seed=0)
                    outs.append(den.clone())
                    x = torch.lerp(den, x, 0.7)
            torch.cuda.synchronize()
            res[spec] = ([o.cpu().numpy() for o in outs], x.cpu().numpy(), torch.cuda.get_rng_state(DEV).clone(),
                         k.PaintMethod.iterations_run, dict(stats))
        finally:
            os.environ.pop("LANPAINT_AMD_SPECULATE", None)
    assert res["0"][4]["spec"] == 0
    assert res["1"][4]["spec"] >= 4 and res["1"][4]["hit"] < res["1"][4]["spec"]        # it guessed, and some guesses were wrong
    assert res["0"][3] == res["1"][3]
    for a, b in zip(res["0"][0] + [res["0"][1]], res["1"][0] + [res["1"][1]]):
        np.testing.assert_array_equal(a, b)
    assert torch.equal(res["0"][2], res["1"][2])


# ---------------------------------------------------------------- caches vs tensors rewritten in place (round 4)
def _oracle_schedule(sig, x0, y, noise, masks_by_call, n_think, draws):
    """The oracle over a VE schedule whose latent mask may change from call to call (masks_by_call[i])."""
    it = iter(draws)
    o = orc.OracleLanPaint(MODELS["linear_tuple"](), n_think, 15.0, 5.0, 1.0, 0.2, randn=lambda like: next(it))
    x, dens = x0.copy(), []
    for i in range(len(sig)):
        s = np.float32([sig[i]] * x0.shape[0])
        den = o(x, y, noise, s, masks_by_call[i], orc.times_from_sigma(s, False), None, 0)
        dens.append(den)
        if i + 1 < len(sig):
            x = (x + (x - den) / sig[i] * (sig[i + 1] - sig[i])).astype(np.float32)
    return x, dens


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("inference", [False, True])
def test_sampler_callable_follows_a_denoise_mask_rewritten_in_place(inference, graph):
    """The reference recomputes its latent mask from `denoise_mask` on EVERY call (nodes.py:277-283), so a mask tensor that is
    rewritten in place between two sigma calls takes effect at once.  KSamplerX0Inpaint keeps a packed copy per mask tensor:
    it has to notice -- through the tensor's version counter, or, under torch.inference_mode() (how ComfyUI runs its nodes:
    no counter), by re-deriving the copy on every call into the same buffers, so that eager launches AND replayed graphs
    read the new mask.  Expected values: the oracle with the mask switched at the same call."""
    import contextlib
    import torch
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes
    shape, n_think, n_sig, switch = (1, 4, 16, 16), 3, 6, 3
    sig = gc.karras_sigmas(n_sig, 0.1, 10.0)[:-1]
    rng = np.random.default_rng(77)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    dm_a = np.zeros(shape, dtype=np.float32)
    dm_a[..., 8:] = 1.0                                   # inpaint the right half ...
    dm_b = np.zeros(shape, dtype=np.float32)
    dm_b[..., :6, :] = 1.0                                # ... then (rewritten in place) the top rows
    x0 = (y + noise * sig[0]).astype(np.float32)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    draws = gc.seeded_xi(4301, shape, n_sig * (2 * n_think - 1))
    masks = [(1.0 - dm_a) if i < switch else (1.0 - dm_b) for i in range(n_sig)]
    x_want, den_want = _oracle_schedule(sig, x0, y, noise, masks, n_think, draws)

    class M(_DummyModel):
        def __call__(self, x, sigma, model_options=None, seed=None):
            return 0.9 * x, 0.8 * x

    with (torch.inference_mode() if inference else contextlib.nullcontext()):
        it = iter([tt(d) for d in draws])
        model = M(_DummySampling())
        model.model_type = "EPS"
        k = nodes.KSamplerX0Inpaint(model, tt(np.concatenate([sig, [0.0]])))
        k.latent_image, k.noise = tt(y), tt(noise)
        # (recorded draws keep an engine eager; the replayed variant draws the same values from torch's generator instead)
        k.PaintMethod = LanPaint(model, n_think, 15.0, 5.0, 1.0, 0.2, MinStepFrac=0.0, graph=graph,
                                 rng=("torch" if graph else (lambda like: next(it))))
        k.LanPaint_early_stop, k.LanPaint_min_step_frac = 0, 0.0
        dm, x, mo = tt(dm_a), tt(x0), {}
        if graph:      # torch's own stream: re-derive the expectation from what torch.randn returns for this seed
            torch.manual_seed(991)
            draws_t = [torch.randn(shape, device=DEV).cpu().numpy() for _ in range(len(draws))]
            x_want, den_want = _oracle_schedule(sig, x0, y, noise, masks, n_think, draws_t)
            torch.manual_seed(991)
        for i in range(n_sig):
            if i == switch:
                dm.copy_(tt(dm_b))                        # IN PLACE: same tensor object, same address
            s = torch.full((1,), float(sig[i]), dtype=torch.float32, device=DEV)
            den = k(x, s, dm, model_options=mo, seed=0)
            assert_close(den.cpu().numpy(), den_want[i], f"denoised[{i}]", rel=5e-5)
            if i + 1 < n_sig:
                x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
        assert_close(x.cpu().numpy(), x_want, "final x", rel=5e-5)
        if graph:
            assert len(k.PaintMethod._graphs) >= 1        # the rewritten mask did not cost the captures


@pytest.mark.parametrize("inference", [False, True])
def test_engine_repacks_a_packed_mask_that_was_rewritten_in_place(inference):
    """`pack_mask(latent_mask)` attaches a bit-packed copy to the caller's mask; rewriting that mask in place afterwards used
    to leave the copy stale.  The engine now re-packs it in place (version counter moved / no counter under inference mode)."""
    import contextlib
    import torch
    from lanpaint_amd import LanPaint, pack_mask
    shape, n_think, n_sig, switch = (2, 4, 16, 16), 2, 4, 2
    sig = gc.karras_sigmas(n_sig, 0.2, 8.0)[:-1]
    rng = np.random.default_rng(78)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    m_a, m_b = gc.box_mask(shape), (rng.random(shape) > 0.5).astype(np.float32)
    x0 = (y + noise * sig[0]).astype(np.float32)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    draws = gc.seeded_xi(4302, shape, n_sig * (2 * n_think - 1))
    masks = [m_a if i < switch else m_b for i in range(n_sig)]
    x_want, den_want = _oracle_schedule(sig, x0, y, noise, masks, n_think, draws)
    with (torch.inference_mode() if inference else contextlib.nullcontext()):
        it = iter([tt(d) for d in draws])
        eng = LanPaint(MODELS["linear_tuple"](), n_think, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
        mask = pack_mask(tt(m_a))
        bits_ptr = mask._lp_bits.data_ptr()
        x, yg, ng = tt(x0), tt(y), tt(noise)
        for i in range(n_sig):
            if i == switch:
                mask.copy_(tt(m_b))
            s = torch.full((shape[0],), float(sig[i]), dtype=torch.float32, device=DEV)
            den = eng(x, yg, ng, s, mask, gc.times_from_sigma(s, False), None, 0)
            assert_close(den.cpu().numpy(), den_want[i], f"denoised[{i}]", rel=5e-5)
            if i + 1 < n_sig:
                x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
        assert_close(x.cpu().numpy(), x_want, "final x", rel=5e-5)
        assert mask._lp_bits.data_ptr() == bits_ptr       # re-packed IN PLACE
        assert np.array_equal(orc.unpack_mask_bits(mask._lp_bits.cpu().numpy(), mask.numel()).reshape(shape), m_b)


def test_early_stop_ring_follows_the_mask_in_place_under_inference_mode():
    """The mask-edge ring of the inner early stop (earlystop.py:32-49) is cached per mask tensor; a tensor without a version
    counter gets it recomputed on every call, into the same buffers (captured early-stop launches bake their addresses)."""
    import torch
    from lanpaint_amd.lanpaint import _DeviceStop
    shape = (1, 2, 12, 20)
    rng = np.random.default_rng(5)
    m_a, m_b = (rng.random(shape) > 0.5).astype(np.float32), (rng.random(shape) > 0.3).astype(np.float32)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    with torch.inference_mode():
        mask = tt(m_a)
        ds = _DeviceStop(mask, 4)
        r1 = ds.ring_for(mask, mask)
        p_ring, p_bits = r1.data_ptr(), ds.ring_bits().data_ptr()
        assert np.array_equal(r1.cpu().numpy(), orc.boundary_weight(m_a, (1 - m_a).astype(np.float32)))
        mask.copy_(tt(m_b))
        r2 = ds.ring_for(mask, mask)
        want = orc.boundary_weight(m_b, (1 - m_b).astype(np.float32))
        assert r2.data_ptr() == p_ring and ds.ring_bits().data_ptr() == p_bits
        assert np.array_equal(r2.cpu().numpy(), want)
        assert np.array_equal(orc.unpack_mask_bits(ds.ring_bits().cpu().numpy(), mask.numel()).reshape(shape), (want > 0.5).astype(np.float32))


def test_noise_verdict_is_rechecked_for_tensors_without_a_version_counter():
    """lanpaint.py:51-52: noise that is all zeros is regenerated.  The verdict is cached per tensor version; an inference tensor
    has none, so the engine re-reads it on every call unless the caller vouches for the run's noise (assume_static_noise,
    which KSAMPLER.sample sets on the per-run engine it builds)."""
    import torch
    from lanpaint_amd import LanPaint
    with torch.inference_mode():
        eng = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox")
        noise = torch.randn(1, 4, 8, 8, device=DEV)
        assert eng._noise_is_zero(noise) is False
        noise.zero_()
        assert eng._noise_is_zero(noise) is True           # no counter: looked at again
        eng.assume_static_noise = True
        noise.normal_()
        assert eng._noise_is_zero(noise) is True           # vouched for: the cached verdict stands
    t = torch.randn(1, 4, 8, 8, device=DEV)
    eng2 = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox")
    assert eng2._noise_is_zero(t) is False
    t.zero_()                                              # version counter moves: noticed without any flag
    assert eng2._noise_is_zero(t) is True


def test_engine_packs_a_binary_fp32_mask_by_itself_on_second_sight(monkeypatch):
    """The reference's interface hands the engine a plain fp32 mask.  When the same mask tensor comes back on the next call
    and it is binary, the engine packs it (one host read per mask tensor) and the hard-mask kernels run from then on --
    bitwise the same results; a soft mask is left alone; a packed mask that is rewritten to soft values is unpacked again."""
    import torch
    from lanpaint_amd import LanPaint
    monkeypatch.delenv("LANPAINT_AMD_AUTO_PACK", raising=False)
    case = gc.build_case("ve_sdxl_shape")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    args = (tt(case["y"]), tt(case["noise"]), tt(case["sigma"]))
    times = tuple(tt(t) for t in case["times"])
    res = {}
    for auto in (False, True):
        eng = LanPaint(MODELS["linear_tuple"](), 3, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=5, graph=False)
        eng.auto_pack_mask = auto
        mask, x, outs = tt(case["mask"]), tt(case["x"]), []
        for k in range(3):
            outs.append(eng(x, *args, mask, times, None, 0).clone())
            assert (getattr(mask, "_lp_bits", None) is not None) == (auto and k >= 1)
        res[auto] = (x.clone(), outs)
    assert torch.equal(res[True][0], res[False][0]) and all(torch.equal(a, b) for a, b in zip(res[True][1], res[False][1]))
    # a soft mask is never packed
    eng = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=5, graph=False)
    soft, x = tt(np.random.default_rng(3).random(case["mask"].shape, dtype=np.float32)), tt(case["x"])
    for _ in range(3):
        eng(x, *args, soft, times, None, 0)
    assert getattr(soft, "_lp_bits", None) is None and eng._mask_seen[2] is True
    # a mask the engine packed, then rewritten in place to soft values: the copy is dropped, results follow the soft mask
    eng = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=5, graph=False)
    mask, x = tt(case["mask"]), tt(case["x"])
    eng(x, *args, mask, times, None, 0)
    eng(x, *args, mask, times, None, 0)
    assert getattr(mask, "_lp_bits", None) is not None
    mask.copy_(soft)
    x1 = tt(case["x"])
    out1 = eng(x1, *args, mask, times, None, 0)
    assert getattr(mask, "_lp_bits", None) is None
    ref = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=5, graph=False)
    ref.auto_pack_mask = False
    ref._philox_offset = eng._philox_offset - 2          # same launch sequence numbers as the call above
    x2 = tt(case["x"])
    out2 = ref(x2, *args, soft.clone(), times, None, 0)
    assert torch.equal(out1, out2) and torch.equal(x1, x2)


# ---- mask index math against the reference's arbiter, torch on the CPU -------------------------------
RULES = {0: "scalar", 1: "generic_fma", 2: "generic"}


def test_reshape_mask_kernel_equals_its_python_twin_for_every_index_rule(hip_lib):
    """lp_reshape_mask's three source-index forms (include/lanpaint_hip.h LP_NN_ATEN_*) against the oracle's numpy twins
    (which the CPU suite pins to torch-CPU over every (in, out) <= 512), through the C ABI: down- and up-sampling pairs,
    every tie pair where the forms differ, on each of the three axes."""
    import torch
    from lanpaint_amd import _cabi
    from oracle import lanpaint_oracle as orc
    pairs = [(a, b) for a in (1, 2, 3, 4, 6, 7, 14, 54, 124, 259) for b in (1, 2, 5, 37, 41, 47, 83, 97, 123, 141, 201, 260, 519)]
    pairs += [(a, b) for a in range(1, 260) for b in range(a + 1, 520)
              if not np.array_equal(orc.nearest_exact_src_index(b, a, "scalar"), orc.nearest_exact_src_index(b, a, "generic_fma"))][:120]
    st = torch.cuda.current_stream().cuda_stream
    for rule, name in RULES.items():
        for axis in range(3):
            for a, b in pairs:
                src_shape, dst_shape = [1, 1, 1], [1, 1, 1]
                src_shape[axis], dst_shape[axis] = a, b
                src = torch.arange(a, dtype=torch.float32, device=DEV).reshape(1, 1, *src_shape).contiguous()
                dst = torch.empty((1, 1, *dst_shape), dtype=torch.float32, device=DEV)
                _cabi.check(hip_lib.lp_reshape_mask(src.data_ptr(), 1, 1, *src_shape, dst.data_ptr(), 1, 1, *dst_shape, 1,
                                                    rule << _cabi.LP_RESHAPE_RULE_SHIFT, st))
                want = orc.nearest_exact_src_index(b, a, name)
                assert np.array_equal(dst.reshape(-1).cpu().numpy().astype(np.int64), want), (name, axis, a, b)
    # an unknown rule is refused, not silently mapped
    src, dst = torch.zeros(4, device=DEV), torch.zeros(4, device=DEV)
    assert hip_lib.lp_reshape_mask(src.data_ptr(), 1, 1, 1, 1, 4, dst.data_ptr(), 1, 1, 1, 1, 4, 1, 3 << 8, st) == _cabi.LP_E_INVALID


def test_reshape_mask_of_a_host_mask_equals_torch_cpu_interpolate_when_upsampling():
    """The reference resamples on the mask's own device BEFORE `.to(device)` (nodes.py:159-160): for ComfyUI's host tensors
    that is torch's CPU kernels, whose index rule differs from the GPU kernels' on up-sampling ties.  lanpaint_amd.reshape_mask
    of a HOST mask must equal the reference's torch pipeline run on the host, bit for bit -- audio [F] -> tokens (1-D call),
    [1, 1, F, 1] (2-D call), low-resolution image masks (2-D, both dispatch regimes), a low-resolution video mask (3-D) --
    and of a DEVICE mask torch's GPU kernels."""
    import torch
    from lanpaint_amd.nodes import reshape_mask
    interp = torch.nn.functional.interpolate
    g = torch.Generator().manual_seed(3)
    for f, t in [(2, 41), (2, 47), (4, 82), (6, 123), (54, 259), (14, 201), (100, 40), (7, 300), (2, 141)]:
        a = torch.rand(f, generator=g)
        want = interp(a[None, None], size=(t,), mode="nearest-exact").expand(1, 1, 3, t)
        assert torch.equal(reshape_mask(a, (1, 2, 3, t)).cpu(), want.repeat(1, 2, 1, 1)), ("[F]", f, t)
        a4 = a.reshape(1, 1, f, 1)
        want = interp(a4, size=(t, 1), mode="nearest-exact").permute(0, 1, 3, 2).expand(1, 1, 3, t)
        assert torch.equal(reshape_mask(a4, (1, 2, 3, t)).cpu(), want.repeat(1, 2, 1, 1)), ("[1,1,F,1]", f, t)
        # the same masks held on the device: torch's GPU kernels are the reference there
        want = interp(a.to(DEV)[None, None], size=(t,), mode="nearest-exact").expand(1, 1, 3, t)
        assert torch.equal(reshape_mask(a.to(DEV), (1, 1, 3, t)), want), ("[F] on the device", f, t)
    for (h, w), (th, tw) in [((2, 2), (41, 47)), ((2, 6), (141, 123)), ((4, 2), (82, 83)), ((54, 14), (259, 201)), ((2, 2), (64, 64)),
                             ((2, 2), (128, 128)), ((6, 4), (123, 94))]:
        m = torch.rand(h, w, generator=g)
        want = interp(m[None, None], size=(th, tw), mode="nearest-exact")
        assert torch.equal(reshape_mask(m, (1, 1, th, tw)).cpu(), want), ((h, w), (th, tw))
        want = interp(m.to(DEV)[None, None], size=(th, tw), mode="nearest-exact")
        assert torch.equal(reshape_mask(m.to(DEV), (1, 1, th, tw)), want), ("device", (h, w), (th, tw))
    for (f, h, w), (tf, th, tw) in [((2, 2, 2), (41, 47, 83)), ((4, 6, 2), (82, 123, 41)), ((3, 5, 7), (21, 60, 104))]:
        m = torch.rand(f, h, w, generator=g)
        want = interp(m[None, None], size=(tf, th, tw), mode="nearest-exact")
        want = torch.nn.functional.max_pool3d(want, kernel_size=(5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0))
        assert torch.equal(reshape_mask(m, (1, 1, tf, th, tw), video_inpainting=True).cpu(), want), ((f, h, w), (tf, th, tw))


def test_merge_video_with_a_low_resolution_host_mask_follows_torch_cpu_index_rule():
    """nodes.py:1078-1081 resamples a lower-resolution mask on the mask's device before the blend: with k = 1 (no dilation, no
    blur) the merge is a per-pixel select, so every pixel shows which mask element it read."""
    import torch
    from lanpaint_amd import blend
    g = torch.Generator().manual_seed(5)
    for (h, w), (H, W) in [((2, 2), (41, 141)), ((4, 6), (82, 123)), ((2, 2), (47, 47))]:
        mask = (torch.rand(1, h, w, generator=g) > 0.5).float()
        orig, inp = torch.zeros(1, H, W, 3), torch.ones(1, H, W, 3)
        want = torch.nn.functional.interpolate(mask.unsqueeze(1), size=(H, W), mode="nearest-exact")[:, 0]
        got = blend.merge_video_with_mask(orig.to(DEV), inp.to(DEV), mask, 1).cpu()
        assert torch.equal(got[..., 0], want), ((h, w), (H, W))


def test_packed_mask_rewritten_to_soft_values_is_reported_not_binarised():
    """refresh_packed_mask re-packs a caller-packed mask after an in-place rewrite; values other than 0 and 1 used
    to be binarised at 0.5 in the bits AND in the fp32 copy without a word.  The re-pack now raises the kernel's flag into
    pinned host memory and the next call reports it."""
    import torch
    from lanpaint_amd import pack_mask
    from lanpaint_amd.lanpaint import refresh_packed_mask
    m = (torch.rand((1, 4, 16, 16), device=DEV) > 0.5).float()
    pack_mask(m)
    m.copy_((torch.rand_like(m) > 0.3).float())                       # a binary rewrite: followed silently
    assert refresh_packed_mask(m) is True
    torch.cuda.synchronize()
    assert refresh_packed_mask(m) is False                            # nothing moved since
    m.mul_(0.25)                                                      # now soft
    assert refresh_packed_mask(m) is True
    torch.cuda.synchronize()
    # the flag is read BEFORE the version shortcut -- the very next look reports the rewrite although the version
    # counter has not moved since the re-pack -- and the packed copy is dropped, so later calls run on the fp32 (soft) mask
    # instead of alternating between a silently binarised run and an error
    with pytest.raises(ValueError, match="values other than 0 and 1"):
        refresh_packed_mask(m)
    assert getattr(m, "_lp_bits", None) is None and getattr(m, "_lp_bits_of", None) is None
    assert refresh_packed_mask(m) is False and refresh_packed_mask(m) is False
