"""The oracle is pinned to the reference: every golden trajectory (produced by the
unmodified reference engine, tests/golden/make_golden.py) is reproduced by the numpy
restatement when it is fed the same xi stream."""
import numpy as np
import pytest

from oracle.lanpaint_oracle import OracleLanPaint
from tests import golden_cases as gc
from oracle import lanpaint_oracle as orc
from tests.helpers import assert_close, assert_digest, assert_matches_golden, load_golden, run_oracle_case, xi_list
from tests.stubs import MODELS


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_oracle_matches_reference_golden(name):
    r = run_oracle_case(name)
    g = r["golden"]
    assert r["leftover"] == 0, "oracle consumed a different number of xi draws than the reference"
    assert r["model"].calls == int(g["model_calls"])
    assert_matches_golden(r["x"], r["out"], g, name, rel=3e-6)
    if "trace_dist" in g.files:
        tr = r["engine"].last_stopper.trace
        assert len(tr) == len(g["trace_dist"])
        np.testing.assert_allclose([t["dist"] for t in tr], g["trace_dist"], rtol=2e-5)
        assert [t["counter"] for t in tr] == list(g["trace_counter"])
        assert [t["stopped"] for t in tr] == list(g["trace_stopped"])


def _options_for_call(sc):
    """A fresh copy of the schedule's model_options with a trace list, per sigma call -- what the fixture's generator handed the
    reference (tests/golden/make_golden.py::run_schedule); None for schedules without the inner early stop."""
    if sc.get("model_options") is None:
        return None
    mo = {k: dict(v) if isinstance(v, dict) else v for k, v in sc["model_options"].items()}
    mo["lanpaint_semantic_trace"] = []
    return mo


def check_schedule_traces(g, iterations, traces, name, rtol=2e-5):
    """Iterations run per sigma call and the stopper's records, call after call, against the reference's (fixture)."""
    assert iterations == list(g["iterations"]), f"{name}: iterations per sigma call"
    flat = [t for tr in traces for t in tr]
    assert [k for k, tr in enumerate(traces) for _ in tr] == list(g["trace_call"])
    assert [t["patience_counter"] for t in flat] == list(g["trace_counter"]) and [t["stopped"] for t in flat] == list(g["trace_stopped"])
    for key in ("dist", "dist_inpaint", "dist_ring", "dist_drift", "threshold_eff", "abt"):
        want = g["trace_" + key]
        got = np.asarray([np.nan if t[key] is None else t[key] for t in flat], dtype=np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: {key}"
        np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=rtol, err_msg=f"{name}: {key}")


@pytest.mark.parametrize("name", sorted(gc.SCHEDULES))
def test_oracle_matches_reference_schedule(name):
    sc = gc.build_schedule(name)
    g = load_golden(name)
    it = iter(xi_list(g))
    model = MODELS[sc["model"]](flow=sc["flow"])
    h = sc["hyper"]
    eng = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], is_flow=sc["flow"],
                         min_step_frac=h["MinStepFrac"], randn=lambda like: next(it))
    x = sc["x"].copy()
    sig = sc["sigmas"]
    iterations, traces = [], []
    for i in range(len(sig) - 1):
        s = np.full((sc["shape"][0],), sig[i], dtype=np.float32)
        mo, calls = _options_for_call(sc), model.calls
        den = eng(x, sc["y"], sc["noise"], s, sc["mask"], gc.times_from_sigma(s, sc["flow"]), mo, 0)
        iterations.append(model.calls - calls - 1)
        traces.append(mo["lanpaint_semantic_trace"] if mo is not None else [])
        assert_close(den, g["denoised"][i], f"{name}: denoised[{i}]", rel=2e-5)
        x = (x + (x - den) / sig[i] * (sig[i + 1] - sig[i])).astype(np.float32)
    assert sum(1 for _ in it) == 0
    assert_close(x, g["x_final"], f"{name}: final x", rel=2e-5)
    if "iterations" in g.files:          # the inner early stop over a schedule: a fresh stopper per call, the threshold moving with abt
        check_schedule_traces(g, iterations, traces, name)
        assert len(set(iterations)) >= 3 and min(iterations) < h["NSteps"] == max(iterations)


@pytest.mark.parametrize("name", sorted(gc.NODE_SCHEDULES))
def test_oracle_matches_the_reference_sampler_callable(name):
    """a1 pinned by a reference RUN: the fixture holds what the reference's own KSamplerX0Inpaint.__call__ (nodes.py:229-315)
    did over a schedule with the node defaults -- the inner-step count it chose at every sigma, every denoised, the final x.
    The oracle's restatement of that callable (sigma -> times, mask threshold + invert, the n_eff rule) must choose the same
    counts and walk the same trajectory."""
    sc = gc.build_node_schedule(name)
    g = load_golden(name)
    h, flow, sig = sc["hyper"], sc["flow"], sc["sigmas"]
    it = iter(gc.seeded_xi(int(g["xi_seed"]), sc["shape"], int(g["n_draws"])))
    model = MODELS["linear_tuple"](flow=flow)
    eng = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], is_flow=flow,
                         min_step_frac=h["MinStepFrac"], randn=lambda like: next(it))
    latent_mask = orc.binarize_and_invert(sc["denoise_mask"])
    x, n_eff = sc["x"].copy(), []
    for i in range(len(sig) - 1):
        s = np.full((sc["shape"][0],), sig[i], dtype=np.float32)
        times = orc.times_from_sigma(s, flow)
        n = orc.effective_inner_steps(h["NSteps"], sig, float(s.mean()), float(times[1].mean()), h["EarlyStop"], h["MinStepFrac"])
        n_eff.append(n)
        den = eng(x, sc["y"], sc["noise"], s, latent_mask, times, None, 0, n_steps=n)
        assert_close(den, g["denoised"][i], f"{name}: denoised[{i}]", rel=2e-5)
        x = (x + (x - den) / sig[i] * (sig[i + 1] - sig[i])).astype(np.float32)
    assert n_eff == list(g["n_eff"])
    assert sum(1 for _ in it) == 0 and model.calls == int(g["model_calls"])
    assert_close(x, g["x_final"], f"{name}: final x", rel=2e-5)
    assert len(set(n_eff)) >= 5 and n_eff[-1] == 0 and max(n_eff) == h["NSteps"]      # the ramp is really exercised


@pytest.mark.parametrize("name", sorted(gc.FULL_SCHEDULES))
def test_oracle_matches_reference_full_baseline_schedule(name):
    """BASELINE.json's C1 ... C5 as WHOLE schedules (C2 = the headline configuration: SDXL 1x4x128x128, 30 sigmas x 5; C3 = four
    rows each on its own sigma ramp; C5 = the 5-D video latent with the temporal mask through reshape_mask) run by the
    unmodified reference engine; the fixture holds digests of every fifth denoised and of the final x."""
    sc = gc.build_full_schedule(name)
    g = load_golden(name)
    it = gc.seeded_xi_stream(int(g["xi_seed"]), sc["shape"])
    h, flow, sig = sc["hyper"], sc["flow"], sc["sigmas"]
    model = MODELS["linear_tuple"](flow=flow)
    drawn = [0]

    def randn(like):
        drawn[0] += 1
        return next(it)
    eng = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], is_flow=flow,
                         min_step_frac=h["MinStepFrac"], randn=randn)
    x = sc["x"].copy()
    if sc["video_mask"]:
        # C5: the job's latent mask through the ORACLE's reshape_mask (video path) from the pixel-resolution mask -- must be
        # the reference's, bit for bit (the fixture keeps the reference's as packed bits)
        dm = orc.reshape_mask(gc.video_pixel_mask(sc["shape"]), sc["shape"], video_inpainting=True)
        sc["mask"] = orc.binarize_and_invert(dm).astype(np.float32)
        assert np.array_equal(np.packbits(sc["mask"].reshape(-1) > 0.5), g["mask_bits"]) and int(sc["mask"].sum()) == int(g["mask_known"])
    for i in range(len(sig)):
        s = np.full((sc["shape"][0],), sig[i], dtype=np.float32) * sc["row_scale"]
        den = eng(x, sc["y"], sc["noise"], s, sc["mask"], gc.times_from_sigma(s, flow), None, 0)
        if f"den{i}_sums" in g.files:
            assert_digest(den, g, f"den{i}", int(g["xi_seed"]) + 10 + i, f"{name}: denoised[{i}]", rel=2e-5)
        if i + 1 < len(sig):
            x = (x + (x - den) / sig[i] * (sig[i + 1] - sig[i])).astype(np.float32)
    assert drawn[0] == int(g["n_draws"]) and model.calls == int(g["model_calls"])
    assert_digest(x, g, "x", int(g["xi_seed"]) + 1, f"{name}: final x", rel=2e-5)


def test_oracle_on_torch_backend_matches_numpy():
    """The cpu_baseline leg runs the same oracle on torch-CPU tensors."""
    import torch
    from oracle.lanpaint_oracle import TorchBackend
    name = "ve_basic"
    case = gc.build_case(name)
    g = load_golden(name)
    it = iter([torch.from_numpy(d) for d in xi_list(g)])
    model = MODELS[case["model"]]()
    h = case["hyper"]
    eng = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"],
                         min_step_frac=h["MinStepFrac"], backend=TorchBackend(), randn=lambda like: next(it))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))   # noqa: E731
    x = t(case["x"].copy())
    out = eng(x, t(case["y"]), t(case["noise"]), t(case["sigma"]), t(case["mask"]),
              tuple(t(v) for v in case["times"]), None, 0)
    assert_close(x.numpy(), g["x_out"], "torch-backend x", rel=3e-6)
    assert_close(out.numpy(), g["out"], "torch-backend out", rel=3e-6)


TRACE_CASES = [n for n in sorted(gc.CASES) if (gc.CASES[n].get("model_options") or {}).get("lanpaint_semantic_stop")]


@pytest.mark.parametrize("backend", ["numpy", "torch"])
@pytest.mark.parametrize("name", TRACE_CASES)
def test_port_writes_the_references_trace_records(name, backend):
    """The record the reference's stopper appends per inner iteration (earlystop.py:315-334: fifteen keys) as the port writes
    it into `model_options["lanpaint_semantic_trace"]`, on both array backends -- the torch one is what the GPU suite runs on
    the device as the same-seed arbiter (tests/test_gpu_port_on_device.py), so it is pinned here to what the unmodified
    reference wrote into the fixture."""
    import torch
    from oracle.lanpaint_oracle import TorchBackend
    case, g = gc.build_case(name), load_golden(name)
    conv = (lambda a: a) if backend == "numpy" else (lambda a: torch.from_numpy(np.ascontiguousarray(a)))
    it = iter([conv(d) for d in xi_list(g)])
    h = case["hyper"]
    eng = OracleLanPaint(MODELS[case["model"]](flow=case["flow"]), h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"],
                         is_flow=case["flow"], min_step_frac=h["MinStepFrac"], randn=lambda like: next(it),
                         backend=None if backend == "numpy" else TorchBackend())
    mo = {k: dict(v) if isinstance(v, dict) else v for k, v in case["model_options"].items()}
    mo.update({"lanpaint_semantic_trace": [], "bench_case_id": "case-7", "bench_outer_step": 3, "bench_timestep": 0.25})
    x = conv(case["x"].copy())
    out = eng(x, conv(case["y"]), conv(case["noise"]), conv(case["sigma"]), conv(case["mask"]), tuple(conv(t) for t in case["times"]),
              mo, 0, n_steps=case["n_steps"])
    assert_matches_golden(np.asarray(x), np.asarray(out), g, name, rel=3e-6)
    tr = mo["lanpaint_semantic_trace"]
    assert len(tr) == len(g["trace_dist"]) and sorted(tr[0]) == list(g["trace_keys"])
    assert [r["inner_step"] for r in tr] == list(g["trace_inner_step"]) and [r["patience_eff"] for r in tr] == list(g["trace_patience_eff"])
    assert [r["patience_counter"] for r in tr] == list(g["trace_counter"]) and [r["stopped"] for r in tr] == list(g["trace_stopped"])
    assert all(r["case_id"] == "case-7" and r["outer_step"] == 3 and r["bench_timestep"] == 0.25 and r["custom_dist"] is False for r in tr)
    for key in ("dist", "dist_inpaint", "dist_ring", "dist_drift", "threshold", "threshold_eff", "abt"):
        want = g["trace_" + key]
        got = np.asarray([np.nan if r[key] is None else r[key] for r in tr], dtype=np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want)), key            # None exactly where the reference has None
        np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=2e-5, err_msg=key)
