"""Shared test plumbing: golden loading, oracle / product drivers."""
from __future__ import annotations

import os

import numpy as np

from oracle.lanpaint_oracle import OracleLanPaint
from tests import golden_cases as gc
from tests.stubs import MODELS

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def xi_list(g):
    if "xi_seed" in g.files:          # compact full-size fixture: the draws are regenerated from the seed
        return gc.seeded_xi(int(g["xi_seed"]), tuple(int(v) for v in g["shape"]), int(g["n_draws"]))
    return [g[f"xi_{i}"] for i in range(int(g["n_draws"]))]


def audio_kwargs(g, conv=lambda a: a):
    if "audio_indicator" not in g.files:
        return {}
    return dict(current_times_audio=(conv(g["ve_a"]), conv(g["abt_a"]), conv(g["flow_a"])),
                audio_indicator=conv(g["audio_indicator"]), audio_correction=conv(g["audio_correction"]))


def run_oracle_case(name, draws=None):
    case = gc.build_case(name)
    g = load_golden(name)
    it = iter(xi_list(g) if draws is None else draws)
    model = MODELS[case["model"]](flow=case["flow"] or case["flux"])
    h = case["hyper"]
    eng = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"],
                         is_flux=case["flux"], is_flow=case["flow"], min_step_frac=h["MinStepFrac"],
                         randn=lambda like: next(it))
    x = case["x"].copy()
    out = eng(x, case["y"], case["noise"], case["sigma"], case["mask"], case["times"], case["model_options"], 0,
              n_steps=case["n_steps"], **audio_kwargs(g))
    return dict(x=x, out=out, model=model, engine=eng, leftover=sum(1 for _ in it), golden=g, case=case)


def run_product_case(name, device="cuda", rng="recorded", model_cls=None, sampling=None, **engine_kw):
    """Run lanpaint_amd.LanPaint on a golden case.  rng="recorded" feeds the reference's xi stream."""
    import torch
    from lanpaint_amd import LanPaint

    case = gc.build_case(name)
    g = load_golden(name)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
    draws = [tt(d) for d in xi_list(g)]
    it = iter(draws)
    if rng == "recorded":
        engine_kw["rng"] = lambda like: next(it)
    else:
        engine_kw["rng"] = rng
    cls = model_cls or MODELS[case["model"]]
    model = cls(flow=case["flow"] or case["flux"], sampling=sampling)
    h = case["hyper"]
    eng = LanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLUX=case["flux"],
                   IS_FLOW=case["flow"], MinStepFrac=h["MinStepFrac"], **engine_kw)
    x = tt(case["x"].copy())
    mo = case["model_options"]
    if mo is not None:
        mo = {k: dict(v) if isinstance(v, dict) else v for k, v in mo.items()}
        mo["lanpaint_semantic_trace"] = []
    out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), tt(case["mask"]),
              tuple(tt(t) for t in case["times"]), mo, 0, n_steps=case["n_steps"], **audio_kwargs(g, tt))
    torch.cuda.synchronize()
    return dict(x=x.cpu().numpy(), out=out.cpu().numpy(), model=model, engine=eng, leftover=sum(1 for _ in it),
                golden=g, case=case, model_options=mo)


def assert_digest(a, g, prefix, sample_seed, what, rel=2e-5):
    """Tensor `a` against the digest stored under `prefix` in fixture `g` (golden_cases.digest): the sampled elements and
    the per-slice sums / sums of squares."""
    d = gc.digest(a, sample_seed)
    assert np.array_equal(d["sample_idx"], g[f"{prefix}_sample_idx"]), f"{what}: sample positions"
    assert_close(d["samples"], g[f"{prefix}_samples"], f"{what}: sampled elements", rel=rel)
    scale = max(1.0, float(np.abs(g[f"{prefix}_samples"]).max()))
    n_slice = np.asarray(a).size / g[f"{prefix}_sums"].size
    # per-element error <= rel * scale, independent roundings: a slice sum moves by ~ sqrt(n) of that; one wrong element
    # (an index, a mask bit, a row's coefficients) moves it by O(scale)
    tol = 4.0 * rel * scale * np.sqrt(n_slice)
    err = float(np.abs(d["sums"] - g[f"{prefix}_sums"]).max())
    assert err <= tol, f"{what}: slice sums off by {err:.3e} > {tol:.3e}"
    np.testing.assert_allclose(d["sumsq"], g[f"{prefix}_sumsq"], rtol=20 * rel, err_msg=f"{what}: slice sums of squares")


def assert_matches_golden(x, out, g, what, rel=2e-5):
    """The written-back x and the returned out of one engine call against a golden fixture: the full tensors, or -- for
    the full-size digest fixtures (golden_cases.digest) -- the sampled elements and the per-slice sums / sums of squares."""
    if "x_out" in g.files:
        assert_close(x, g["x_out"], f"{what}: in-place x", rel=rel)
        assert_close(out, g["out"], f"{what}: out", rel=rel)
        return
    for tag, a, salt in (("x", x, 1), ("out", out, 2)):
        assert_digest(a, g, tag, int(g["xi_seed"]) + salt, f"{what}: {tag}", rel=rel)


def assert_close(a, b, what, rel=2e-5, mse=1e-9):
    """|a-b|_inf <= rel * max(1, |b|_inf) and MSE <= mse (far inside BASELINE's 1e-5)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    m = float(np.mean((a - b) ** 2))
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} > {rel * scale:.3e}"
    assert m <= mse * scale * scale, f"{what}: MSE {m:.3e}"
