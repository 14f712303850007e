"""Seeded random sweep of the engine against the CPU oracle: shapes (odd, 5-D, multi-row, both sides of the
VEC=1 / VEC=4 switch), schedules, hyper-parameters, mask densities, step counts, model output forms."""
import numpy as np
import pytest

from oracle.lanpaint_oracle import OracleLanPaint, times_from_sigma
from tests.helpers import assert_close
from tests.stubs import MODELS

pytestmark = pytest.mark.gpu

SHAPES = [(1, 4, 8, 8), (2, 4, 9, 7), (1, 3, 5, 7), (3, 2, 6, 10), (1, 16, 3, 6, 8), (2, 4, 2, 5, 5), (1, 1, 33),
          (4, 4, 16, 16), (1, 4, 64, 64), (5, 1, 7, 3)]
BIG = [(3, 4, 224, 224), (2, 3, 301, 301), (1, 16, 21, 60, 104)]        # > 512 K elements: float4 / large scalar paths


def _case(seed, shape=None):
    rng = np.random.default_rng(seed)
    shape = shape or SHAPES[int(rng.integers(len(SHAPES)))]
    flow = bool(rng.integers(2))
    rows = shape[0]
    per_row = bool(rng.integers(2))
    if flow:
        sig = rng.uniform(0.03, 0.97, size=rows if per_row else 1)
    else:
        sig = np.exp(rng.uniform(np.log(0.03), np.log(14.6), size=rows if per_row else 1))
    sigma = np.broadcast_to(sig, (rows,)).astype(np.float32).copy()
    hyper = dict(lamb=float(rng.choice([1.0, 5.0, 8.0, 0.3])), beta=float(rng.choice([1.0, 0.5, 2.0])),
                 step=float(rng.choice([0.2, 0.05, 0.4])), msf=float(rng.choice([0.0, 0.0, 1.0, 0.3])))
    n_steps = int(rng.choice([0, 1, 2, 3, 5]))
    kind = rng.choice(["box", "random", "soft", "ones", "zeros"], p=[0.35, 0.35, 0.1, 0.1, 0.1])
    if kind == "box":
        mask = np.zeros(shape, dtype=np.float32)
        mask[..., : max(1, shape[-1] // 2)] = 1.0
    elif kind == "random":
        mask = (rng.random(shape) > rng.uniform(0.2, 0.8)).astype(np.float32)
    elif kind == "soft":
        mask = rng.random(shape, dtype=np.float32)
    else:
        mask = np.full(shape, 1.0 if kind == "ones" else 0.0, dtype=np.float32)
    model = str(rng.choice(["linear_tuple", "denoiser_single", "offset_tuple", "list_one"]))
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    x = rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 + sigma.max())
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(max(0, 2 * n_steps - 1))]
    return dict(shape=shape, flow=flow, sigma=sigma, hyper=hyper, n_steps=n_steps, mask=mask, model=model, y=y, noise=noise,
                x=x, draws=draws, kind=kind)


def _run(c):
    import torch
    from lanpaint_amd import LanPaint
    h = c["hyper"]
    times = times_from_sigma(c["sigma"], c["flow"])
    it = iter(c["draws"])
    o = OracleLanPaint(MODELS[c["model"]](flow=c["flow"]), 5, 15.0, h["lamb"], h["beta"], h["step"], is_flow=c["flow"],
                       min_step_frac=h["msf"], randn=lambda like: next(it))
    xo = c["x"].copy()
    out_o = o(xo, c["y"], c["noise"], c["sigma"], c["mask"], times, None, 0, n_steps=c["n_steps"])
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    it2 = iter(c["draws"])
    eng = LanPaint(MODELS[c["model"]](flow=c["flow"]), 5, 15.0, h["lamb"], h["beta"], h["step"], IS_FLOW=c["flow"],
                   MinStepFrac=h["msf"], rng=lambda like: tt(next(it2)))
    xg = tt(c["x"])
    out_g = eng(xg, tt(c["y"]), tt(c["noise"]), tt(c["sigma"]), tt(c["mask"]), tuple(tt(t) for t in times), None, 0,
                n_steps=c["n_steps"])
    what = f"shape={c['shape']} flow={c['flow']} sigma={c['sigma']} {h} n={c['n_steps']} mask={c['kind']} model={c['model']}"
    assert_close(xg.cpu().numpy(), xo, "x | " + what, rel=5e-5, mse=1e-8)
    assert_close(out_g.cpu().numpy(), out_o, "out | " + what, rel=5e-5, mse=1e-8)
    assert next(it, None) is None and next(it2, None) is None


@pytest.mark.parametrize("seed", range(48))
def test_random_config_matches_oracle(seed):
    _run(_case(1000 + seed))


@pytest.mark.parametrize("idx", range(len(BIG)))
def test_large_shapes_both_kernel_widths(idx):
    c = _case(7000 + idx, BIG[idx])
    c["n_steps"] = 2
    rng = np.random.default_rng(idx)
    c["draws"] = [rng.standard_normal(c["shape"], dtype=np.float32) for _ in range(3)]
    _run(c)


def _run_engine(c, rng, pack, graph, calls=2):
    """`calls` consecutive sigma calls on the same tensors (the second one replays in graph mode)."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import LanPaint
    h = c["hyper"]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    times = tuple(tt(t) for t in times_from_sigma(c["sigma"], c["flow"]))
    eng = LanPaint(MODELS[c["model"]](flow=c["flow"]), 5, 15.0, h["lamb"], h["beta"], h["step"], IS_FLOW=c["flow"],
                   MinStepFrac=h["msf"], rng=rng, graph=graph)
    mask = tt(c["mask"])
    if pack:
        mask = lanpaint_amd.pack_mask(mask)
    x, y, noise, sigma = tt(c["x"]), tt(c["y"]), tt(c["noise"]), tt(c["sigma"])
    torch.manual_seed(99)
    outs = []
    for _ in range(calls):
        out = eng(x, y, noise, sigma, mask, times, {}, 0, n_steps=c["n_steps"])
        outs.append((x.clone().cpu(), out.clone().cpu()))
    return outs, torch.cuda.default_generators[0].get_offset()


@pytest.mark.parametrize("seed", range(40))
def test_fast_paths_equal_the_plain_path_bitwise(seed):
    """Everything that makes the think loop fast -- bit-packed mask (hard-mask kernels), the torch noise stream
    generated in-kernel, the coefficient table folded into the replace launch, hipGraph replay through
    lp_replay_call -- against the plain path (fp32 mask, torch.randn_like tensors, eager launches) on random
    configurations: bitwise the same x / out on two consecutive calls, same generator state afterwards."""
    c = _case(3000 + seed, BIG[seed % len(BIG)] if seed % 8 == 7 else None)
    if c["kind"] == "soft":                    # a soft mask cannot be packed: exercise the rest of the fast path
        pack = False
    else:
        pack = True
    if seed % 8 == 7:
        c["n_steps"] = 2
    plain, off_plain = _run_engine(c, "torch-eager", pack=False, graph=False)
    fast, off_fast = _run_engine(c, "torch", pack=pack, graph=True)
    what = f"shape={c['shape']} flow={c['flow']} n={c['n_steps']} mask={c['kind']} model={c['model']} {c['hyper']}"
    assert off_plain == off_fast, what
    for (xa, oa), (xb, ob) in zip(plain, fast):
        assert np.array_equal(xa.numpy(), xb.numpy()), "x | " + what
        assert np.array_equal(oa.numpy(), ob.numpy()), "out | " + what


# ---------------------------------------------------------------- inner early stop on the device vs the oracle stopper
def _es_case(seed):
    rng = np.random.default_rng(10_000 + seed)
    shape = [(1, 4, 8, 8), (2, 4, 9, 7), (1, 3, 5, 7), (3, 2, 6, 10), (1, 8, 3, 6, 8), (4, 4, 16, 16), (1, 4, 40, 40),
             (1, 1, 33)][int(rng.integers(8))]
    flow = bool(rng.integers(2))
    rows = shape[0]
    per_row = bool(rng.integers(2))
    sig = rng.uniform(0.25, 0.75, size=rows if per_row else 1) if flow else \
        np.exp(rng.uniform(np.log(0.4), np.log(2.5), size=rows if per_row else 1))
    sigma = np.broadcast_to(sig, (rows,)).astype(np.float32).copy()
    kind = str(rng.choice(["box", "random", "soft"], p=[0.45, 0.4, 0.15]))
    if kind == "box":
        mask = np.zeros(shape, dtype=np.float32)
        mask[..., : max(1, shape[-1] // 2)] = 1.0
    elif kind == "random":
        mask = (rng.random(shape) > rng.uniform(0.3, 0.7)).astype(np.float32)
    else:
        mask = rng.random(shape, dtype=np.float32)
    n = int(rng.choice([4, 7, 10]))
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    sb = sigma.reshape((-1,) + (1,) * (len(shape) - 1))
    x = ((sb * noise + (1 - sb) * y) if flow else (y + noise * sb)).astype(np.float32)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(2 * n)]
    return dict(shape=shape, flow=flow, sigma=sigma, mask=mask, n=n, y=y, noise=noise, x=x, draws=draws, kind=kind,
                patience=int(rng.choice([1, 1, 2])), lamb=float(rng.choice([5.0, 2.0])), quantile=float(rng.uniform(0.3, 0.8)))


@pytest.mark.parametrize("seed", range(16))
def test_random_early_stop_configurations_match_the_oracle_stopper(seed):
    """Seeded sweep of the device-side stop rule (LP_FL_ES) against the oracle's restatement of earlystop.py:58-336:
    shapes with and without a ring (4-D / other ranks), per-row sigmas, hard and soft masks, patience 1-2.  The
    threshold is placed inside the run's own distance distribution (a quantile of a stopper-free dry run, nudged away
    from every observed distance so fp32-vs-double summation cannot flip a comparison), so stops, resets and drift
    vetoes all occur."""
    import torch
    from lanpaint_amd import LanPaint
    c = _es_case(seed)
    times = times_from_sigma(c["sigma"], c["flow"])

    def oracle(thr):
        it = iter(c["draws"])
        o = OracleLanPaint(MODELS["linear_tuple"](flow=c["flow"]), c["n"], 15.0, c["lamb"], 1.0, 0.2, is_flow=c["flow"],
                           randn=lambda like: next(it))
        xo = c["x"].copy()
        out = o(xo, c["y"], c["noise"], c["sigma"], c["mask"], times,
                {"lanpaint_semantic_stop": {"threshold": thr, "patience": c["patience"]}}, 0)
        return o, xo, out, sum(1 for _ in it)

    dry, _, _, _ = oracle(1e-30)                                        # never stops: the distances of the full run
    dists = np.asarray([t["dist"] for t in dry.last_stopper.trace])
    scale = dry.last_stopper.threshold_eff / 1e-30                      # abt scaling of the threshold
    thr_eff = float(np.quantile(dists, c["quantile"]))
    gaps = np.abs(dists - thr_eff) / thr_eff
    if gaps.min() < 1e-3:
        thr_eff *= 1.0 + 4e-3
    o, xo, out_o, left_o = oracle(thr_eff / scale)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    it2 = iter([tt(d) for d in c["draws"]])
    eng = LanPaint(MODELS["linear_tuple"](flow=c["flow"]), c["n"], 15.0, c["lamb"], 1.0, 0.2, IS_FLOW=c["flow"],
                   rng=lambda like: next(it2))
    trace = []
    mo = {"lanpaint_semantic_stop": {"threshold": thr_eff / scale, "patience": c["patience"]}, "lanpaint_semantic_trace": trace}
    xg = tt(c["x"])
    out_g = eng(xg, tt(c["y"]), tt(c["noise"]), tt(c["sigma"]), tt(c["mask"]), tuple(tt(t) for t in times), mo, 0)
    torch.cuda.synchronize()
    what = f"seed={seed} shape={c['shape']} flow={c['flow']} mask={c['kind']} n={c['n']} patience={c['patience']}"
    tr_o = o.last_stopper.trace
    assert eng.iterations_run == o.iterations_run, what
    assert [t["patience_counter"] for t in trace] == [t["counter"] for t in tr_o], what
    assert [t["stopped"] for t in trace] == [t["stopped"] for t in tr_o], what
    np.testing.assert_allclose([t["dist"] for t in trace], [t["dist"] for t in tr_o], rtol=3e-4, err_msg=what)
    assert sum(1 for _ in it2) == left_o, what
    assert_close(xg.cpu().numpy(), xo, what + " x", rel=5e-5)
    assert_close(out_g.cpu().numpy(), out_o, what + " out", rel=5e-5)
