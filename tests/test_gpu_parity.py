"""Parity of the HIP path (through the C ABI) with the reference's golden trajectories
and with the CPU oracle, on identical (latent, mask, sigma, xi stream).

Tolerance: max-abs error <= 2e-5 * max(1, |ref|_inf) and MSE <= 1e-9 * scale^2 -- four
orders of magnitude inside BASELINE.json's "output MSE vs reference < 1e-5"."""
import numpy as np
import pytest

from oracle.lanpaint_oracle import OracleLanPaint
from tests import golden_cases as gc
from tests.helpers import assert_close, assert_matches_golden, load_golden, run_oracle_case, run_product_case, xi_list
from tests.stubs import MODELS, OpaqueVESampling

pytestmark = pytest.mark.gpu

EARLYSTOP = {"ve_earlystop", "ve_earlystop_fast", "ve_earlystop_run_all"}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from lanpaint_amd import _cabi
    _cabi.load()


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_hip_matches_reference_golden(name):
    r = run_product_case(name)
    g = r["golden"]
    assert r["leftover"] == 0, "HIP path consumed a different number of xi draws than the reference"
    assert r["model"].calls == int(g["model_calls"])
    assert_matches_golden(r["x"], r["out"], g, name)
    if name in EARLYSTOP:
        tr = r["model_options"]["lanpaint_semantic_trace"]
        assert len(tr) == len(g["trace_dist"])
        np.testing.assert_allclose([t["dist"] for t in tr], g["trace_dist"], rtol=1e-4)
        assert [t["patience_counter"] for t in tr] == list(g["trace_counter"])
        assert [t["stopped"] for t in tr] == list(g["trace_stopped"])


@pytest.mark.parametrize("name", ["ve_basic", "flow_basic", "ve_batch_rows", "ve_soft_mask", "av_flat_pack"])
def test_hip_matches_cpu_oracle(name):
    o = run_oracle_case(name)
    r = run_product_case(name)
    assert_close(r["x"], o["x"], f"{name}: x vs oracle")
    assert_close(r["out"], o["out"], f"{name}: out vs oracle")


@pytest.mark.parametrize("name", sorted(gc.SCHEDULES))
def test_hip_matches_reference_schedule(name):
    """Whole schedules the unmodified reference walked (fixtures), its recorded xi stream fed to the kernels (a callable rng:
    eager launches).  `sched_ve_earlystop8` runs with the inner early stop armed: the iterations run per sigma call (8, 8, 4, 2,
    ...: the abt-scaled threshold moves with sigma) and every record of the stopper's trace must be the reference's -- here
    through the device-side rule with the host reading one verdict per iteration; the device-gated launches of a replayed graph
    are compared with the port under one torch seed in tests/test_gpu_port_on_device.py."""
    import torch
    from lanpaint_amd import LanPaint
    from tests.test_oracle_golden import _options_for_call, check_schedule_traces
    sc = gc.build_schedule(name)
    g = load_golden(name)
    dev = "cuda"
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    it = iter([tt(d) for d in xi_list(g)])
    model = MODELS[sc["model"]](flow=sc["flow"])
    h = sc["hyper"]
    eng = LanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=sc["flow"],
                   MinStepFrac=h["MinStepFrac"], rng=lambda like: next(it))
    x, y, noise, mask = tt(sc["x"].copy()), tt(sc["y"]), tt(sc["noise"]), tt(sc["mask"])
    sig = sc["sigmas"]
    iterations, traces = [], []
    for i in range(len(sig) - 1):
        s = torch.full((sc["shape"][0],), float(sig[i]), dtype=torch.float32, device=dev)
        times = gc.times_from_sigma(s, sc["flow"])
        mo, before = _options_for_call(sc), eng.iterations_run
        den = eng(x, y, noise, s, mask, times, mo, 0)
        iterations.append(eng.iterations_run - before)
        traces.append(mo["lanpaint_semantic_trace"] if mo is not None else [])
        assert_close(den.cpu().numpy(), g["denoised"][i], f"{name}: denoised[{i}]", rel=5e-5)
        x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
    assert sum(1 for _ in it) == 0
    assert_close(x.cpu().numpy(), g["x_final"], f"{name}: final x", rel=5e-5)
    if "iterations" in g.files:
        check_schedule_traces(g, iterations, traces, name, rtol=2e-4)


def test_noise_scaling_callback_path_equals_fused():
    """An undeclared model_sampling goes through its own noise_scaling (generic
    drop-in path); the result equals the fused VE form."""
    a = run_product_case("ve_basic")
    b = run_product_case("ve_basic", sampling=OpaqueVESampling())
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["out"], b["out"])


def test_inplace_write_back_and_fresh_output():
    import torch
    from lanpaint_amd import LanPaint
    case = gc.build_case("ve_basic")
    dev = "cuda"
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    model = MODELS["linear_tuple"]()
    eng = LanPaint(model, 2, 15.0, 5.0, 1.0, 0.2, rng="philox")
    x = tt(case["x"].copy())
    x_before = x.clone()
    ptr = x.data_ptr()
    args = (tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), tt(case["mask"]), tuple(tt(t) for t in case["times"]))
    out1 = eng(x, *args, None, 0)
    assert x.data_ptr() == ptr and not torch.equal(x, x_before)       # mutated in place (lanpaint.py:156)
    out2 = eng(x, *args, None, 0)
    assert out1.data_ptr() != out2.data_ptr()                         # samplers keep old `denoised` tensors
    assert torch.equal(model.last_input, x)                           # final model call saw the written-back x
    known = case["mask"] == 1
    np.testing.assert_array_equal(out2.cpu().numpy()[known], case["y"][known])   # hard reprojection (lanpaint.py:154)


def test_non_contiguous_and_half_model_outputs():
    """Backbones may return bf16/fp16 or strided tensors; x may be a strided view."""
    import torch
    from lanpaint_amd import LanPaint
    case = gc.build_case("ve_basic")
    g = load_golden("ve_basic")
    dev = "cuda"
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

    class Strided(MODELS["linear_tuple"]):
        def __call__(self, x, t, model_options=None, seed=None):
            a, b = super().__call__(x, t)
            wide = torch.empty(a.shape[:-1] + (a.shape[-1] * 2,), device=a.device)
            wide[..., ::2] = a
            return wide[..., ::2], b.to(torch.float64)

    it = iter([tt(d) for d in xi_list(g)])
    eng = LanPaint(Strided(), 5, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
    big = torch.zeros((1, 4, 8, 16), device=dev)
    x = big[..., ::2]
    x.copy_(tt(case["x"]))
    out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), tt(case["mask"]),
              tuple(tt(t) for t in case["times"]), None, 0)
    assert_close(x.cpu().numpy(), g["x_out"], "strided x")
    assert_close(out.cpu().numpy(), g["out"], "strided out")

    class Bf16(MODELS["linear_tuple"]):
        def __call__(self, x, t, model_options=None, seed=None):
            a, b = super().__call__(x, t)
            return a.to(torch.bfloat16), b.to(torch.bfloat16)

    it = iter([tt(d) for d in xi_list(g)])
    eng = LanPaint(Bf16(), 5, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it))
    x = tt(case["x"].copy())
    out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), tt(case["mask"]),
              tuple(tt(t) for t in case["times"]), None, 0)
    # bf16 heads: 8 mantissa bits -> ~4e-3 relative; still far inside MSE < 1e-5 * scale^2
    assert float(np.mean((out.cpu().numpy() - g["out"]) ** 2)) < 1e-3
    assert np.isfinite(x.cpu().numpy()).all()


def test_zero_step_size_skips_iterations():
    """StepSize <= 0: every iteration is a no-op and the backbone is only called for the
    final denoise (lanpaint.py:205)."""
    r = run_product_case("ve_n0")
    import torch
    from lanpaint_amd import LanPaint
    case = gc.build_case("ve_n0")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    model = MODELS["linear_tuple"]()
    eng = LanPaint(model, 5, 15.0, 5.0, 1.0, 0.0, rng="philox", graph=False)      # (a capture would call the backbone while warming up)
    x = tt(case["x"].copy())
    out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), tt(case["mask"]),
              tuple(tt(t) for t in case["times"]), None, 0)
    assert model.calls == 1
    assert_close(out.cpu().numpy(), r["out"], "zero-step out")
    assert_close(x.cpu().numpy(), r["x"], "zero-step x")


# ------------------------------------------------------------------ hipGraph replay mode
def _sched_run(graph, rng, seed=123, n_sigmas=5, shape=(1, 4, 16, 16)):
    import torch
    from lanpaint_amd import LanPaint
    dev = "cuda"
    g = np.random.default_rng(7)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = tt(gc.box_mask(shape))
    sig = gc.karras_sigmas(n_sigmas)[:-1]
    x = y + noise * float(sig[0])
    model = MODELS["linear_tuple"]()
    eng = LanPaint(model, 3, 15.0, 5.0, 1.0, 0.2, rng=rng, philox_seed=seed, graph=graph)
    torch.manual_seed(seed)
    outs = []
    for i in range(len(sig)):
        s = torch.full((shape[0],), float(sig[i]), dtype=torch.float32, device=dev)
        den = eng(x, y, noise, s, mask, gc.times_from_sigma(s, False), None, seed)
        outs.append(den)
        if i + 1 < len(sig):
            x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
    torch.cuda.synchronize()
    return x.cpu().numpy(), [o.cpu().numpy() for o in outs], eng, model


def test_graph_replay_equals_eager_with_torch_rng():
    """Same torch seed -> the captured-and-replayed sigma calls consume the device generator
    exactly like eager launches, so the whole trajectory is identical."""
    xe, oe, eng_e, _ = _sched_run(False, "torch")
    xg, og, eng_g, _ = _sched_run(True, "torch")
    assert len(eng_g._graphs) == 1 and eng_g.iterations_run == eng_e.iterations_run == 15
    np.testing.assert_array_equal(xe, xg)
    for a, b in zip(oe, og):
        np.testing.assert_array_equal(a, b)
    assert len({o.ctypes.data for o in og}) == len(og)


def test_graph_replay_philox_draws_fresh_noise_each_replay():
    xg, og, eng, _ = _sched_run(True, "philox")
    xe, oe, _, _ = _sched_run(False, "philox")
    assert np.isfinite(xg).all()
    # different sigma calls (replays of ONE graph) must not repeat the same noise
    import torch
    from lanpaint_amd import LanPaint
    dev = "cuda"
    shape = (1, 4, 16, 16)
    z = torch.zeros(shape, device=dev)
    m = torch.zeros(shape, device=dev)
    eng = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=5, graph=True)
    s = torch.full((1,), 1.0, device=dev)
    res = []
    for _ in range(3):
        x = z.clone()
        eng(x, z, z + 1.0, s, m, gc.times_from_sigma(s, False), None, 0)
        res.append(x.cpu().numpy().copy())
    assert not np.array_equal(res[0], res[1]) and not np.array_equal(res[1], res[2])
    # same statistics as the eager path: zero-mean noise of equal scale
    assert abs(np.std(res[0]) / np.std(res[1]) - 1.0) < 0.1
    # the trajectory statistics of graph and eager philox runs agree loosely (different streams)
    assert abs(np.std(xg) - np.std(xe)) < 0.25 * np.std(xe)


def test_graph_replay_follows_hyperparameter_changes_like_the_reference():
    """The reference reads Lambda / Beta / StepSize / MinStepFrac on every call (lanpaint.py:81,183,316): changing
    them between sigma calls must take effect under graph replay too (they are part of the graph key)."""
    import torch
    from lanpaint_amd import LanPaint
    dev, shape = "cuda", (1, 4, 16, 16)
    g = np.random.default_rng(3)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = tt(gc.box_mask(shape))
    s = torch.full((1,), 1.3, device=dev)
    times = gc.times_from_sigma(s, False)
    res = {}
    for graph in (False, True):
        torch.manual_seed(11)
        eng = LanPaint(MODELS["linear_tuple"](), 3, 15.0, 5.0, 1.0, 0.2, rng="torch", graph=graph)
        outs = []
        for lamb, step, beta, msf in ((5.0, 0.2, 1.0, 0.0), (5.0, 0.2, 1.0, 0.0), (2.0, 0.2, 1.0, 0.0), (2.0, 0.1, 1.0, 0.0),
                                      (2.0, 0.1, 0.5, 0.0), (2.0, 0.1, 0.5, 0.7), (5.0, 0.2, 1.0, 0.0)):
            eng.chara_lamb, eng.step_size, eng.chara_beta, eng.min_step_frac = lamb, step, beta, msf
            x = y + noise * 1.3
            outs.append((eng(x, y, noise, s, mask, times, None, 0).cpu().numpy(), x.cpu().numpy()))
        res[graph] = outs
        if graph:
            assert len(eng._graphs) == 5          # one capture per distinct hyper-parameter set
    for (oe, xe), (og, xg) in zip(res[False], res[True]):
        np.testing.assert_array_equal(oe, og)
        np.testing.assert_array_equal(xe, xg)
    assert not np.array_equal(res[True][1][0], res[True][2][0])


def test_graph_replay_outputs_are_fresh_tensors_that_stay_valid():
    """Multi-step samplers keep earlier `denoised` tensors (dpmpp_2m: old_denoised): every replay returns its own
    tensor, written through the I/O table by the captured lp_finalize, and x may move between calls."""
    import torch
    from lanpaint_amd import LanPaint
    dev, shape = "cuda", (2, 4, 24, 24)
    g = np.random.default_rng(4)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = tt(gc.box_mask(shape))
    s = torch.full((2,), 0.9, device=dev)
    times = gc.times_from_sigma(s, False)
    runs = {}
    for graph in (False, True):
        torch.manual_seed(5)
        eng = LanPaint(MODELS["linear_tuple"](), 2, 15.0, 5.0, 1.0, 0.2, rng="torch", graph=graph)
        kept, xs = [], []
        for k in range(5):
            x = (y + noise * 0.9 + 0.01 * k).clone()           # a new tensor (new address) every call
            kept.append(eng(x, y, noise, s, mask, times, None, 0))
            xs.append(x)
        torch.cuda.synchronize()
        assert len({t.data_ptr() for t in kept}) == 5
        runs[graph] = ([t.cpu().numpy() for t in kept], [t.cpu().numpy() for t in xs])
        if graph:
            cap = next(iter(eng._graphs.values()))
            assert cap.final_in_graph and cap.fast
    for a, b in zip(runs[False][0] + runs[False][1], runs[True][0] + runs[True][1]):
        np.testing.assert_array_equal(a, b)


def test_one_graph_launch_per_sigma_call_sees_each_calls_own_arguments(monkeypatch):
    """Round 3: the replace launch is node 0 of the captured call and every replay rewrites that node's arguments
    (hipGraphExecKernelNodeSetParams) instead of launching it eagerly.  The update must only affect launches enqueued
    AFTER it: with the GPU held busy, sixteen sigma calls on sixteen different x tensors are queued back to back -- each
    must read ITS x / sigma and write ITS out -- and the result must equal both eager launches and the round-2 layout
    (replace outside the graph, LANPAINT_AMD_REPLACE_IN_GRAPH=0) bit for bit."""
    import torch
    from lanpaint_amd import LanPaint
    dev, shape, n_calls = "cuda", (2, 4, 32, 32), 16
    g = np.random.default_rng(31)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = tt(gc.box_mask(shape))
    sigs = [torch.full((2,), 0.5 + 0.1 * k, device=dev) for k in range(n_calls)]
    times = [gc.times_from_sigma(s, False) for s in sigs]
    x_src = [tt(g.standard_normal(shape, dtype=np.float32)) for _ in range(n_calls)]

    def run(graph, in_graph, busy):
        monkeypatch.setenv("LANPAINT_AMD_REPLACE_IN_GRAPH", "1" if in_graph else "0")
        torch.manual_seed(9)
        eng = LanPaint(MODELS["linear_tuple"](), 3, 15.0, 5.0, 1.0, 0.2, rng="torch", graph=graph)
        xs = [t.clone() for t in x_src]
        eng(xs[0].clone(), y, noise, sigs[0], mask, times[0], None, 0)          # capture + first replay
        torch.manual_seed(9)
        torch.cuda.synchronize()
        if busy:
            torch.cuda._sleep(int(40e6))                                          # ~20 ms: the host runs far ahead
        outs = [eng(xs[k], y, noise, sigs[k], mask, times[k], None, 0) for k in range(n_calls)]
        torch.cuda.synchronize()
        if graph:
            cap = next(iter(eng._graphs.values()))
            assert (cap.binding is not None) == in_graph and cap.fast and cap.final_in_graph
            assert (cap.tail is not None)
        return [o.cpu().numpy() for o in outs], [x.cpu().numpy() for x in xs]

    eager = run(False, True, False)
    one_launch = run(True, True, True)
    round2 = run(True, False, True)
    for a, b, c in zip(eager[0] + eager[1], one_launch[0] + one_launch[1], round2[0] + round2[1]):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)
    assert not np.array_equal(one_launch[0][0], one_launch[0][1])


# ------------------------------------------------------------------ inner early stop decided on the device
@pytest.mark.parametrize("name", sorted(EARLYSTOP))
@pytest.mark.parametrize("rng", ["torch", "philox"])
def test_early_stop_inside_a_replayed_graph_equals_the_host_watched_loop(name, rng):
    """Graph replay cannot ask the host after every iteration: its launches are gated on the device-side stop flag
    (LP_FL_ES_GATED: tentative half-step, redone by the next launch from the same noise).  Against the eager loop,
    which reads the verdict from the mailbox each iteration and is pinned to the reference's golden traces above:
    same iteration count, same trace, bitwise the same x / out, and torch's generator left in the same place --
    over several sigma calls (replays) with the stop landing on different iterations."""
    import torch
    from lanpaint_amd import LanPaint
    case = gc.build_case(name)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    h = case["hyper"]
    stop = case["model_options"]["lanpaint_semantic_stop"]
    res = {}
    for graph in (False, True):
        torch.manual_seed(4242)
        model = MODELS[case["model"]](flow=case["flow"] or case["flux"])
        eng = LanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLUX=case["flux"],
                       IS_FLOW=case["flow"], MinStepFrac=h["MinStepFrac"], rng=rng, philox_seed=9, graph=graph)
        y, noise, mask = tt(case["y"]), tt(case["noise"]), tt(case["mask"])
        times = tuple(tt(t) for t in case["times"])
        sig = tt(case["sigma"])
        runs = []
        for rep in range(4):
            mo = {"lanpaint_semantic_stop": dict(stop), "lanpaint_semantic_trace": []} if rep != 2 else \
                {"lanpaint_semantic_stop": dict(stop)}
            x = tt(case["x"] + np.float32(0.05 * rep))
            it0 = eng.iterations_run
            out = eng(x, y, noise, sig, mask, times, mo, 0, n_steps=case["n_steps"])
            ran = eng.iterations_run - it0
            tr = mo.get("lanpaint_semantic_trace", [])
            runs.append((x.cpu().numpy(), out.cpu().numpy(), ran,
                         [(t["inner_step"], t["patience_counter"], t["stopped"]) for t in tr],
                         [t["dist"] for t in tr], int(torch.cuda.default_generators[0].get_offset())))
        res[graph] = runs
        if graph:
            assert len(eng._graphs) >= 1 and all(c.es is not None for c in eng._graphs.values())
    n = case["n_steps"] if case["n_steps"] is not None else h["NSteps"]
    for e, g in zip(res[False], res[True]):
        for r in (e, g):
            assert 1 <= r[2] <= n and np.isfinite(r[0]).all() and np.isfinite(r[1]).all()
            assert len(r[3]) in (0, r[2]) and [t[0] for t in r[3]] == list(range(1, len(r[3]) + 1))
            assert all(not t[2] for t in r[3][:-1])                  # only the last record may say "stopped"
        if rng == "torch":                       # same noise stream -> the very same bits, generator state included
            assert e[2] == g[2] and e[3] == g[3]
            np.testing.assert_allclose(e[4], g[4], rtol=1e-12)
            np.testing.assert_array_equal(e[0], g[0])
            np.testing.assert_array_equal(e[1], g[1])
            assert e[5] == g[5]
    if name == "ve_earlystop_run_all":
        assert all(r[2] == n for r in res[True])
    else:
        assert any(r[2] < n for r in res[True])


@pytest.mark.parametrize("shape", [(1, 4, 160, 160), (1, 4, 256, 256), (1, 3, 301, 301), (1, 4, 384, 384), (3, 4, 96, 96)],
                         ids=["fold_two_rows_per_thread", "fold_16B_lanes", "vec1_decide_kernel", "vec4_decide_kernel",
                              "fold_batch_rows"])
def test_early_stop_graph_equals_eager_at_every_launch_geometry(shape):
    """The gated (captured) loop against the watched (eager) one across launch geometries: 102 400 elements = 400 blocks,
    262 144 = 1 024 blocks at 4 B per lane, 271 803 (odd: 1 062 blocks), 589 824 (16 B per lane, 576 blocks), a batch of rows.
    Every block adds its sums into the iteration's 64-slot accumulator set; the gated loop applies the verdict at the top
    of its next launch at any of these sizes, the watched one in a one-wave kernel.  torch's stream on both sides, so:
    the same iteration count, the same trace, bitwise the same x / out over replays that stop on different iterations."""
    import torch
    from lanpaint_amd import LanPaint, pack_mask
    rng = np.random.default_rng(shape[-1])
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    mask = np.ones(shape, dtype=np.float32)
    h, w = shape[-2], shape[-1]
    mask[..., h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 0.0
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    n = 7
    res = {}
    for graph in (False, True):
        torch.manual_seed(77)
        eng = LanPaint(MODELS["linear_tuple"](), n, 15.0, 5.0, 1.0, 0.2, rng="torch", graph=graph)
        yg, ng, mg = tt(y), tt(noise), pack_mask(tt(mask))
        runs = []
        for rep, (sg, thr) in enumerate([(1.0, 0.3), (1.4, 0.45), (1.0, 1e-9), (0.8, 0.3)]):
            sigma = np.full((shape[0],), sg, dtype=np.float32)
            times = tuple(tt(t) for t in gc.times_from_sigma(sigma, False))
            mo = {"lanpaint_semantic_stop": {"threshold": thr, "patience": 1}}
            if rep != 1:                 # (without a trace the captured loop closes itself: LP_FL_ES_CLOSE)
                mo["lanpaint_semantic_trace"] = []
            x = tt(y + noise * np.float32(sg))
            it0 = eng.iterations_run
            out = eng(x, yg, ng, tt(sigma), mg, times, mo, 0)
            tr = mo.get("lanpaint_semantic_trace", [])
            runs.append((x.cpu().numpy(), out.cpu().numpy(), eng.iterations_run - it0,
                         [(t["patience_counter"], t["stopped"]) for t in tr], [t["dist"] for t in tr]))
        res[graph] = runs
        if graph:
            assert eng._graphs and all(c.es is not None for c in eng._graphs.values())
    for e, g in zip(res[False], res[True]):
        assert e[2] == g[2] and e[3] == g[3]
        np.testing.assert_allclose(e[4], g[4], rtol=1e-6)      # (double accumulators filled by atomics: order-exact to ~1e-16)
        np.testing.assert_array_equal(e[0], g[0])
        np.testing.assert_array_equal(e[1], g[1])
    ran = [r[2] for r in res[True]]
    assert ran[2] == n and min(ran) < n, ran            # the 1e-9 threshold never stops; the others stop early


def test_early_stop_graph_equals_eager_with_bf16_backbone_input_and_fused_cfg_heads():
    """The gated loop next to the other launch options: the backbone input emitted as bf16 (`model_dtype`: stopped
    launches must re-emit it, the last one the fp32 x that is written back) and CFG heads combined in the kernel."""
    import torch
    from lanpaint_amd import FusedCFGHeads, LanPaint

    class Heads(MODELS["linear_tuple"]):
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            xf = x.float()
            return FusedCFGHeads(0.9 * xf + 0.05, 0.7 * xf - 0.1, 4.0, -0.5)

    case = gc.build_case("ve_earlystop")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    res = {}
    for graph in (False, True):
        torch.manual_seed(99)
        eng = LanPaint(Heads(), 10, 15.0, 5.0, 1.0, 0.2, rng="torch", graph=graph, model_dtype=torch.bfloat16,
                       EarlyStopThreshold=20.0, EarlyStopPatience=2)
        y, noise, mask = tt(case["y"]), tt(case["noise"]), tt(case["mask"])
        runs = []
        for rep in range(3):
            x = tt(case["x"] + np.float32(0.03 * rep))
            it0 = eng.iterations_run
            out = eng(x, y, noise, tt(case["sigma"]), mask, tuple(tt(t) for t in case["times"]), {}, 0)
            runs.append((x.cpu().numpy(), out.cpu().numpy(), eng.iterations_run - it0))
        res[graph] = runs
    assert any(r[2] < 10 for r in res[False])
    for e, g in zip(res[False], res[True]):
        assert e[2] == g[2]
        np.testing.assert_array_equal(e[0], g[0])
        np.testing.assert_array_equal(e[1], g[1])


ES_ORACLE_CASES = {
    # name: (shape, flow, sigma per row, mask kind, threshold, patience, n_steps, pack bits)
    "rows2_ve":      ((2, 4, 12, 12), False, [1.4, 0.8], "blob", 0.45, 1, 8, False),
    "video5d_flow":  ((1, 4, 3, 6, 10), True, [0.55], "box", 0.2, 1, 8, False),
    "soft_ve":       ((1, 4, 10, 10), False, [1.0], "soft", 0.5, 1, 8, False),
    "odd_numel":     ((1, 3, 5, 7), False, [1.2], "box", 0.45, 2, 9, False),
    "vec4_f32":      ((1, 4, 384, 384), False, [1.0], "blob", 0.3, 1, 6, False),
    "vec4_bits":     ((1, 4, 384, 384), False, [1.0], "blob", 0.3, 1, 6, True),
    "never_enabled": ((1, 4, 8, 8), False, [1.0], "ones", 0.5, 1, 4, False),       # no inpaint weight: stopper off (:115-117)
}


@pytest.mark.parametrize("name", sorted(ES_ORACLE_CASES))
def test_device_side_early_stop_matches_the_oracle_stopper(name):
    """The stop rule evaluated on the device (LP_FL_ES) against the oracle's restatement of earlystop.py:58-336 on
    shapes the reference-generated goldens do not cover: batch rows with their own sigma (abt mean over rows), a 5-D
    latent (no ring), a soft mask (general arithmetic path), the 16 B/lane kernels with fp32 and bit-packed masks,
    and a mask without inpaint region (stopper disabled).  Same recorded xi stream on both sides."""
    import torch
    from lanpaint_amd import LanPaint, pack_mask
    shape, flow, sig, mkind, thr, pat, n, bits = ES_ORACLE_CASES[name]
    rng = np.random.default_rng(sorted(ES_ORACLE_CASES).index(name) + 100)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    sigma = np.float32(sig)
    sb = sigma.reshape((-1,) + (1,) * (len(shape) - 1))
    x = ((sb * noise + (1 - sb) * y) if flow else (y + noise * sb)).astype(np.float32)
    if mkind == "soft":
        mask = rng.random(shape, dtype=np.float32)
    elif mkind == "ones":
        mask = np.ones(shape, dtype=np.float32)
    elif mkind == "blob":
        mask = np.ones(shape, dtype=np.float32)
        h, w = shape[-2], shape[-1]
        mask[..., h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 0.0
    else:
        mask = gc.box_mask(shape)
    times = gc.times_from_sigma(sigma, flow)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(2 * n)]
    mo = {"lanpaint_semantic_stop": {"threshold": thr, "patience": pat}}
    it_o = iter(draws)
    o = OracleLanPaint(MODELS["linear_tuple"](flow=flow), n, 15.0, 5.0, 1.0, 0.2, is_flow=flow, randn=lambda like: next(it_o))
    xo = x.copy()
    out_o = o(xo, y, noise, sigma, mask, times, dict(mo), 0)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    it_g = iter([tt(d) for d in draws])
    model = MODELS["linear_tuple"](flow=flow)
    eng = LanPaint(model, n, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, rng=lambda like: next(it_g))
    mg = tt(mask)
    if bits:
        mg = pack_mask(mg)
    mo_g = dict(mo, lanpaint_semantic_trace=[])
    xg = tt(x)
    out_g = eng(xg, tt(y), tt(noise), tt(sigma), mg, tuple(tt(t) for t in times), mo_g, 0)
    torch.cuda.synchronize()
    tr_o = o.last_stopper.trace if o.last_stopper is not None else []
    tr_g = mo_g["lanpaint_semantic_trace"]
    assert eng.iterations_run == o.iterations_run and model.calls == o.iterations_run + 1
    assert len(tr_g) == len(tr_o)
    assert [t["patience_counter"] for t in tr_g] == [t["counter"] for t in tr_o]
    assert [t["stopped"] for t in tr_g] == [t["stopped"] for t in tr_o]
    np.testing.assert_allclose([t["dist"] for t in tr_g], [t["dist"] for t in tr_o], rtol=2e-4)
    assert sum(1 for _ in it_g) == sum(1 for _ in it_o)             # the same number of draws consumed
    assert_close(xg.cpu().numpy(), xo, f"{name}: x", rel=5e-5)
    assert_close(out_g.cpu().numpy(), out_o, f"{name}: out", rel=5e-5)
    if name == "never_enabled":
        assert tr_g == [] and eng.iterations_run == n
    elif name == "odd_numel":
        assert eng.iterations_run == n and any(t["dist_drift"] is not None for t in tr_g)   # the drift anchor vetoes the stop
    else:
        assert eng.iterations_run < n                                # every other case does stop early


# ------------------------------------------------------------------ CFG combination fused into the step kernel
@pytest.mark.parametrize("name,dtype", [("ve_basic", "float32"), ("flow_batch", "float32"), ("ve_odd_numel", "float32"),
                                        ("ve_basic", "bfloat16")])
def test_fused_cfg_heads_equal_eager_cfg(name, dtype):
    """A backbone handing back FusedCFGHeads(cond, uncond, s, s_BIG) gives the same trajectory as one
    that forms `uncond + (cond - uncond) * scale` eagerly (reference nodes.py:161-175)."""
    import torch
    from lanpaint_amd import FusedCFGHeads
    dt = getattr(torch, dtype)

    class Eager(MODELS["linear_tuple"]):
        def preds(self, x):
            return (0.9 * x + 0.05).to(dt), (0.7 * x - 0.1).to(dt)

        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            c, u = self.preds(x)
            return u + (c - u) * 5.0, u + (c - u) * -0.5

    class Fused(Eager):
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            c, u = self.preds(x)
            return FusedCFGHeads(c, u, 5.0, -0.5)

    a = run_product_case(name, model_cls=Eager)
    b = run_product_case(name, model_cls=Fused)
    assert a["model"].calls == b["model"].calls
    rel = 2e-5 if dtype == "float32" else 3e-2            # bf16: eager rounds the heads to bf16, fused keeps fp32
    assert_close(b["x"], a["x"], f"{name}: fused-CFG x", rel=rel, mse=1e-9 if dtype == "float32" else 1e-4)
    assert_close(b["out"], a["out"], f"{name}: fused-CFG out", rel=rel, mse=1e-9 if dtype == "float32" else 1e-4)


@pytest.mark.parametrize("name", ["ve_basic", "flow_batch", "ve_sdxl_full"])
def test_fused_cfg_heads_match_the_oracle_fed_the_reference_cfg_combination(name):
    """Direct check (no HIP on the expected side): the CPU oracle -- pinned to the reference -- runs with a backbone
    that forms both CFG heads the way the reference's sampling_function_LanPaint does through ComfyUI's stock
    cfg_function, `uncond + (cond - uncond) * scale` twice (nodes.py:161-175); the HIP engine gets the same cond /
    uncond predictions as FusedCFGHeads and combines them inside the step and finalise kernels."""
    from lanpaint_amd import FusedCFGHeads
    s0, s1 = 4.5, -0.5

    class Heads(MODELS["linear_tuple"]):
        def preds(self, x):
            return 0.9 * x + 0.05, 0.7 * x - 0.1                    # cond, uncond (operators only: numpy and torch)

    class RefCFG(Heads):                                             # what the reference's model function returns
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            c, u = self.preds(x)
            return u + (c - u) * s0, u + (c - u) * s1

    class Fused(Heads):
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            c, u = self.preds(x)
            return FusedCFGHeads(c, u, s0, s1)

    case = gc.build_case(name)
    g = load_golden(name)
    it = iter(xi_list(g))
    h = case["hyper"]
    omodel = RefCFG(flow=case["flow"] or case["flux"])
    o = OracleLanPaint(omodel, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], is_flux=case["flux"],
                       is_flow=case["flow"], min_step_frac=h["MinStepFrac"], randn=lambda like: next(it))
    xo = case["x"].copy()
    out_o = o(xo, case["y"], case["noise"], case["sigma"], case["mask"], case["times"], None, 0, n_steps=case["n_steps"])
    r = run_product_case(name, model_cls=Fused)
    assert r["model"].calls == omodel.calls and r["leftover"] == 0
    assert_close(r["x"], xo, f"{name}: fused-CFG x vs oracle", rel=3e-5)
    assert_close(r["out"], out_o, f"{name}: fused-CFG out vs oracle", rel=3e-5)


def test_engine_in_front_of_dummy_unet_matches_oracle():
    """A real nn.Module backbone (random-init SD1.5-shaped UNet, fp32 here) with a dual-head output:
    HIP engine vs CPU oracle driving the SAME network on the same xi stream."""
    import torch
    from lanpaint_amd import LanPaint
    from oracle.lanpaint_oracle import OracleLanPaint, times_from_sigma
    from tests.dummy_unet import DummyUNetBackbone
    dev = "cuda"
    net = DummyUNetBackbone(dev, dtype=torch.float32)
    rng = np.random.default_rng(4)
    shape = (1, 4, 16, 16)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    sigma = np.float32([1.2])
    x = (y + noise * sigma[0]).astype(np.float32)
    mask = gc.box_mask(shape)
    times = times_from_sigma(sigma, False)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(5)]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

    def np_model(xn, t, model_options=None, seed=None):      # the oracle calls the same GPU network
        a, b = net(tt(xn.astype(np.float32)), tt(np.asarray(t, dtype=np.float32)))
        return a.cpu().numpy(), b.cpu().numpy()

    it = iter(draws)
    o = OracleLanPaint(np_model, 3, 15.0, 5.0, 1.0, 0.2, randn=lambda like: next(it))
    xo = x.copy()
    out_o = o(xo, y, noise, sigma, mask, times, None, 0)
    it2 = iter([tt(d) for d in draws])
    eng = LanPaint(net, 3, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it2))
    xg = tt(x)
    out_g = eng(xg, tt(y), tt(noise), tt(sigma), tt(mask), tuple(tt(t) for t in times), None, 0)
    assert_close(xg.cpu().numpy(), xo, "dummy-UNet x", rel=1e-4, mse=1e-8)
    assert_close(out_g.cpu().numpy(), out_o, "dummy-UNet out", rel=1e-4, mse=1e-8)


def test_model_dtype_bf16_emits_half_precision_inputs():
    """model_dtype=bfloat16: the kernel emits the backbone input in bf16 (no cast pass); state, the
    written-back x and the arithmetic stay fp32."""
    import torch
    seen = []

    class Spy(MODELS["linear_tuple"]):
        def __call__(self, x, t, model_options=None, seed=None):
            seen.append(x.dtype)
            return super().__call__(x.float(), t)

    a = run_product_case("ve_basic")
    b = run_product_case("ve_basic", model_cls=Spy, model_dtype=torch.bfloat16)
    assert seen and all(dt == torch.bfloat16 for dt in seen)
    assert b["x"].dtype == np.float32
    assert float(np.mean((b["out"] - a["out"]) ** 2)) < 1e-3 and float(np.abs(b["x"] - a["x"]).max()) < 0.15
    c = run_product_case("ve_n0", model_cls=Spy, model_dtype=torch.float16)       # n_steps = 0: only the final emit
    g = c["golden"]
    assert_close(c["x"], g["x_out"], "fp16 model_dtype keeps the written-back x in fp32")


def test_uint8_mask_attachment_is_bitwise_equivalent():
    """A binary mask may travel as uint8 (KSamplerX0Inpaint attaches it): identical results, 1 B/element."""
    import torch
    from lanpaint_amd import LanPaint
    for name in ("ve_basic", "ve_odd_numel", "flow_video5d"):
        case = gc.build_case(name)
        g = load_golden(name)
        outs = []
        for attach in (False, True):
            it = iter([torch.from_numpy(d).cuda() for d in xi_list(g)])
            model = MODELS[case["model"]](flow=case["flow"])
            eng = LanPaint(model, 5, 15.0, 5.0, 1.0, 0.2, IS_FLOW=case["flow"], rng=lambda like: next(it))
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
            mask = tt(case["mask"])
            if attach:
                mask._lp_u8 = mask.to(torch.uint8)
            x = tt(case["x"].copy())
            out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), mask, tuple(tt(t) for t in case["times"]), None, 0)
            outs.append((x.cpu().numpy(), out.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert_close(outs[1][0], g["x_out"], f"{name}: u8-mask x")


def test_bit_packed_mask_is_bitwise_equivalent():
    """pack_mask attaches the 1-bit/element form (LP_FL_MASK_BITS): identical results on golden cases (VEC=1 path)."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import LanPaint
    for name in ("ve_basic", "ve_odd_numel", "flow_video5d", "ve_batch_rows"):
        if name not in gc.CASES:
            continue
        case = gc.build_case(name)
        g = load_golden(name)
        outs = []
        for attach in (False, True):
            it = iter([torch.from_numpy(d).cuda() for d in xi_list(g)])
            model = MODELS[case["model"]](flow=case["flow"])
            eng = LanPaint(model, 5, 15.0, 5.0, 1.0, 0.2, IS_FLOW=case["flow"], rng=lambda like: next(it))
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
            mask = tt(case["mask"])
            if attach:
                mask = lanpaint_amd.pack_mask(mask)
                assert hasattr(mask, "_lp_bits")
            x = tt(case["x"].copy())
            out = eng(x, tt(case["y"]), tt(case["noise"]), tt(case["sigma"]), mask, tuple(tt(t) for t in case["times"]), None, 0)
            outs.append((x.cpu().numpy(), out.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert_close(outs[1][0], g["x_out"], f"{name}: bit-mask x")


def test_bit_packed_mask_vec4_rows_not_word_aligned():
    """Float4 path (n_el > 512 Ki) with rows whose length is a multiple of 4 but not of 32: every row starts in
    the middle of a mask word.  Philox noise, fp32 / u8 / bit masks must agree bitwise, eager and under graph replay."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import LanPaint
    torch.manual_seed(5)
    shape = (2, 4, 258, 262)
    assert (shape[1] * shape[2] * shape[3]) % 32 == 16
    y = torch.randn(shape, device="cuda")
    noise = torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sig = torch.tensor([3.0, 1.5], device="cuda")
    times = (sig.clone(), 1 / (1 + sig ** 2), torch.sqrt(1 - 1 / (1 + sig ** 2)))
    res = []
    for kind in ("f32", "u8", "bits", "f32_graph", "bits_graph"):
        m = mask.clone()
        if kind == "u8":
            m._lp_u8 = m.to(torch.uint8)
        elif kind.startswith("bits"):
            m = lanpaint_amd.pack_mask(m)
        eng = LanPaint(MODELS["linear_tuple"](flow=False), 4, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=11,
                       graph=kind.endswith("graph"))
        x = (y + noise * 3.0).clone()
        out = eng(x, y, noise, sig, m, times, {}, 0)
        res.append((x.cpu(), out.cpu()))
    for r in res[1:3]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
    # a replayed graph draws its Philox sequence numbers from the device-side counter: compare graph with graph
    assert torch.equal(res[4][0], res[3][0]) and torch.equal(res[4][1], res[3][1])


@pytest.mark.parametrize("shape", [(2, 4, 40, 36), (2, 4, 258, 262)], ids=["vec1", "vec4"])
@pytest.mark.parametrize("flow", [False, True], ids=["ve", "flow"])
def test_kernel_instantiation_matrix_is_consistent(shape, flow):
    """Every hot instantiation of the step kernel (mask fp32 / uint8 / bits x backbone output fp32 / bf16 / fp16 x
    plain / fused-CFG heads, both vector widths): for one output dtype all mask formats are BITWISE equal, the
    16-bit variants stay within their storage precision of the fp32 run, and fused CFG equals CFG formed eagerly."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import FusedCFGHeads, LanPaint
    torch.manual_seed(3)
    y = torch.randn(shape, device="cuda")
    noise = torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    if flow:
        sig = torch.tensor([0.8, 0.45], device="cuda")
        abt = (1 - sig) ** 2 / ((1 - sig) ** 2 + sig ** 2)
        times = (sig / (1 - sig), abt, sig.clone())
        x_start = sig.view(-1, 1, 1, 1) * noise + (1 - sig.view(-1, 1, 1, 1)) * y
    else:
        sig = torch.tensor([3.0, 0.9], device="cuda")
        abt = 1 / (1 + sig ** 2)
        times = (sig.clone(), abt, torch.sqrt(1 - abt) / (torch.sqrt(1 - abt) + torch.sqrt(abt)))
        x_start = y + noise * sig.view(-1, 1, 1, 1)

    class Net:
        """two 'heads' with a little structure; `dtype` selects the storage type of the outputs."""
        def __init__(self, dtype, fused):
            self.inner_model = self
            self.model_sampling = (MODELS["linear_tuple"](flow=flow)).inner_model.model_sampling
            self.dtype, self.fused = dtype, fused

        def __call__(self, x, t, model_options=None, seed=None):
            cond, uncond = (0.9 * x + 0.05).to(self.dtype), (0.7 * x - 0.02).to(self.dtype)
            if self.fused:
                return FusedCFGHeads(cond, uncond, 3.0, 5.0)
            c, u = cond.float(), uncond.float()
            return (u + (c - u) * 3.0).to(self.dtype), (u + (c - u) * 5.0).to(self.dtype)

    def run(mask_kind, dtype, fused):
        m = mask.clone()
        if mask_kind == "u8":
            m._lp_u8 = m.to(torch.uint8)
        elif mask_kind == "bits":
            m = lanpaint_amd.pack_mask(m)
        eng = LanPaint(Net(dtype, fused), 4, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, rng="philox", philox_seed=5)
        x = x_start.clone()
        out = eng(x, y, noise, sig, m, times, {}, 0)
        assert torch.isfinite(x).all() and torch.isfinite(out).all()
        return x, out

    ref = {}
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for fused in (False, True):
            base = run("f32", dtype, fused)
            for mk in ("u8", "bits"):
                other = run(mk, dtype, fused)
                assert torch.equal(other[0], base[0]) and torch.equal(other[1], base[1]), (mk, dtype, fused)
            ref[(dtype, fused)] = base
    f32 = ref[(torch.float32, False)]
    scale = float(f32[0].abs().max())
    # CFG formed inside the kernel (fp32, fmaf) vs formed eagerly and stored: fp32 identical up to rounding
    assert float((ref[(torch.float32, True)][0] - f32[0]).abs().max()) <= 2e-5 * scale
    for dtype, tol in ((torch.bfloat16, 6e-2), (torch.float16, 8e-3)):
        for fused in (False, True):
            assert float((ref[(dtype, fused)][0] - f32[0]).abs().max()) <= tol * scale, (dtype, fused)


@pytest.mark.parametrize("shape", [(2, 4, 24, 20), (2, 4, 258, 262), (1, 4, 521, 301), (2, 4, 520, 520), (1, 4, 800, 800),
                                   (2, 16, 21, 60, 104), (3, 4, 548, 548), (5, 2, 729, 729)],
                         ids=["vec1", "vec4", "strided", "strided_rows", "strided_rounds", "strided_video_batch",
                              "strided_three_rows", "five_rows_4B_lanes"])
@pytest.mark.parametrize("mode", ["eager", "graph", "early_stop"])
def test_in_kernel_torch_stream_equals_randn_like_tensors(shape, mode):
    """rng="torch" generates the reference's noise (torch.randn_like(x_t), POST draw then PRE draw) inside the step
    kernel.  Against rng="torch-eager" (the same draws as explicit tensors): bitwise the same x / out over several
    sigma calls, and the device generator ends in the same state -- so whatever draws from it next (an ancestral
    sampler's noise) is unchanged too."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import LanPaint
    torch.manual_seed(0)
    y = torch.randn(shape, device="cuda")
    noise = torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sigmas = [torch.tensor([3.0, 2.0, 2.5, 1.7, 2.2], device="cuda"), torch.tensor([1.2, 0.8, 1.0, 0.9, 1.1], device="cuda"),
              torch.tensor([0.5, 0.3, 0.4, 0.35, 0.45], device="cuda")]
    sigmas = [s_[:shape[0]] for s_ in sigmas]       # "strided": one batch row, 627 k elements > ATen's block * grid;
    # "strided_rows" / "_video_batch": two rows whose boundary cuts through a round of ATen's grid-stride loop (2.16 M and
    # 4.19 M elements, two rounds), "strided_rounds": one row of 2.56 M elements = two rounds, "strided_three_rows": 3.6 M
    # elements in rows of 1.2 M (every round holds parts of two or three rows); "five_rows_4B_lanes": rows of 1 062 882
    # elements (not a multiple of 4: the 4 B-per-lane kernels, one Philox block per element, 5.3 M elements = 3 rounds)
    kw = dict(graph=True) if mode == "graph" else dict(EarlyStopThreshold=1e-7, EarlyStopPatience=1) if mode == "early_stop" else {}
    res = {}
    for rng in ("torch-eager", "torch"):
        eng = LanPaint(MODELS["linear_tuple"](flow=False), 4, 15.0, 5.0, 1.0, 0.2, rng=rng, **kw)
        torch.manual_seed(77)
        x = (y + noise * 3.0).clone()
        mask_in = lanpaint_amd.pack_mask(mask.clone()) if shape[0] == 1 else mask      # also the hard-mask kernels
        outs = []
        for rep in range(2):                    # second pass: replays (graph mode), generator keeps moving
            for sig in sigmas:
                abt = 1 / (1 + sig ** 2)
                times = (sig.clone(), abt, torch.sqrt(1 - abt) / (torch.sqrt(1 - abt) + torch.sqrt(abt)))
                out = eng(x, y, noise, sig, mask_in, times, {}, 0)
                between = torch.randn(3, device="cuda")          # somebody else draws between the calls
                outs.append((x.clone(), out.clone(), between))
        res[rng] = (outs, torch.cuda.default_generators[0].get_offset(), eng.iterations_run)
    a, b = res["torch-eager"], res["torch"]
    assert a[1] == b[1] and a[2] == b[2]
    for (xa, oa, ba), (xb, ob, bb) in zip(a[0], b[0]):
        assert torch.equal(xa, xb) and torch.equal(oa, ob) and torch.equal(ba, bb)


def test_in_kernel_torch_stream_steps_aside_when_the_backbone_draws_inside_the_graph():
    """A backbone that consumes torch's generator inside the loop cannot be replayed with one published offset:
    the engine stays eager for it (and still matches the explicit-tensor path bit for bit)."""
    import torch
    from lanpaint_amd import LanPaint
    base = MODELS["linear_tuple"]

    class Noisy(base):
        def __call__(self, x, t, model_options=None, seed=None):
            a, b = super().__call__(x, t)
            return a + 1e-3 * torch.randn_like(a), b

    shape = (1, 4, 16, 16)
    torch.manual_seed(0)
    y, noise = torch.randn(shape, device="cuda"), torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sig = torch.tensor([2.0], device="cuda")
    abt = 1 / (1 + sig ** 2)
    times = (sig.clone(), abt, torch.sqrt(1 - abt) / (torch.sqrt(1 - abt) + torch.sqrt(abt)))
    res = []
    for rng, graph in (("torch-eager", False), ("torch", True)):
        eng = LanPaint(Noisy(flow=False), 3, 15.0, 5.0, 1.0, 0.2, rng=rng, graph=graph)
        torch.manual_seed(5)
        x = (y + noise * 2.0).clone()
        outs = [eng(x, y, noise, sig, mask, times, {}, 0).clone() for _ in range(3)]
        res.append((x.clone(), outs, torch.cuda.default_generators[0].get_offset()))
        if graph:
            assert eng._graph_blocked and not eng._graphs
    assert torch.equal(res[0][0], res[1][0]) and res[0][2] == res[1][2]
    for u, v in zip(res[0][1], res[1][1]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("name", ["av_flat_pack", "av_flat_binary", "ve_soft_mask", "ve_earlystop", "flow_video5d", "ve_batch_rows"])
def test_in_kernel_torch_stream_on_the_general_paths(name):
    """rng="torch" vs rng="torch-eager" on the golden cases that take the run-time kernels (per-element AV times,
    soft masks, semantic early stop) and on 5-D / multi-row latents: bitwise the same x / out, same generator state."""
    import torch
    if name not in gc.CASES:
        pytest.skip(f"no golden case {name}")
    res = []
    for rng in ("torch-eager", "torch"):
        torch.manual_seed(2024)
        r = run_product_case(name, rng=rng)
        res.append((r["x"], r["out"], torch.cuda.default_generators[0].get_offset(), r["engine"].iterations_run))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]


# ------------------------------------------------------------------ every instantiation the library holds is launched
# (scripts/instantiation_coverage.py; tests/test_cabi_exports.py checks the committed coverage file against the library)
@pytest.mark.parametrize("shape", [(2, 4, 24, 20), (2, 4, 258, 262), (1, 4, 521, 301), (2, 4, 520, 520), (1, 4, 384, 384)],
                         ids=["vec1", "vec4", "strided", "strided_rows", "vec4_one_row"])
@pytest.mark.parametrize("mask_fmt", ["f32", "bits"])
@pytest.mark.parametrize("heads", ["f32", "bf16"])
def test_torch_stream_variants_with_half_heads_and_either_mask_format(shape, mask_fmt, heads):
    """The in-kernel torch stream (RNG = 1) against explicit torch.randn_like tensors for the storage formats production
    uses together with it: bf16 backbone heads (X0W = 2) and fp32 / bit-packed masks (MODE 0 / 2), one element and four per
    lane, ATen-strided lanes and not -- bitwise the same x / out, same generator offset."""
    import torch
    import lanpaint_amd
    from lanpaint_amd import LanPaint
    dtype = torch.bfloat16 if heads == "bf16" else torch.float32

    class Net:
        def __init__(self):
            self.inner_model = self
            self.model_sampling = MODELS["linear_tuple"](flow=False).inner_model.model_sampling

        def __call__(self, x, t, model_options=None, seed=None):
            return (0.9 * x + 0.05).to(dtype), (0.7 * x - 0.02).to(dtype)

    torch.manual_seed(0)
    y = torch.randn(shape, device="cuda")
    noise = torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sig = torch.tensor([1.2, 0.8], device="cuda")[:shape[0]]
    abt = 1 / (1 + sig ** 2)
    times = (sig.clone(), abt, torch.sqrt(1 - abt) / (torch.sqrt(1 - abt) + torch.sqrt(abt)))
    res = {}
    for rng in ("torch-eager", "torch"):
        eng = LanPaint(Net(), 3, 15.0, 5.0, 1.0, 0.2, rng=rng, graph=False)
        torch.manual_seed(5)
        m = lanpaint_amd.pack_mask(mask.clone()) if mask_fmt == "bits" else mask
        x = (y + noise * 1.2).clone()
        out = eng(x, y, noise, sig, m, times, {}, 0)
        res[rng] = (x, out, torch.cuda.default_generators[0].get_offset())
    a, b = res["torch-eager"], res["torch"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]


@pytest.mark.parametrize("rng", ["torch", "philox"])
@pytest.mark.parametrize("mask_fmt", ["bits", "f32"])
def test_early_stop_on_a_streaming_size_latent_in_both_loop_forms(rng, mask_fmt):
    """The inner early stop at 16 B per lane (589 824 elements), watched (eager) and gated (replayed), with either
    generator and either mask format: with torch's stream the two loop forms agree bit for bit; Philox streams differ
    between the forms by construction, there both must run, stop (or not) on a sensible iteration and stay finite."""
    import torch
    from lanpaint_amd import LanPaint, pack_mask
    shape, n = (1, 4, 384, 384), 6
    g = np.random.default_rng(9)
    y = g.standard_normal(shape, dtype=np.float32)
    noise = g.standard_normal(shape, dtype=np.float32)
    mask = np.ones(shape, dtype=np.float32)
    mask[..., 96:288, 96:288] = 0.0
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    res = {}
    for graph in (False, True):
        torch.manual_seed(41)
        eng = LanPaint(MODELS["linear_tuple"](), n, 15.0, 5.0, 1.0, 0.2, rng=rng, philox_seed=3, graph=graph)
        yg, ng = tt(y), tt(noise)
        mg = pack_mask(tt(mask)) if mask_fmt == "bits" else tt(mask)
        runs = []
        for sg, thr in [(1.0, 0.3), (1.0, 1e-9), (1.4, 0.45)]:
            sigma = np.full((1,), sg, dtype=np.float32)
            times = tuple(tt(t) for t in gc.times_from_sigma(sigma, False))
            mo = {"lanpaint_semantic_stop": {"threshold": thr, "patience": 1}, "lanpaint_semantic_trace": []}
            x = tt(y + noise * np.float32(sg))
            it0 = eng.iterations_run
            out = eng(x, yg, ng, tt(sigma), mg, times, mo, 0)
            runs.append((x.cpu().numpy(), out.cpu().numpy(), eng.iterations_run - it0,
                         [t["dist"] for t in mo["lanpaint_semantic_trace"]]))
        res[graph] = runs
    for e, gr in zip(res[False], res[True]):
        assert np.isfinite(e[0]).all() and np.isfinite(gr[0]).all() and 1 <= e[2] <= n and 1 <= gr[2] <= n
        if rng == "torch":
            assert e[2] == gr[2]
            np.testing.assert_allclose(e[3], gr[3], rtol=1e-6)
            np.testing.assert_array_equal(e[0], gr[0])
            np.testing.assert_array_equal(e[1], gr[1])
    assert res[False][1][2] == n and res[True][1][2] == n          # the 1e-9 threshold never stops
    assert res[True][0][2] < n                                     # 0.3 does


def test_per_element_times_on_a_streaming_size_pack():
    """Per-element times (MiniMax-H3 style AV pack, lanpaint.py:60-74) on a pack large enough for 16 B per lane: in-kernel
    torch stream against explicit randn_like tensors, bitwise."""
    import torch
    from lanpaint_amd import LanPaint
    shape = (1, 1, 655360)
    torch.manual_seed(1)
    y = torch.randn(shape, device="cuda")
    noise = torch.randn(shape, device="cuda")
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    ai = torch.zeros(shape, device="cuda")
    ai[..., 400000:] = 1.0
    flow_t, flow_a = torch.tensor([0.6], device="cuda"), torch.tensor([0.35], device="cuda")
    tm = lambda t: (t / (1 - t), (1 - t) ** 2 / ((1 - t) ** 2 + t ** 2), t)      # noqa: E731
    res = []
    for rng in ("torch-eager", "torch"):
        torch.manual_seed(8)
        eng = LanPaint(MODELS["linear_tuple"](flow=True), 3, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, rng=rng, graph=False)
        x = (0.6 * noise + 0.4 * y).clone()
        out = eng(x, y, noise, flow_t, mask, tm(flow_t), {}, 0, current_times_audio=tm(flow_a), audio_indicator=ai,
                  audio_correction=(1.0 - ai) + 0.8 * ai)
        res.append((x, out, torch.cuda.default_generators[0].get_offset()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    assert torch.isfinite(res[0][0]).all()


def test_armed_early_stop_that_can_never_fire_runs_the_plain_loop():
    """Nothing to inpaint (mask all known): the reference's LanPaintEarlyStopper.from_options returns None
    (earlystop.py:115-117) and the plain loop runs.  The watched loop learns the same from the first verdict (enabled = 0)
    and goes on with the plain fused launches: same draws, same result as an engine without the stopper, full count."""
    import torch
    from lanpaint_amd import LanPaint
    shape, n = (1, 4, 24, 24), 5
    g = np.random.default_rng(5)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = torch.ones(shape, device="cuda")
    s = torch.full((1,), 1.1, device="cuda")
    times = gc.times_from_sigma(s, False)
    res = []
    for thr in (0.0, 0.3):
        torch.manual_seed(12)
        m = MODELS["linear_tuple"]()
        eng = LanPaint(m, n, 15.0, 5.0, 1.0, 0.2, EarlyStopThreshold=thr, EarlyStopPatience=1, rng="torch", graph=False)
        x = (y + noise * 1.1).clone()
        mo = {"lanpaint_semantic_trace": []}
        out = eng(x, y, noise, s, mask, times, mo, 0)
        torch.cuda.synchronize()
        res.append((x.clone(), out.clone(), eng.iterations_run, m.calls, torch.cuda.default_generators[0].get_offset(),
                    len(mo["lanpaint_semantic_trace"])))
    assert res[0][2:] == res[1][2:] == (n, n + 1, res[0][4], 0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
