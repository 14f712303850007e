"""The op-for-op port of the reference engine ON the MI355X as the same-seed arbiter.

The reference is a Python package and does not travel to the GPU box in any form (no source, no bytecode).  What runs here is
`oracle/lanpaint_oracle.py` -- the CPU restatement pinned to reference-generated fixtures (tests/test_oracle_golden.py, all
array backends; the early-stop trace records included) -- handed DEVICE tensors through its torch backend: the reference's
eager ATen call sequence (~164 launches per think iteration), drawing its noise with `torch.randn_like` from the device
generator in the reference's order.  Next to it the product engine built with NO optional keyword (`rng="torch"`,
`graph="auto"`, the reference's fp32 mask), both started from the same `torch.manual_seed`: BASELINE.json's "identical (seed,
latent, mask, sigmas)".  No recorded xi stream sits in between; every sigma call's returned `out` and in-place `x` are
compared directly, and the device generator must end in the same state (the product generates the draws inside its step
kernel and advances the generator by what they consume).

Tolerance (fp32, stated): max-abs <= 2e-5 * max(1, |ref|_inf) and MSE <= 1e-9 * max(1, |ref|_inf)^2 per tensor -- four orders
inside BASELINE's MSE < 1e-5."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.stubs import FlowSampling, VESampling      # noqa: E402

REL, MSE = 2e-5, 1e-9


class TwoHeads:
    """x -> (0.9 x, 0.8 x) (SURVEY.md 8d's stub), plain tensor operators: the same object type serves both engines."""

    def __init__(self, flow):
        self.inner_model = self
        self.model_sampling = FlowSampling() if flow else VESampling()
        self.calls = 0

    def __call__(self, x, t, model_options=None, seed=None):
        self.calls += 1
        return 0.9 * x, 0.8 * x


class PortOnDevice:
    """The port behind the reference's constructor / call signature (lanpaint.py:8,44): the replace step goes through the
    model_sampling's own noise_scaling like the reference's does (lanpaint.py:84-92)."""

    def __init__(self, Model, NSteps, Friction, Lambda, Beta, StepSize, IS_FLUX=False, IS_FLOW=False, EarlyStopThreshold=0.0,
                 EarlyStopPatience=1, EarlyStopHook=None, MinStepFrac=0.0):
        from oracle.lanpaint_oracle import OracleLanPaint, TorchBackend
        ms = Model.inner_model.model_sampling
        self.inner_model = Model
        self.port = OracleLanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize, is_flux=IS_FLUX, is_flow=IS_FLOW,
                                   early_stop_threshold=EarlyStopThreshold, early_stop_patience=EarlyStopPatience,
                                   min_step_frac=MinStepFrac, backend=TorchBackend(), noise_scaling=ms.noise_scaling,
                                   noise_scale=float(getattr(ms, "noise_scale", 1.0)))

    def __call__(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, n_steps=None, **kw):
        return self.port(x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, n_steps=n_steps, **kw)


def _reference_class():
    return PortOnDevice


def _job(workload, row_ramp=None, n_sig=None):
    """bench.py's synthetic job for `workload`; `row_ramp` gives every batch row its own sigma ramp (per-row times)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    shape, flow, ns, n_think = bench.WORKLOADS[workload]
    sig_np = bench.flow_sigmas(ns) if flow else bench.karras_sigmas(ns)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
    ramp = np.ones(shape[0], np.float32) if row_ramp is None else np.asarray(row_ramp, np.float32)
    sig_list = [tt(np.float32(s) * ramp) for s in sig_np]
    if row_ramp is not None:
        r = tt(ramp).reshape((-1,) + (1,) * (len(shape) - 1))
        s0 = float(sig_np[0])
        x0 = (s0 * r * noise + (1 - s0 * r) * y) if flow else (y + noise * (s0 * r))
    times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
    ratios = bench.euler_ratios(sig_list, len(shape))
    n_sig = ns if n_sig is None else n_sig
    return dict(x0=x0, y=y, noise=noise, mask=mask, sig_list=sig_list[:n_sig], times_list=times_list[:n_sig], ratios=ratios,
                n_think=n_think, flow=flow, shape=shape)


def _walk(engine, job, seed, inference=False):
    """The sigma schedule with k-diffusion's Euler update between calls; returns every call's (out, x after the call) and
    the device generator's state at the end."""
    import torch
    torch.manual_seed(seed)
    x = job["x0"].clone()
    outs, xs = [], []
    ctx = torch.inference_mode() if inference else torch.no_grad()
    with ctx:
        for i, (s, t) in enumerate(zip(job["sig_list"], job["times_list"])):
            den = engine(x, job["y"], job["noise"], s, job["mask"], t, None, 0, n_steps=job["n_think"])
            outs.append(den.clone())
            xs.append(x.clone())
            if i + 1 < len(job["sig_list"]):
                x = torch.lerp(den, x, job["ratios"][i])
    torch.cuda.synchronize()
    return outs, xs, torch.cuda.get_rng_state(0).clone()


def _compare(got, want, what):
    import torch
    g, w = got.double(), want.double()
    assert torch.isfinite(g).all(), f"{what}: non-finite values"
    scale = max(1.0, float(w.abs().max()))
    err = float((g - w).abs().max())
    mse = float(((g - w) ** 2).mean())
    assert err <= REL * scale, f"{what}: max abs err {err:.3e} > {REL * scale:.3e}"
    assert mse <= MSE * scale * scale, f"{what}: MSE {mse:.3e}"
    return err / scale, mse


def _engines(job, **product_kw):
    import bench
    from lanpaint_amd import LanPaint
    ref_cls = _reference_class()
    h = bench.HYPER
    args = (job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], False, job["flow"])
    return ref_cls(TwoHeads(job["flow"]), *args), LanPaint(TwoHeads(job["flow"]), *args, **product_kw)


@pytest.fixture(autouse=True)
def _defaults_only(monkeypatch):
    for var in ("LANPAINT_AMD_GRAPH", "LANPAINT_AMD_RNG", "LANPAINT_AMD_AUTO_PACK"):
        monkeypatch.delenv(var, raising=False)


def _same_seed_run(job, seed, inference=False, **product_kw):
    import torch
    ref, mine = _engines(job, **product_kw)
    want_out, want_x, want_state = _walk(ref, job, seed)
    got_out, got_x, got_state = _walk(mine, job, seed, inference=inference)
    worst = [0.0, 0.0]
    for i, (a, b, c, d) in enumerate(zip(got_out, want_out, got_x, want_x)):
        for tag, g, w in (("out", a, b), ("in-place x", c, d)):
            e, m = _compare(g, w, f"sigma call {i}: {tag}")
            worst = [max(worst[0], e), max(worst[1], m)]
    assert torch.equal(got_state, want_state), "the product leaves the device generator in another state than the reference"
    return mine, worst


def test_c2_whole_schedule_port_on_gpu_vs_default_engine_same_seed():
    """BASELINE configs[1]: SDXL 1x4x128x128, 30 Karras sigmas x 5 think iterations -- 30 sigma calls, 270 randn draws."""
    job = _job("c2_sdxl")
    mine, worst = _same_seed_run(job, 20250924)
    assert mine.graph == "auto" and mine.rng == "torch"
    assert len(mine._graphs) == 1 and not mine._graph_blocked        # the default engine did capture (and verify) the call
    assert mine.iterations_run == 150
    assert worst[1] < 1e-10, worst                                   # what the build achieves


def test_c2_port_on_gpu_vs_default_engine_under_inference_mode():
    """ComfyUI runs its nodes under torch.inference_mode(): tensors without version counters (the first 8 sigma calls)."""
    job = _job("c2_sdxl", n_sig=8)
    _same_seed_run(job, 7, inference=True)


def test_c3_four_rows_each_on_its_own_sigma_ramp():
    """BASELINE configs[2] per GPU: 4x4x128x128 with PER-ROW sigmas (rows on 1 / 0.9 / 0.8 / 0.7 x the Karras ramp): the
    per-row broadcast of every time tensor and the flow-form replace step the reference uses for a non-scalar sigma
    (lanpaint.py:89-92), whole schedule."""
    job = _job("c3_sdxl_b4", row_ramp=(1.0, 0.9, 0.8, 0.7))
    mine, _ = _same_seed_run(job, 31337)
    assert mine.iterations_run == 150


def test_c5_video_latent_first_three_sigmas():
    """BASELINE configs[4]: Wan 1x16x21x60x104 (5-D, flow, temporal mask), the first 3 sigma calls: past ATen's grid cap
    the reference's randn is a grid-stride kernel -- the product's ATen-strided lanes must reproduce it."""
    job = _job("c5_wan", n_sig=3)
    _same_seed_run(job, 99)


def test_c4_flux_flow_schedule_ten_iterations_per_sigma():
    """BASELINE configs[3]: Flux 1x16x64x64, flow, 10 think iterations per sigma (first 10 sigma calls)."""
    job = _job("c4_flux", n_sig=10)
    _same_seed_run(job, 4)


@pytest.mark.parametrize("kw", [dict(graph=False), dict(graph=True)])
def test_c2_port_on_gpu_vs_forced_launch_modes(kw):
    """The same comparison with the launch mode forced (eager launches / hipGraph replay from the first call), 10 sigmas."""
    job = _job("c2_sdxl", n_sig=10)
    _same_seed_run(job, 11, **kw)


def test_the_arbiter_has_teeth():
    """Another seed on the product side must NOT pass: the comparison is not vacuous."""
    import torch
    job = _job("c1_sd15", n_sig=3)
    ref, mine = _engines(job)
    want_out, _, _ = _walk(ref, job, 1)
    got_out, _, _ = _walk(mine, job, 2)
    assert float((got_out[-1].double() - want_out[-1].double()).abs().max()) > 1e-2


# ------------------------------------------------------------------ the rest of the engine's call surface, same arbiter
def _walk_with_options(engine, job, seed, options_for_call, extra_kw=None):
    """`_walk` with a fresh model_options dict per sigma call (the stopper reads and the trace list fills it) and optional
    keyword arguments of the AV form; returns per call (out, x, trace records) and the generator state."""
    import torch
    torch.manual_seed(seed)
    x = job["x0"].clone()
    calls = []
    with torch.no_grad():
        for i, (s, t) in enumerate(zip(job["sig_list"], job["times_list"])):
            mo = options_for_call(i)
            den = engine(x, job["y"], job["noise"], s, job["mask"], t, mo, 0, n_steps=job["n_think"], **(extra_kw or {}))
            calls.append((den.clone(), x.clone(), list(mo.get("lanpaint_semantic_trace", [])) if mo else []))
            if i + 1 < len(job["sig_list"]):
                x = torch.lerp(den, x, job["ratios"][i])
    torch.cuda.synchronize()
    return calls, torch.cuda.get_rng_state(0).clone()


@pytest.mark.parametrize("workload,thr,pat,n_think,n_sig,ramp", [
    ("c2_sdxl", 8.0, 1, 10, 14, None),    # the threshold scales with abt: the early (noisy) calls run out, the later ones stop
    ("c2_sdxl", 8.0, 2, 10, 14, None),    # patience_eff = patience + 1 quiet checks in a row (the drift anchor is live in between)
    ("c2_sdxl", 1e-9, 2, 6, 6, None),     # armed, never fires: the watched loop runs to the end
    ("c3_sdxl_b4", 8.0, 1, 8, 13, (1.0, 0.9, 0.8, 0.7)),   # four rows on their own sigmas: abt is the mean over rows
    ("c5_wan", 2000.0, 1, 6, 2, None),    # 5-D video latent: no ring weight (earlystop.py:32-50 returns None)
], ids=["c2_patience1", "c2_patience2", "c2_never_fires", "c3_rows", "c5_video"])
@pytest.mark.parametrize("kw", [{}, {"graph": True}], ids=["default_engine", "gated_graph_launches"])
def test_inner_early_stop_port_on_gpu_vs_default_engine_same_seed(workload, thr, pat, n_think, n_sig, ramp, kw):
    """earlystop.py's stopper inside the reference loop on the GPU (it breaks out of its Python loop and stops drawing) next
    to the product's device-side verdict (graph launches gated on the device, the generator rewound by the draws the
    skipped iterations would have made): per sigma call the same iteration count, the same trace records, outputs inside the
    fp32 tolerance above -- and the same generator state at the end, which fails if a single draw is miscounted.
    (The default engine meets a fresh model_options dict per call here and so stays on watched eager launches; `graph=True`
    puts the same schedule through the device-gated launches of a replayed graph.)"""
    import torch
    job = _job(workload, row_ramp=ramp, n_sig=n_sig)
    job["n_think"] = n_think
    ref, mine = _engines(job, **kw)
    opt = lambda i: {"lanpaint_semantic_stop": {"threshold": thr, "patience": pat}, "lanpaint_semantic_trace": []}   # noqa: E731
    want, want_state = _walk_with_options(ref, job, 5150, opt)
    got, got_state = _walk_with_options(mine, job, 5150, opt)
    ran = []
    for i, ((go, gx, gt), (wo, wx, wt)) in enumerate(zip(got, want)):
        assert len(gt) == len(wt), f"sigma call {i}: {len(gt)} iterations against the reference's {len(wt)}"
        for key in ("inner_step", "patience_counter", "patience_eff", "stopped"):
            assert [r[key] for r in gt] == [r[key] for r in wt], (i, key)
        for key in ("dist", "threshold", "threshold_eff", "abt"):
            np.testing.assert_allclose([r[key] for r in gt], [r[key] for r in wt], rtol=2e-4, err_msg=f"call {i}: {key}")
        assert [r["dist_ring"] is None for r in gt] == [r["dist_ring"] is None for r in wt]
        assert [r["dist_drift"] is None for r in gt] == [r["dist_drift"] is None for r in wt]
        _compare(go, wo, f"sigma call {i}: out")
        _compare(gx, wx, f"sigma call {i}: in-place x")
        ran.append(len(wt))
    assert torch.equal(got_state, want_state), "generator state differs: a skipped iteration's draws were not rewound exactly"
    assert mine.iterations_run == sum(ran) == ref_iterations(ref, ran)
    assert (len(mine._graphs) >= 1 and all(c.es is not None for c in mine._graphs.values())) if kw else not mine._graphs
    if thr < 1e-6:
        assert all(r == n_think for r in ran)
    else:
        assert any(r < n_think for r in ran), ran          # the stop did fire somewhere on this schedule
    print(f"[early stop on the reference, {workload} thr={thr} patience={pat}] iterations per sigma call: {ran}")


def ref_iterations(ref, ran):
    """The reference keeps no iteration counter; its backbone stub's call count is one final call + one per iteration."""
    return ref.inner_model.calls - len(ran)


@pytest.mark.parametrize("shape,split,rows_differ", [((1, 8, 66000), 40001, False), ((3, 4, 777), 500, True)],
                         ids=["one_row_16B_lanes", "rows_on_their_own_time_pairs"])
def test_av_pack_port_on_gpu_vs_default_engine_same_seed(shape, split, rows_differ):
    """The flat audio+video pack (lanpaint.py:60-74, 173-180: per-element blended times, the audio correction on the score)
    through the reference on the GPU and through the product's two-row table path, same seed, three calls with moving times."""
    import torch
    from tests.test_gpu_av import _case
    from lanpaint_amd import LanPaint
    from oracle import lanpaint_oracle as orc
    c = _case(shape, split, seed=77, rows_differ=rows_differ)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda")   # noqa: E731
    ref_cls = _reference_class()
    res = {}
    for tag, cls in (("ref", ref_cls), ("mine", LanPaint)):
        torch.manual_seed(2718)
        eng = cls(TwoHeads(True), 4, 15.0, 5.0, 1.0, 0.2, False, True)
        outs = []
        x = tt(c["x"])
        with torch.no_grad():
            for k in range(3):
                f = np.float32(1.0 - 0.15 * k)
                tv, ta = orc.times_from_sigma(c["sigma"] * f, True), orc.times_from_sigma(np.asarray(c["times_a"][1]) * f, True)
                out = eng(x, tt(c["y"]), tt(c["noise"]), tt(c["sigma"] * f), tt(c["mask"]), tuple(tt(t) for t in tv), None, 0,
                          current_times_audio=tuple(tt(t) for t in ta), audio_indicator=tt(c["ai"]),
                          audio_correction=tt(c["corr"]))
                outs.append((out.clone(), x.clone()))
        torch.cuda.synchronize()
        res[tag] = (outs, torch.cuda.get_rng_state(0).clone(), eng)
    assert res["mine"][2]._desc.flags & (1 << 17), "the product did not take the two-row table (LP_FL_AV)"
    for k, ((go, gx), (wo, wx)) in enumerate(zip(res["mine"][0], res["ref"][0])):
        _compare(go, wo, f"AV call {k}: out")
        _compare(gx, wx, f"AV call {k}: in-place x")
    assert torch.equal(res["mine"][1], res["ref"][1])


@pytest.mark.parametrize("variant", ["is_flux", "soft_mask", "undeclared_noise_scaling", "single_head_backbone"])
def test_engine_variants_port_on_gpu_vs_default_engine_same_seed(variant):
    """IS_FLUX (lanpaint.py:20-22 folds it into the flow form), a soft mask (per-element replace / score weights), a
    model_sampling the product cannot recognise (its noise_scaling is called back, lanpaint.py:86-92), and a backbone that
    returns ONE tensor (unpack_model_output's list / tensor branches, lanpaint.py:31-46)."""
    import torch
    import bench
    from lanpaint_amd import LanPaint
    from tests.stubs import OpaqueVESampling
    flow = variant == "is_flux"
    job = _job("c4_flux" if flow else "c1_sd15", n_sig=6)
    if variant == "soft_mask":
        g = torch.Generator(device="cuda").manual_seed(3)
        job["mask"] = torch.rand(job["mask"].shape, device="cuda", generator=g)

    class OneHead(TwoHeads):
        def __call__(self, x, t, model_options=None, seed=None):
            self.calls += 1
            return x / (1.0 + t.reshape((-1,) + (1,) * (x.ndim - 1)) ** 2)

    def model():
        m = OneHead(flow) if variant == "single_head_backbone" else TwoHeads(flow)
        if variant == "undeclared_noise_scaling":
            m.model_sampling = OpaqueVESampling()
        return m

    h = bench.HYPER
    args = (job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], flow, False)
    ref, mine = _reference_class()(model(), *args), LanPaint(model(), *args)
    want_out, want_x, want_state = _walk(ref, job, 1234)
    got_out, got_x, got_state = _walk(mine, job, 1234)
    for i, (a, b, c, d) in enumerate(zip(got_out, want_out, got_x, want_x)):
        _compare(a, b, f"{variant}: sigma call {i}: out")
        _compare(c, d, f"{variant}: sigma call {i}: in-place x")
    assert torch.equal(got_state, want_state)


@pytest.mark.parametrize("kw", [{}, {"graph": True}], ids=["default_engine", "gated_graph_launches"])
def test_av_pack_with_inner_early_stop_port_on_gpu_vs_default_engine_same_seed(kw):
    """Both at once: the reference's stopper on a flat audio+video pack (threshold from the mean of the BLENDED abt tensor,
    earlystop.py:104-110) against the product on the two-row table -- watched eager launches (default engine, a fresh options
    dict per call) and the device-gated launches of a replayed graph (round 5)."""
    import torch
    from tests.test_gpu_av import _case
    from lanpaint_amd import LanPaint
    shape, split, n_steps = (1, 8, 66000), 40001, 8
    c = _case(shape, split, seed=78)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda")   # noqa: E731
    res = {}
    for tag, cls in (("ref", _reference_class()), ("mine", LanPaint)):
        torch.manual_seed(1618)
        eng = cls(TwoHeads(True), n_steps, 15.0, 5.0, 1.0, 0.2, False, True, **(kw if tag == "mine" else {}))
        x, y, noise, mask, ai, corr = (tt(c[k]) for k in ("x", "y", "noise", "mask", "ai", "corr"))
        sig, tv, ta = tt(c["sigma"]), tuple(tt(t) for t in c["times_v"]), tuple(tt(t) for t in c["times_a"])
        calls = []
        with torch.no_grad():
            for k in range(4):
                mo = {"lanpaint_semantic_stop": {"threshold": 5.0 if k != 2 else 1e-9, "patience": 1}, "lanpaint_semantic_trace": []}
                out = eng(x, y, noise, sig, mask, tv, mo, 0, current_times_audio=ta, audio_indicator=ai, audio_correction=corr)
                calls.append((out.clone(), x.clone(), list(mo["lanpaint_semantic_trace"])))
                x = x + 0.1 * (out - x)
        torch.cuda.synchronize()
        res[tag] = (calls, torch.cuda.get_rng_state(0).clone(), eng)
    ran = []
    for k, ((go, gx, gt), (wo, wx, wt)) in enumerate(zip(res["mine"][0], res["ref"][0])):
        assert [(r["inner_step"], r["patience_counter"], r["stopped"]) for r in gt] == \
               [(r["inner_step"], r["patience_counter"], r["stopped"]) for r in wt], k
        np.testing.assert_allclose([r["dist"] for r in gt], [r["dist"] for r in wt], rtol=2e-4)
        np.testing.assert_allclose([r["threshold_eff"] for r in gt], [r["threshold_eff"] for r in wt], rtol=1e-5)
        _compare(go, wo, f"AV + early stop, call {k}: out")
        _compare(gx, wx, f"AV + early stop, call {k}: in-place x")
        ran.append(len(wt))
    assert torch.equal(res["mine"][1], res["ref"][1])
    assert ran[2] == n_steps and any(r < n_steps for r in ran), ran
    mine = res["mine"][2]
    assert mine._desc.flags & (1 << 17), "the product did not take the two-row table (LP_FL_AV)"
    assert (len(mine._graphs) >= 1) == bool(kw)
