"""The configuration bench.py PUBLISHES, tested as such: the engine object the bench times --
hipGraph replay of every sigma call, noise generated inside the kernels, bit-packed mask -- over the whole BASELINE
schedule against the CPU oracle fed the very draws the kernels generate, through the function the bench itself runs before
its timed region (bench.parity_check).  Tolerance: BASELINE.json's MSE < 1e-5 on the final latent and on every denoised;
what the build achieves (asserted too) is < 1e-9."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _job(workload, mask_format, seed=0, mask_kind=None):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    shape, flow, n_sig, n_think = bench.WORKLOADS[workload]
    sig_np = bench.flow_sigmas(n_sig) if flow else bench.karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), seed, dev, tt, mask_kind=mask_kind)
    mask = bench.attach_mask_format(mask, mask_format)
    sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
    times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
    ratios = bench.euler_ratios(sig_list, len(shape))
    return dict(x0=x0, y=y, noise=noise, mask=mask, sig_list=sig_list, times_list=times_list, ratios=ratios,
                n_think=n_think, flow=flow, n_sig=n_sig, shape=shape)


def _engine(job, **kw):
    import bench
    from lanpaint_amd import LanPaint
    h = bench.HYPER
    return LanPaint(bench.StubBackbone(job["flow"]), job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"],
                    IS_FLOW=job["flow"], MinStepFrac=h["MinStepFrac"], **kw)


def _check(job, eng, **kw):
    import bench
    return bench.parity_check(eng, job["x0"], job["y"], job["noise"], job["mask"], job["sig_list"], job["times_list"],
                              job["ratios"], job["n_think"], job["flow"], **kw)


def test_the_published_configuration_against_the_oracle():
    """BENCH.config literally: c2_sdxl 1x4x128x128, 30 Karras sigmas x 5, rng=philox, graph replay, bit-packed mask."""
    job = _job("c2_sdxl", "bits")
    eng = _engine(job, rng="philox", philox_seed=0, graph=True)
    r = _check(job, eng)
    assert r["ok"] and r["sigmas_checked"] == 30 and r["draws"] == 30 * 9 and r["think_iterations_checked"] == 150
    assert r["launch_modes"] == {"graph": 30}, r["launch_modes"]          # every call was a replay, none fell back to eager
    assert r["mse_x"] < 1e-5 and r["mse_denoised_max"] < 1e-5, r
    assert r["mse_x"] < 1e-9 and r["mse_denoised_max"] < 1e-9, r          # what the build achieves
    assert len(eng._graphs) == 1 and eng.iterations_run == 150


def test_the_engine_built_with_no_optional_keyword_against_the_oracle(monkeypatch):
    """`engine_defaults`: LanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize, IS_FLUX, IS_FLOW) -- graph="auto",
    rng="torch" (the reference's own randn stream, generated in-kernel), the reference's fp32 mask -- against the oracle on
    what torch.randn itself returns from the same generator state; the generator must end every call where the reference's
    draws would leave it (checked inside parity_check)."""
    import torch
    import bench
    from lanpaint_amd import LanPaint
    for var in ("LANPAINT_AMD_GRAPH", "LANPAINT_AMD_RNG", "LANPAINT_AMD_AUTO_PACK"):      # (this test is about the defaults)
        monkeypatch.delenv(var, raising=False)
    job = _job("c2_sdxl", "f32")
    h = bench.HYPER
    eng = LanPaint(bench.StubBackbone(False), job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], False, False)
    assert eng.graph == "auto" and eng.rng == "torch"
    torch.manual_seed(20240924)
    r = _check(job, eng)
    assert r["ok"] and r["launch_modes"] == {"torch": 30} and r["draws"] == 270
    assert r["mse_x"] < 1e-9 and r["mse_denoised_max"] < 1e-9, r
    assert len(eng._graphs) == 1 and not eng._graph_blocked               # auto mode did capture (and verify) the call


@pytest.mark.parametrize("workload,kw,fmt,max_sigmas", [
    ("c2_sdxl", dict(rng="philox", philox_seed=3, graph=False), "bits", None),      # eager launches, host-side launch counter
    ("c2_sdxl", dict(rng="torch", graph=True), "bits", None),                        # the reference's stream inside a replayed graph
    ("c1_sd15", dict(rng="philox", philox_seed=1, graph=True), "bits", None),        # BASELINE configs[0] shape
    ("c3_sdxl_b4", dict(rng="philox", philox_seed=2, graph=True), "bits", None),     # configs[2]: 4 rows per GPU
    ("c4_flux", dict(rng="philox", philox_seed=4, graph=True), "bits", 10),          # configs[3], flow, 10 iterations per sigma
    ("c5_wan", dict(rng="philox", philox_seed=5, graph=True), "bits", 2),            # configs[4]: video latent, 16 B per lane
    ("c5_wan", dict(rng="torch", graph=True), "f32", 1),                             # ... ATen-strided lanes, fp32 mask
])
def test_other_bench_configurations_against_the_oracle(workload, kw, fmt, max_sigmas):
    import torch
    job = _job(workload, fmt)
    eng = _engine(job, **kw)
    torch.manual_seed(7)
    r = _check(job, eng, max_sigmas=max_sigmas)
    assert r["ok"] and r["mse_x"] < 1e-9 and r["mse_denoised_max"] < 1e-9, r
    want = "torch" if kw["rng"] == "torch" else ("graph" if kw["graph"] else "eager")
    assert set(r["launch_modes"]) == {want}, r["launch_modes"]


def test_the_check_has_teeth():
    """A checker that cannot fail proves nothing: hand it the wrong noise stream (sequence numbers off by one) and it must
    report a failure, not a small number."""
    job = _job("c1_sd15", "bits")
    eng = _engine(job, rng="philox", philox_seed=0, graph=True)
    good = _check(job, eng, max_sigmas=3)
    assert good["ok"]
    real = eng.rng_position
    eng.rng_position = lambda dev: tuple(v + (1 if k == 0 else 0) for k, v in enumerate(real(dev)))
    bad = _check(job, eng, max_sigmas=3)
    assert not bad["ok"] and bad["mse_x"] > 1e-3, bad


@pytest.mark.timeout(600)
def test_bench_line_carries_the_parity_check():
    """`python bench.py` prints `parity_check` for the configuration it times and a `value` only next to ok = true."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "0",
               "--prewarm-seconds", "0.05", "--no-cpu-baseline", "--no-summary", "--sidecar", os.path.join(tmp, "x.json")]
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
        assert p.returncode == 0, p.stderr[-3000:]
        side = json.load(open(os.path.join(tmp, "x.json")))
    last = p.stdout.rstrip().splitlines()[-1]
    assert last.startswith("{") and len(last) < 6000
    line = json.loads(last)
    pc = line["parity_check"]
    # the default run IS the drop-in engine: the reference's torch stream, graph="auto" (captured on the job's second call)
    assert pc["ok"] and pc["sigmas_checked"] == 30 and pc["launch_modes"] == {"torch": 30}
    assert pc["mse_x"] < 1e-9 and pc["mse_denoised_max"] < 1e-9 and line["value"] > 0
    cfg = line["config"]
    assert cfg["rng"] == "torch" and cfg["graph"] == "auto" and cfg["mask_format"] == "f32" and cfg["mask_seen_by_kernels"] == "bits"
    assert cfg["captured_calls"] == 1 and "drop-in" in cfg["engine"]
    assert "RNG" not in line["roofline"]["kernel"] and ", 1, false, 0>" in line["roofline"]["kernel"]      # the torch-stream instantiation
    assert side["parity_check"]["draws"] == 270 and side["roofline"]["launches_timed"] == 120 and abs(side["line"]["value"] - line["value"]) <= 1e-5 * line["value"]


@pytest.mark.parametrize("workload,max_sigmas", [("c2_sdxl", None), ("c5_wan", 2)])
def test_bf16_backbone_configuration_against_the_oracle(workload, max_sigmas):
    """BASELINE configs[1] says "bf16": the backbone's dtype.  The engine with model_dtype=torch.bfloat16 (x_in emitted as bf16,
    both heads read as bf16 -- at the video latent through the 16-byte lane-pair accesses) against the oracle whose stub is
    restated in bf16 (bench.Bf16StubOracle): a rounding that flips on an fp32 last-bit difference moves an element by one
    bf16 ulp, so the bound is the BASELINE tolerance, with the measured value far below."""
    import torch
    job = _job(workload, "bits")
    eng = _engine(job, rng="philox", philox_seed=6, graph=True, model_dtype=torch.bfloat16)
    r = _check(job, eng, max_sigmas=max_sigmas)
    assert r["ok"] and r["mse_x"] < 1e-5 and r["mse_denoised_max"] < 1e-5, r      # (measured 2e-8 / 2e-6: a few one-ulp bf16 flips
    assert set(r["launch_modes"]) == {"graph"}                                       # at |x| ~ 15, the first sigma of the schedule)


@pytest.mark.parametrize("graph", [True, False])
def test_sdxl_shaped_backbone_with_fused_cfg_heads_against_the_oracle(graph):
    """BASELINE configs[1] / [3] behind a backbone that exercises the matrix cores (round 5): the SDXL-shaped stand-in
    (tests/sdxl_standin.py: ResBlocks at 128 / 64 / 32 px, self- + cross-attention over 1024 tokens), one batched cond + uncond
    pass per call handed over as FusedCFGHeads -- the kernels form both CFG heads themselves.  With fp32 weights the module is
    reproducible to ~1e-5 and the strict bound holds against the oracle driving the SAME module with the reference's eager
    cfg_function form (measured 2e-10); with bf16 weights the comparison is bounded by the network's own run-to-run noise
    (MIOpen / hipBLASLt bf16 kernels are not bitwise reproducible; one flipped rounding of eps is amplified by sigma and the CFG
    scale), so that variant is checked against TWICE-RUN noise of the same engine, not against 1e-5."""
    import torch
    import bench
    from lanpaint_amd import LanPaint, _cabi
    from tests.sdxl_standin import SDXLShapedBackbone
    job = _job("c2_sdxl", "bits")
    h = bench.HYPER
    dev = torch.device("cuda", 0)
    net32 = SDXLShapedBackbone(dev, dtype=torch.float32)
    eng = LanPaint(net32, job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], rng="philox", philox_seed=9, graph=graph)
    r = bench.parity_check(eng, job["x0"], job["y"], job["noise"], job["mask"], job["sig_list"], job["times_list"], job["ratios"],
                           job["n_think"], job["flow"], max_sigmas=3, oracle_model=net32.as_oracle_model())
    assert r["ok"] and r["mse_x"] < 1e-7 and r["mse_denoised_max"] < 1e-7, r
    assert set(r["launch_modes"]) == {"graph" if graph else "eager"}
    if not graph:       # the think-loop launches really took the fused-CFG form
        assert eng._desc.flags & _cabi.LP_FL_CFG_FUSED and eng._desc.flags & _cabi.LP_FL_MASK_BITS
    del net32, eng
    torch.cuda.empty_cache()
    # bf16 weights, every stream half-width (model_dtype = bf16: x_in emitted as bf16; half_out: bf16 predictions)
    net = SDXLShapedBackbone(dev, half_out=True)
    mk = lambda: LanPaint(net, job["n_think"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], rng="torch", graph=graph,      # noqa: E731
                          model_dtype=torch.bfloat16)
    eng = mk()
    finals = []
    for _ in range(2):
        torch.manual_seed(77)
        finals.append(bench.schedule_pass(eng, job["x0"], job["y"], job["noise"], job["mask"], job["sig_list"][:3], job["times_list"][:3],
                                          job["ratios"][:2], job["n_think"]).double())
    noise_floor = float(((finals[0] - finals[1]) ** 2).mean())
    torch.manual_seed(77)
    r = bench.parity_check(mk(), job["x0"], job["y"], job["noise"], job["mask"], job["sig_list"], job["times_list"], job["ratios"],
                           job["n_think"], job["flow"], max_sigmas=3, oracle_model=net.as_oracle_model(input_dtype=torch.bfloat16))
    assert np.isfinite(r["mse_x"]) and r["mse_x"] <= max(1e-5, 50.0 * noise_floor) and r["mse_x"] < 0.5, (r["mse_x"], noise_floor)
    if not graph:
        fl = eng._desc.flags           # (of the call's LAST think launch, whose emit is the fp32 x that is written back: no XIN flag there)
        assert fl & _cabi.LP_FL_CFG_FUSED and fl & _cabi.LP_FL_X0_BF16 and eng.model_dtype == torch.bfloat16
