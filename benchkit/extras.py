"""Secondary measurements of the same build.  None of this is on the headline line: `bench.py --extras 1` writes it to the
side-car file (bench_extras.json), and two of its numbers appear on the line as scalars (`summary`)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

from .parity import PARITY_TOL, check_job, parity_check
from .roofline import (ROOT, _latest_profile_json, graph_burst_us_per_launch, measure_steady_launch, roofline_fields)
from .workloads import HYPER, WORKLOADS, Job, StubBackbone, euler_ratios, karras_sigmas, make_inputs, schedule_pass, times_from_sigma


def build_node_sampler(args, dev, rng=None, **engine_kw):
    """C2 behind KSamplerX0Inpaint with the node defaults (MinStepFrac = 1.0, EarlyStop = 1): (the sampler callable, a function
    that walks the schedule once with k-diffusion's Euler update, the number of sigmas)."""
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes as lpn
    shape, flow, n_sig, n_think = WORKLOADS["c2_sdxl"]
    sig_np = karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
    sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
    ratios = euler_ratios(sig_list, 4)
    model = StubBackbone(flow)
    model.model_type = "EPS"
    k = lpn.KSamplerX0Inpaint(model, torch.cat([tt(sig_np), torch.zeros(1, device=dev)]))
    k.latent_image, k.noise = y, noise
    kw = dict(engine_kw)
    if rng is not None:
        kw.update({"rng": rng, "philox_seed": args.seed})
        if args.graph is not None:
            kw["graph"] = bool(args.graph)
    k.PaintMethod = LanPaint(model, n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                             MinStepFrac=1.0, **kw)
    k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
    denoise_mask = 1.0 - mask
    model_options = {}                     # ComfyUI hands the SAME dict to every step

    def node_pass(record=None):
        x = x0.clone()
        for i in range(n_sig):
            den = k(x, sig_list[i], denoise_mask, model_options=model_options, seed=args.seed)
            if record is not None:
                record.append(int(k._node_desc.n_eff) if k._node_desc is not None else None)
            if i + 1 < n_sig:
                x = torch.lerp(den, x, ratios[i])
        return x
    return k, node_pass, n_sig


def node_default_schedule(args, dev, rng=None, passes=None):
    """C2 driven through KSamplerX0Inpaint with the node defaults (MinStepFrac = 1.0 => n_eff = round(N (1 - abt)), last
    sigma skipped: SURVEY.md 8d's second line): the path ComfyUI's sampler functions call."""
    k, node_pass, n_sig = build_node_sampler(args, dev, rng=rng)
    for _ in range(8):          # (captures for every inner-step count of the ramp, then a few steady passes)
        node_pass()
    torch.cuda.synchronize()
    reps = passes or max(20, args.steps // 8)
    it0, t0 = k.PaintMethod.iterations_run, time.perf_counter()
    for _ in range(reps):
        node_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iters = k.PaintMethod.iterations_run - it0
    return {"value": iters / dt, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps, "iterations_per_step": iters // reps,
            "us_per_sigma_call": 1e6 * dt / (reps * n_sig), "rng": k.PaintMethod.rng, "graph": k.PaintMethod.graph,
            "captured_calls": len(k.PaintMethod._graphs),
            "note": "C2 through KSamplerX0Inpaint, MinStepFrac=1.0, EarlyStop=1 (n_eff = round(5(1-abt)), last sigma 0)"}


def summary_scalars(args, dev, value, _cabi):
    """A handful of scalars next to the headline (N = 1): the same workload in the fastest configuration (independent in-kernel
    Philox stream, caller-packed mask, forced graph), the node-default schedule through KSamplerX0Inpaint, and the steady
    kernel's bandwidth fraction where it is bandwidth-bound.  Details go to the side-car."""
    out, detail = {}, {}
    try:
        job = Job(args.workload, dev, seed=args.seed, mask_kind=args.mask, mask_format="bits")
        eng = job.engine(rng="philox", philox_seed=args.seed, graph=True)
        par = check_job(job, eng, max_sigmas=4)
        it_s, ms = job.timed(eng, max(5, args.steps // 2))
        out["philox_bits_it_s"] = it_s if par["ok"] else None
        out["value_over_philox_bits"] = (value / it_s) if (value and par["ok"]) else None
        detail["philox_bits"] = {"value": it_s, "ms_per_step": ms, "parity_check": par,
                                 "config": "rng='philox', graph=True, mask bit-packed by the caller (lanpaint_amd.pack_mask)"}
    except Exception as e:
        detail["philox_bits"] = {"error": repr(e)}
    try:
        nd = node_default_schedule(args, dev, **({} if args.rng is None else {"rng": args.rng}))
        out["node_default_schedule_it_s"] = nd["value"]
        detail["node_default_schedule"] = nd
    except Exception as e:
        detail["node_default_schedule"] = {"error": repr(e)}
    try:
        from .floor import engine_floor
        fl = engine_floor(_cabi, dev, workload=args.workload, seed=args.seed, steps=max(10, args.steps), mask_kind=args.mask,
                          **({} if args.rng is None else {"rng": args.rng, "graph": True if args.graph is None else bool(args.graph)}))
        detail["launch_floor"] = fl
        if "error" not in fl:
            out["launch_floor_it_s"] = fl["floor_it_s"]
            out["value_over_launch_floor"] = (value / fl["floor_it_s"]) if value else None
    except Exception as e:
        detail["launch_floor"] = {"error": repr(e)}
    try:      # what one steady launch costs as a node of a replayed graph (kernel + the dependent-dispatch gap), the form the timed
        # region launches it in: beside roofline.duration_used_us, which is the begin -> end of an eager dispatch
        from .roofline import graph_burst_us_per_launch
        out["steady_graph_node_us"] = graph_burst_us_per_launch(_cabi, args.workload, dev, reps=200, replays=10, mask_kind=args.mask,
                                                                rng="torch" if (args.rng or "torch").startswith("torch") else "philox")
    except Exception as e:
        detail["steady_graph_node_us"] = {"error": repr(e)}
    if not args.no_large_shape:
        for key, wl, rng in (("hbm_frac_c5_wan", "c5_wan", "philox"), ("hbm_frac_c5_wan_torch_stream", "c5_wan", "torch"),
                             ("hbm_frac_past_l3", "x_wan_b16", "philox")):
            if wl == args.workload and rng == (args.rng or "torch"):
                continue
            try:
                r = measure_steady_launch(_cabi, dev, workload=wl, launches=60, warm_s=0.2, rng=rng)
                out[key] = r["frac"]
                detail[key] = r
                torch.cuda.empty_cache()
            except Exception as e:
                detail[key] = {"error": repr(e)}
    return out, detail


def variant(args, dev, label, note, mask_format="bits", steps=None, check=4, **engine_kw):
    """The headline workload with another engine configuration: parity (bounded) then it/s."""
    try:
        job = Job(args.workload, dev, seed=args.seed, mask_kind=args.mask, mask_format=mask_format)
        eng = job.engine(**engine_kw)
        par = check_job(job, eng, max_sigmas=check) if check else None
        it_s, ms = job.timed(eng, steps or max(5, args.steps // 2))
        ok = par is None or par["ok"]
        return {"value": it_s if ok else None, "unit": "think-iterations/s", "ms_per_step": ms, "note": note,
                "parity_check": None if par is None else {k: par[k] for k in ("mse_x", "mse_denoised_max", "ok", "sigmas_checked", "launch_modes")},
                "captured_calls": len(eng._graphs)}
    except Exception as e:
        return {"error": repr(e), "label": label}


def past_l3(_cabi, dev, workload="x_wan_b16", launches=100, warm_s=0.3, rounds=2):
    """The 1.2 GB point, both variants of the launch (every operand streamed / region-aware as shipped), INTERLEAVED on this
    box: per-dispatch event pairs after a warm burst and the un-profiled graph-burst cost per launch."""
    res = {}
    for _ in range(rounds):
        for every in (True, False):
            key = "every_stream" if every else "region_aware"
            m = measure_steady_launch(_cabi, dev, workload=workload, launches=launches, every_stream=every, warm_s=warm_s)
            g = graph_burst_us_per_launch(_cabi, workload, dev, reps=50, replays=20, every_stream=every)
            r = res.setdefault(key, {"event_mean_us": [], "graph_burst_us": [], "last": None})
            r["event_mean_us"].append(m["mean_launch_us"])
            r["graph_burst_us"].append(g)
            r["last"] = m
            torch.cuda.empty_cache()
    out = {}
    for key, r in res.items():
        m, ev = r["last"], float(np.mean(r["event_mean_us"]))
        blk = roofline_fields(m["algorithmic_bytes_per_launch"], ev, m["traffic"], m["every_stream_bytes_per_launch"])
        blk.update({"bytes_model": m["bytes_model"], "event_mean_us_per_round": r["event_mean_us"],
                    "graph_burst_us_per_round": r["graph_burst_us"], "committed_profile": m["committed_profile"], "regime": m["regime"]})
        out[key] = blk
    return out


def port_gpu_eager(workload, dev, product_it_s, budget_s=15.0, passes=5):
    """The CPU port of the reference (oracle/lanpaint_oracle.py, TorchBackend) handed DEVICE tensors: the reference's own eager
    ATen launch sequence (~164 per think iteration, its torch.randn_like draws, one host sync per iteration) on THIS GPU -- what a
    ComfyUI user of the reference gets on this device -- next to the product."""
    from oracle.lanpaint_oracle import OracleLanPaint, TorchBackend
    job = Job(workload, dev, seed=0)
    eng = OracleLanPaint(StubBackbone(job.flow), job.n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                         is_flow=job.flow, min_step_frac=HYPER["MinStepFrac"], backend=TorchBackend())

    def one_pass(n_s):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        job.run(eng, n_s)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    state = torch.cuda.get_rng_state(dev)
    try:
        per_sigma = min(one_pass(1), one_pass(1))
        n_s = job.n_sig if (passes + 1) * job.n_sig * per_sigma <= budget_s else max(1, int(budget_s / ((passes + 1) * per_sigma)))
        one_pass(n_s)
        vals = [n_s * job.n_think / one_pass(n_s) for _ in range(passes)]
    finally:
        torch.cuda.set_rng_state(state, dev)
    med = float(np.median(vals))
    return {"value": med, "unit": "think-iterations/s", "passes": len(vals), "sigma_calls_per_pass": n_s,
            "product_over_port_same_gpu": (product_it_s / med) if med > 0 else None,
            "note": "eager ATen launches of the reference's op sequence (the port on device tensors); the reference has no other mode"}


def rccl_single_rank_selftest(timeout_s=150):
    """First contact with RCCL at N = 1: a CHILD process brings up a ONE-rank "nccl" process group on this GPU and pushes a job
    through the very functions the N > 1 path uses (lanpaint_amd.distributed.single_rank_selftest), with a time limit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LANPAINT_AMD_FORCE_COLLECTIVES="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, "-c",
                            "import sys, json; sys.path.insert(0, %r); from lanpaint_amd import distributed as d; "
                            "print(json.dumps(d.single_rank_selftest()))" % ROOT],
                           env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"ok": False, "error": f"child exited {p.returncode}", "stderr_tail": p.stderr[-600:]}
        out = json.loads(lines[-1])
        out["wall_s"] = time.perf_counter() - t0
        return out
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": f"no answer within {timeout_s} s"}
    except Exception as e:
        return {"ok": False, "error": repr(e)}


def live_pmc_traffic(workload, rng, mask_kind=None, launches=20):
    """roofline.traffic measured in THIS run: the steady-launch micro-benchmark (scripts/microbench_step.py) re-executed under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (two separate passes, no other trace domain, as
    MI355X_MICROARCH.md's HBM section prescribes), bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 reports half
    of a coalesced streaming read in FETCH_SIZE)."""
    import re
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    kb = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="lp_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", tmp, "-o", "t", "--", sys.executable,
                   os.path.join(ROOT, "scripts", "microbench_step.py"), workload, "steady", str(launches)] + (["torch"] if rng == "torch" else [])
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            db = os.path.join(tmp, "t_results.db")
            if not os.path.exists(db):
                found = [os.path.join(r, f) for r, _d, fs in os.walk(tmp) for f in fs if f.endswith("_results.db")]
                db = found[0] if found else db
            s = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), db, "--pmc"],
                               capture_output=True, text=True, timeout=120)
            for ln in s.stdout.splitlines():
                m = re.match(r"\| `lp::lp_step_kernel<(\d), \w+, 28u[^>]*>.*?` \| (\w+) \| (\d+) \| ([0-9.]+) \|", ln)
                if m and m.group(2) == ctr:
                    kb[ctr] = (float(m.group(4)), int(m.group(3)))
                    break
            if ctr not in kb:
                return {"error": f"no {ctr} row for the steady kernel", "rocprofv3_rc": p.returncode, "stderr_tail": (p.stderr or "")[-400:],
                        "summary_tail": s.stdout[-400:]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    tb = int(round((2 * kb["FETCH_SIZE"][0] + kb["WRITE_SIZE"][0]) * 1024))
    return {"traffic_bytes_per_launch": tb, "FETCH_SIZE_KB": kb["FETCH_SIZE"][0], "WRITE_SIZE_KB": kb["WRITE_SIZE"][0],
            "dispatches": kb["FETCH_SIZE"][1], "workload": workload, "rng": rng}


def sdxl_shaped_backbone(args, dev):
    """BASELINE configs[1] with a backbone that exercises what the path was built for: SDXL 1x4x128x128, a random-init
    SDXL-SHAPED stand-in (tests/sdxl_standin.py; MIOpen / hipBLASLt / SDPA on the matrix cores), ONE batched cond + uncond pass per
    call handed over as FusedCFGHeads.  `value` is tied to the strict parity statement: the SAME architecture with fp32 weights
    against the oracle driving that module (MSE < 1e-5).  The bf16-weights run is reported beside the network's own run-to-run
    noise (two engine passes from one seed), with the factor between them, never as a parity claim."""
    from lanpaint_amd import LanPaint
    from tests.sdxl_standin import SDXLShapedBackbone
    job = Job("c2_sdxl", dev, seed=args.seed, mask_format="bits")
    n_think = job.n_think
    mk = lambda net, **kw: LanPaint(net, n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],   # noqa: E731
                                    philox_seed=args.seed, graph=True, **kw)
    net32 = SDXLShapedBackbone(dev, flow=False, dtype=torch.float32)
    par32 = check_job(job, mk(net32, rng="philox"), max_sigmas=6, oracle_model=net32.as_oracle_model())
    del net32
    torch.cuda.empty_cache()
    net = SDXLShapedBackbone(dev, flow=False)
    eng = mk(net, rng="philox")
    par = check_job(job, eng, max_sigmas=6, oracle_model=net.as_oracle_model())
    eng_t, finals = mk(net, rng="torch"), []
    for _ in range(2):
        torch.manual_seed(1234)
        finals.append(job.run(eng_t, 6).double())
    self_mse = float(((finals[0] - finals[1]) ** 2).mean())
    it_s, ms = job.timed(eng, 3, warm=2)
    call_ms = ms / job.n_sig
    gb, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)      # the backbone alone: n_think + 1 forward passes per sigma call
    for _ in range(2):
        net.predict(job.x0, job.sig_list[3])
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(gb, stream=side):
        for _ in range(n_think + 1):
            keep = net.predict(job.x0, job.sig_list[3])
    for _ in range(3):
        gb.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        gb.replay()
    torch.cuda.synchronize()
    bb_ms = 1e3 * (time.perf_counter() - t0) / 30
    del keep
    split, split_file = _latest_profile_json("r*_sdxl_standin_time_split.json")
    out = {"value": it_s if par32["ok"] else None, "unit": "think-iterations/s", "ms_per_step": ms, "sigma_call_ms": call_ms,
           "backbone_only_ms_per_sigma_call": bb_ms, "backbone_passes_per_sigma_call": n_think + 1,
           "langevin_path_ms_per_sigma_call": max(0.0, call_ms - bb_ms), "backbone_parameters": net.n_params,
           "parity_check_fp32_weights": {k: par32[k] for k in ("mse_x", "mse_denoised_max", "ok", "sigmas_checked", "launch_modes")},
           "bf16_weights": {"engine_vs_oracle_mse_x": par["mse_x"], "backbone_run_to_run_mse": self_mse,
                            "factor": (par["mse_x"] / self_mse) if self_mse > 0 else None,
                            "note": "not a parity statement: MIOpen / hipBLASLt bf16 kernels are not bitwise reproducible"},
           "langevin_path_share_kernel_trace": (split or {}).get("lp_share_of_gpu_time"), "langevin_path_share_source": split_file}
    del net, eng, eng_t
    torch.cuda.empty_cache()
    return out


def with_dummy_unet(args, dev):
    """BASELINE configs[0] shape (1x4x64x64, 20 sigmas x 5) in front of a random-init SD1.5-shaped dummy UNet in bf16
    (tests/dummy_unet.py), the stand-in (ii) of SURVEY.md 8d."""
    from tests.dummy_unet import DummyUNetBackbone
    job = Job("c1_sd15", dev, seed=args.seed)
    eng = job.engine(DummyUNetBackbone(dev, flow=False), rng="philox", philox_seed=args.seed, graph=True)
    it_s, ms = job.timed(eng, 3, warm=2)
    return {"value": it_s, "unit": "think-iterations/s", "ms_per_step": ms,
            "backbone": "random-init SD1.5-shaped dummy UNet (conv/GroupNorm/SiLU + 1 self-attention block, 1.3 M params, bf16, dual-head output)"}


def extra_measurements(args, dev, value):
    from lanpaint_amd import _cabi
    out = {}

    def guarded(key, fn, *a, **kw):
        try:
            out[key] = fn(*a, **kw)
        except Exception as e:
            out[key] = {"error": repr(e)}

    out["forced_graph_torch_bits"] = variant(args, dev, "forced", "rng='torch', graph=True, caller-packed mask", rng="torch", graph=True)
    out["eager_launches"] = variant(args, dev, "eager", "the drop-in engine with graph=False: eager launches", mask_format="f32", graph=False)
    out["inner_early_stop_armed"] = variant(args, dev, "es", "EarlyStopThreshold > 0 (never reached): the stop rule evaluated on the "
                                            "device inside every replayed launch", check=0, rng="philox", philox_seed=args.seed, graph=True,
                                            EarlyStopThreshold=1e-30, EarlyStopPatience=1)
    out["bf16_backbone"] = variant(args, dev, "bf16", "model_dtype=torch.bfloat16: x_in emitted and both heads read as bf16", check=30 if
                                   int(np.prod(WORKLOADS[args.workload][0])) <= 512 * 1024 else 2, rng="philox", philox_seed=args.seed, graph=True,
                                   model_dtype=torch.bfloat16)
    guarded("roofline_hbm_past_l3", past_l3, _cabi, dev)
    guarded("bf16_heads", lambda: {wl: measure_steady_launch(_cabi, dev, workload=wl, launches=100, model_dtype=torch.bfloat16)
                                   for wl in ("c5_wan", "x_wan_b16")})
    torch.cuda.empty_cache()
    guarded("port_gpu_eager", port_gpu_eager, args.workload, dev, value)
    guarded("rccl_single_rank_selftest", rccl_single_rank_selftest)
    guarded("sdxl_shaped_backbone", sdxl_shaped_backbone, args, dev)
    guarded("with_backbone", with_dummy_unet, args, dev)
    return out
