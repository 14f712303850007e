"""The bandwidth statement of the dominant kernel (the steady-state fused lp_step launch): algorithmic bytes per launch
(SURVEY.md 8d's per-element figure x elements), per-dispatch HIP-event timing on the launch stream, and look-ups of the
rocprofv3 measurements committed under profiles/ (kernel-trace mean per dispatch, PMC bytes per launch)."""
from __future__ import annotations

import ctypes
import glob
import json
import os
import re
import time

import numpy as np
import torch

from .workloads import HYPER, WORKLOADS, attach_mask_format, make_mask, times_from_sigma

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES_PER_EL_STEADY = 36          # read x_t,x0,x0_BIG,y,m,C ; write x_t,C,x_in (fp32, in-kernel RNG)
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable copy)
HBM_COPY_GBPS = 6290.0


# ---------------------------------------------------------------- committed profiles
def _profile_order(path):
    """Sort key of a profiles/ file: round number, then rNN_ (the round's final pass) AFTER rNNa_, rNNb_ (its earlier passes,
    kept for the box-to-box spread) -- plain string order would put `r04a_` behind `r04_`."""
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), m.group(2) == "", m.group(2)) if m else (-1, False, "")


def _latest_profile_json(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=_profile_order)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_traffic(workload):
    """HBM-side bytes per steady-state launch measured with rocprofv3 PMC counters (FETCH_SIZE and WRITE_SIZE in separate
    passes, gfx950 x2 read correction) -- a committed measurement (profiles/r*_pmc_traffic.json, produced by
    scripts/gpu_profile.sh), or None when no profile covers this workload.  (`bench.py --pmc` measures it live instead.)"""
    data, _f = _latest_profile_json("r*_pmc_traffic.json")
    try:
        entry = (data or {}).get(workload)
        return int(entry["traffic_bytes_per_launch"]) if entry else None
    except Exception:
        return None


def committed_profile(workload):
    """What profiles/ holds for this shape, as a cross-reference NEXT TO this run's own measurement (never folded into
    it): the rocprofv3 --kernel-trace mean per dispatch and the PMC bytes per launch, with the files they come from."""
    dur, f_dur = _latest_profile_json("r*_kernel_durations.json")
    pmc, f_pmc = _latest_profile_json("r*_pmc_traffic.json")
    d, t = (dur or {}).get(workload), (pmc or {}).get(workload)
    return {"rocprofv3_mean_launch_us": d.get("mean_us") if d else None, "rocprofv3_source": d.get("source") if d else None,
            "kernel_durations_file": f_dur, "pmc_traffic_bytes_per_launch": t.get("traffic_bytes_per_launch") if t else None,
            "pmc_traffic_file": f_pmc}


# ---------------------------------------------------------------- bytes model
def steady_bytes_per_launch(mask_np, mask_format, n_el, every_stream=False, model_dtype=None):
    """ALGORITHMIC bytes of one steady-state launch (POST_STEADY | PRE_HALF | EMIT), two figures:
      every_stream -- SURVEY.md 8(d)'s per-unit figure for this storage: read x_t, x0, x0_BIG, y, m, C and write x_t, C, x_in
                      for EVERY element (36 B with fp32 streams and the reference's fp32 mask; the mask's own width as used:
                      0.125 B bit-packed, 1 B as bytes; half-width heads / x_in with a bf16 backbone);
      required     -- the bytes THIS job needs, from the mask actually used: an inpaint element (m = 0) reads head 0 only, a
                      known one (m = 1) head 1 and y only (lanpaint.py:182-184 with m in {0, 1}).  It applies to the launches
                      that act on it -- the region-aware streaming kernels (bit-packed mask, 16 B per lane: more than 512 Ki
                      elements, not LP_FL_NO_REGION_SKIP); every other launch streams every operand and `required` equals
                      `every_stream`.
    `roofline.frac` is computed on `required`: a fraction of peak on bytes the kernel never has to move is not a bandwidth
    fraction."""
    half = model_dtype is not None
    head, xin = (2.0, 2.0) if half else (4.0, 4.0)
    m_b = {"bits": 0.125, "u8": 1.0}.get(mask_format, 4.0)
    every = 8.0 + 8.0 + xin + 2 * head + 4.0 + m_b            # x_t, C in; x_t, C out; x_in out; two heads; y; mask
    region_aware = mask_format == "bits" and n_el > 512 * 1024 and not every_stream
    required = every
    known_frac = None
    if mask_np is not None:
        known_frac = float(np.count_nonzero(np.asarray(mask_np) > 0.5)) / float(np.asarray(mask_np).size)
        if region_aware:
            required = 8.0 + 8.0 + xin + m_b + (1.0 - known_frac) * head + known_frac * (head + 4.0)
    return {"every_stream": every * n_el, "required": required * n_el, "bytes_per_element_every_stream": every,
            "bytes_per_element_required": required, "known_fraction": known_frac, "region_aware_launch": region_aware}


def roofline_fields(bytes_alg, duration_us, traffic, bytes_every_stream=None, traffic_source=None):
    """The bandwidth statement of one launch.  `bytes_alg`: the algorithmic bytes the launch has to move
    (steady_bytes_per_launch's `required`); `achieved` / `frac` = that / duration -- at most the rate the bytes really moved
    at, so a fraction of peak.  `frac_every_stream`: the same duration against SURVEY.md 8(d)'s every-operand figure (can
    exceed what HBM delivers when the kernel skips streams; never `frac`).  `frac_counter`: min(algorithmic, PMC) / duration."""
    achieved = bytes_alg / (duration_us * 1e-6) / 1e9
    moved = min(bytes_alg, traffic) if traffic else None
    counter = (moved / (duration_us * 1e-6) / 1e9) if moved else None
    every = (bytes_every_stream / (duration_us * 1e-6) / 1e9) if bytes_every_stream else None
    if traffic and traffic_source is None:
        traffic_source = "committed_constant: profiles/r*_pmc_traffic.json (rocprofv3 --pmc passes of the same launch, another box)"
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "frac_algorithmic": achieved / HBM_PEAK_GBPS, "frac_counter": (counter / HBM_PEAK_GBPS) if counter else None,
            "frac_every_stream": (every / HBM_PEAK_GBPS) if every else None,
            "every_stream_bytes_per_launch": bytes_every_stream,
            "frac_vs_6290": achieved / HBM_COPY_GBPS, "traffic": traffic, "traffic_source": traffic_source if traffic else None,
            "algorithmic_bytes_per_launch": bytes_alg, "duration_used_us": duration_us}


def steady_kernel_name(workload, rng, mask_format):
    """The instantiation lp_step dispatches the steady-state launch of this workload to (step_kernel.hip; template
    arguments: VEC, MODE (2 = bit-packed hard mask), PH (28 = POST_STEADY|PRE_HALF|EMIT), head dtype, RNG (0 Philox2x32,
    1 torch's Philox4x32 stream), ST (ATen's thread order past the grid cap), early stop)."""
    shape = WORKLOADS[workload][0]
    n_el = int(np.prod(shape))
    vec = 4 if n_el > 512 * 1024 else 1
    strided = rng == "torch" and vec == 4 and shape[0] == 1
    return (f"lp::lp_step_kernel<{vec}, {2 if mask_format == 'bits' else 0}, 28u, 4, {1 if rng == 'torch' else 0}, "
            f"{'true' if strided else 'false'}, 0>")


def shape_regime(n_el, streams=9):
    working_set = streams * 4 * n_el                  # the nine fp32 streams of the steady launch
    regime = ("past the 256 MiB Infinity Cache: every byte comes from / goes to HBM" if working_set > 2 * 256 * 2 ** 20 else
              "L3-resident: the working set fits the 256 MiB Infinity Cache, the rates are fabric-side, not DRAM-side"
              if working_set > 32 * 2 ** 20 else "cache resident (L2): launch-latency bound")
    return working_set, regime


# ---------------------------------------------------------------- a steady launch on synthetic buffers
def tune_from_env(_cabi):
    """lp_step_desc.tune for the micro-benchmark scripts (scripts/microbench_*.py): the library does not look at the
    environment, the scripts translate LANPAINT_AMD_TUNE_* into the descriptor field."""
    t = 0
    vec = os.environ.get("LANPAINT_AMD_TUNE_VEC")
    if vec == "1":
        t |= _cabi.LP_TUNE_VEC1
    elif vec == "4":
        t |= _cabi.LP_TUNE_VEC4
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_DECIDE"):
        t |= _cabi.LP_TUNE_ES_NO_DECIDE
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_FOLD"):
        t |= _cabi.LP_TUNE_ES_NO_FOLD
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_ATOMICS"):
        t |= _cabi.LP_TUNE_ES_NO_ATOMICS
    return t


def standalone_step(_cabi, workload, dev, phase=None, model_dtype=None, mask_kind=None, mask_format="bits", rng="philox"):
    """A self-contained steady-state lp_step launch on synthetic buffers of `workload`'s shape: (descriptor, tensors to keep
    alive, element count).  rng="torch": the launch generates the device generator's randn stream (the engine's default)."""
    lib = _cabi.load()
    shape, flow, _, _ = WORKLOADS[workload]
    n_el, rows = int(np.prod(shape)), shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    bufs = {k: torch.randn(shape, device=dev, generator=g) for k in ("x", "y", "noise", "x_t", "C", "x0", "x0b", "x_in")}
    if model_dtype is not None:          # a half-precision backbone: its two heads arrive, and x_in leaves, in that dtype
        for k in ("x0", "x0b", "x_in"):
            bufs[k] = bufs[k].to(model_dtype)
    mask = torch.from_numpy(make_mask(shape, mask_kind)).to(dev)
    h = _cabi.LpHyper()
    h.lambda_, h.beta, h.step_size, h.min_step_frac = HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"], 0.0
    h.is_flow, h.one_plus_lambda = int(flow), 1.0 + HYPER["Lambda"]
    sig = torch.full((rows,), 0.7 if flow else 1.5, device=dev)
    ve, abt, _ = times_from_sigma(sig, flow)
    coef = torch.empty((rows, _cabi.LP_COEF_STRIDE), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _cabi.check(lib.lp_coeffs(ctypes.byref(h), ve.data_ptr(), 1, abt.data_ptr(), 1, sig.data_ptr(), 1, None, 0, None, 0, rows,
                              coef.data_ptr(), st))
    d = _cabi.LpStepDesc()
    d.n_el, d.el_per_row, d.rows = n_el, n_el // rows, rows
    d.phases = phase or (_cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT)
    mask = attach_mask_format(mask, mask_format)
    d.flags = (_cabi.LP_FL_FLOW if flow else 0) | (_cabi.LP_FL_NO_REGION_SKIP if os.environ.get("LANPAINT_AMD_NO_REGION_SKIP") else 0)
    d.replace_kind, d.lambda_, d.one_plus_lambda, d.beta = _cabi.LP_REPLACE_VE, h.lambda_, h.one_plus_lambda, h.beta
    d.step_size, d.noise_scale = h.step_size, 1.0
    d.coef, d.x, d.noise, d.y, d.mask = (coef.data_ptr(), bufs["x"].data_ptr(), bufs["noise"].data_ptr(),
                                         bufs["y"].data_ptr(), mask.data_ptr())
    if mask_format == "bits":
        d.mask, d.flags = mask._lp_bits.data_ptr(), d.flags | _cabi.LP_FL_MASK_BITS
    elif mask_format == "u8":
        d.mask, d.flags = mask._lp_u8.data_ptr(), d.flags | _cabi.LP_FL_MASK_U8
    d.x_t, d.C, d.x0, d.x0_big, d.x_in = (bufs[k].data_ptr() for k in ("x_t", "C", "x0", "x0b", "x_in"))
    if model_dtype is not None:
        half = model_dtype == torch.bfloat16
        d.flags |= (_cabi.LP_FL_X0_BF16 | _cabi.LP_FL_XIN_BF16) if half else (_cabi.LP_FL_X0_F16 | _cabi.LP_FL_XIN_F16)
    d.rng_seed = 1
    if rng == "torch":
        from lanpaint_amd.lanpaint import aten_randn_policy
        p = torch.cuda.get_device_properties(dev)
        d.rng_kind = _cabi.LP_RNG_TORCH
        d.rng_bg, d.rng_inc = aten_randn_policy(n_el, p.multi_processor_count, p.max_threads_per_multi_processor)
    d.tune = tune_from_env(_cabi)
    keep = (bufs, mask, coef, sig, ve, abt)
    return d, keep, n_el


def graph_burst_us_per_launch(_cabi, workload, dev, reps=200, replays=20, every_stream=False, model_dtype=None, **kw):
    """Un-profiled steady-state cost of one launch: `reps` launches of the steady kernel captured in a
    hipGraph on synthetic buffers of the workload's shape, replayed; wall time / launches (kernel +
    the dependent-launch boundary; a bare torch elementwise kernel costs ~1.66 us this way)."""
    lib = _cabi.load()
    d, keep, _n = standalone_step(_cabi, workload, dev, model_dtype=model_dtype, **kw)
    if every_stream:
        d.flags |= _cabi.LP_FL_NO_REGION_SKIP

    def launches(n):
        st = torch.cuda.current_stream(dev).cuda_stream
        for k in range(n):
            d.rng_offset = k
            _cabi.check(lib.lp_step(ctypes.byref(d), st))

    launches(5)
    torch.cuda.synchronize(dev)
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(graph, stream=side):
        launches(reps)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize(dev)
    us = (time.perf_counter() - t0) / (replays * reps) * 1e6
    del keep
    return us


def warm_burst(_cabi, lib, d, stream, dev, seconds):
    """Untimed launches of the very launch about to be timed, for `seconds`: the first launches of a process (or after
    an idle gap) run while the chip's clocks are still ramping."""
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(32):
            d.rng_offset = n
            n += 1
            _cabi.check(lib.lp_step(ctypes.byref(d), stream))
        torch.cuda.synchronize(dev)
    return n


def timed_burst(_cabi, lib, d, stream, dev, launches, lead=16):
    """Per-dispatch durations (s) of `launches` back-to-back launches from ONE host call (`lead` more in front, not
    counted: they start on an idle chip).  hipExtLaunchKernelGGL start / stop events bound to each dispatch on `stream`."""
    timers = (ctypes.c_void_p * (launches + lead))()
    for k in range(launches + lead):
        t = ctypes.c_void_p()
        _cabi.check(lib.lp_timer_create(ctypes.byref(t)))
        timers[k] = t
    d.rng_offset = 100
    torch.cuda.synchronize(dev)
    _cabi.check(lib.lp_step_timed_burst(ctypes.byref(d), stream, timers, launches + lead))   # one host call: GPU stays busy
    torch.cuda.synchronize(dev)
    durs = []
    for t in timers:
        ns = ctypes.c_double()
        _cabi.check(lib.lp_timer_elapsed_ns(ctypes.c_void_p(t), ctypes.byref(ns)))
        durs.append(ns.value * 1e-9)
        lib.lp_timer_destroy(ctypes.c_void_p(t))
    return np.asarray(durs[lead:])


def measure_steady_launch(_cabi, dev, workload="c5_wan", launches=100, every_stream=False, warm_s=0.3, model_dtype=None,
                          mask_kind=None, mask_format="bits", rng="philox", traffic=None, traffic_source=None):
    """The steady-state kernel of `workload`'s shape launched back to back through lp_step_timed_burst after a warm burst of
    the same launch: the full roofline block of that launch (live event timing; `traffic` = the committed PMC bytes unless a
    live figure is handed in)."""
    lib = _cabi.load()
    d, keep, n_el = standalone_step(_cabi, workload, dev, model_dtype=model_dtype, mask_kind=mask_kind, mask_format=mask_format,
                                    rng=rng)
    if every_stream:
        d.flags |= _cabi.LP_FL_NO_REGION_SKIP
    st = torch.cuda.current_stream(dev).cuda_stream
    warmed = warm_burst(_cabi, lib, d, st, dev, warm_s)
    durs = timed_burst(_cabi, lib, d, st, dev, launches)
    shape = WORKLOADS[workload][0]
    nbytes = steady_bytes_per_launch(make_mask(shape, mask_kind), mask_format, n_el, every_stream=every_stream,
                                     model_dtype=model_dtype)
    del keep
    prof_key = workload + ("_every_stream" if every_stream else "") + ("_bf16" if model_dtype is not None else "") \
        + ("_torch" if rng == "torch" else "")
    event_us = float(durs.mean()) * 1e6
    working_set, regime = shape_regime(n_el)
    if traffic is None:
        traffic = pmc_traffic(prof_key)
        if traffic is None and rng == "torch":            # the noise generator does not change the bytes a launch moves
            traffic = pmc_traffic(prof_key[:-len("_torch")])
    out = roofline_fields(nbytes["required"], event_us, traffic, nbytes["every_stream"], traffic_source)
    out.update({"bytes_model": nbytes, "workload": f"{workload}: latent {'x'.join(map(str, shape))}, steady-state lp_step back to back"
                            + (", every operand streamed (LP_FL_NO_REGION_SKIP)" if every_stream else "")
                            + (", bf16 heads in / bf16 x_in out" if model_dtype is not None else ""),
                "kernel": steady_kernel_name(workload, rng, mask_format), "rng": rng, "mask_format": mask_format,
                "regime": regime, "working_set_bytes": working_set,
                "mean_launch_us": event_us, "median_launch_us": float(np.median(durs)) * 1e6,
                "min_launch_us": float(durs.min()) * 1e6, "launches_timed": int(durs.size), "warm_burst_s": warm_s,
                "warm_burst_launches": warmed, "committed_profile": committed_profile(prof_key),
                "timer": "hipExtLaunchKernelGGL start/stop events per dispatch (kernel begin->end) on the launch stream, THIS run"})
    rp = out["committed_profile"].get("rocprofv3_mean_launch_us")
    out["frac_rocprofv3"] = (nbytes["required"] / (rp * 1e-6) / 1e9 / HBM_PEAK_GBPS) if rp else None
    return out


def compact_roofline(r):
    """The part of a roofline block that goes on the headline line (the rest lives in the side-car file)."""
    if not isinstance(r, dict) or "frac" not in r:
        return r
    keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
            "duration_used_us", "kernel", "launches_timed", "frac_rocprofv3", "regime")
    out = {k: r.get(k) for k in keys}
    out["rocprofv3_mean_launch_us"] = (r.get("committed_profile") or {}).get("rocprofv3_mean_launch_us")
    out["rocprofv3_source"] = (r.get("committed_profile") or {}).get("rocprofv3_source")
    return out
