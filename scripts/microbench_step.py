#!/usr/bin/env python3
"""Micro-benchmark of the fused lp_step kernel alone: R back-to-back launches of one phase
combination captured in a hipGraph, replayed; reports us per launch (kernel + the dependent
launch boundary) for the current LANPAINT_AMD_TUNE_* environment.

    python scripts/microbench_step.py c2_sdxl [steady|first|last|replace] [reps] [philox|torch] [box|temporal|blob]
(LANPAINT_AMD_NO_REGION_SKIP=1 streams every operand regardless of the mask; LANPAINT_AMD_BENCH_DTYPE=bf16: the two heads
arrive and x_in leaves as bf16, 30 algorithmic bytes per element.)
The FALLBACK (run-time-phase, PH = 0) kernels, round 5: LANPAINT_AMD_BENCH_MASK_FORMAT=u8 | f32 (a byte mask / the reference's
fp32 mask), LANPAINT_AMD_BENCH_SOFT=1 (fp32 mask with soft values on every 7th element: the general branch),
LANPAINT_AMD_BENCH_HOSTXI=1 (host-supplied noise tensors), LANPAINT_AMD_BENCH_AV=1 (AV pack: two-row table + indicator bits).
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import _cabi                  # noqa: E402

PH = {"steady": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
      "first": _cabi.LP_PH_POST_FIRST | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
      "last": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_EMIT,
      "replace": _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT,
      "replacec": _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT | _cabi.LP_PH_COEFFS}      # the engine's replace launch (table folded in)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2_sdxl"
    phase = sys.argv[2] if len(sys.argv) > 2 else "steady"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dev = torch.device("cuda", 0)
    lib = _cabi.load()
    mask_kind = sys.argv[5] if len(sys.argv) > 5 else None
    mask_format = os.environ.get("LANPAINT_AMD_BENCH_MASK_FORMAT", "bits")
    half = os.environ.get("LANPAINT_AMD_BENCH_DTYPE") == "bf16"
    d, keep, n_el = bench.standalone_step(_cabi, wl, dev, PH[phase], model_dtype=torch.bfloat16 if half else None,
                                          mask_kind=mask_kind, mask_format=mask_format)
    bufs = keep[0]
    if phase == "replacec":
        _b, _m, coef, sig, ve, abt = keep
        d.t_ve, d.t_abt, d.t_rsig, d.t_ve_stride, d.t_abt_stride, d.t_rsig_stride = ve.data_ptr(), abt.data_ptr(), sig.data_ptr(), 1, 1, 1
        d.coef_out = coef.data_ptr()
    rng = sys.argv[4] if len(sys.argv) > 4 else "philox"
    extra = []
    if os.environ.get("LANPAINT_AMD_BENCH_SOFT"):           # soft mask values: the per-element general branch (expf / expm1f)
        m = keep[1]
        m.view(-1)[::7] = 0.5
    if os.environ.get("LANPAINT_AMD_BENCH_HOSTXI"):         # recorded / explicit noise tensors: routed to the run-time-phase kernels
        xa, xb = torch.randn_like(bufs["x_t"]), torch.randn_like(bufs["x_t"])
        d.xi_post, d.xi_pre = xa.data_ptr(), xb.data_ptr()
        extra += [xa, xb]
    if os.environ.get("LANPAINT_AMD_BENCH_AV"):             # AV pack: video / audio time rows, the last third of every row is audio
        import numpy as np
        shape, flow, _, _ = bench.WORKLOADS[wl]
        rows = shape[0]
        ind = torch.zeros(shape, device=dev)
        ind.view(rows, -1)[:, (n_el // rows) * 2 // 3:] = 1.0
        bits = torch.empty(_cabi.mask_bits_bytes(n_el), dtype=torch.uint8, device=dev)
        _cabi.check(lib.lp_pack_mask(ind.data_ptr(), n_el, 0, bits.data_ptr(), None, torch.cuda.current_stream().cuda_stream))
        h = _cabi.LpHyper()
        h.lambda_, h.beta, h.step_size, h.min_step_frac = 5.0, 1.0, 0.2, 0.0
        h.is_flow, h.one_plus_lambda = int(flow), 6.0
        sig = torch.tensor([0.7, 0.4] * rows, device=dev)
        ve, abt, _ = bench.times_from_sigma(sig, flow)
        coef2 = torch.empty((2 * rows, _cabi.LP_COEF_STRIDE), device=dev)
        _cabi.check(lib.lp_coeffs(ctypes.byref(h), ve.data_ptr(), 1, abt.data_ptr(), 1, sig.data_ptr(), 1, None, 0, None, 0, 2 * rows,
                                  coef2.data_ptr(), torch.cuda.current_stream().cuda_stream))
        d.coef, d.av_bits, d.av_frac, d.flags = coef2.data_ptr(), bits.data_ptr(), 1.0 / 3.0, d.flags | _cabi.LP_FL_AV
        extra += [ind, bits, coef2, sig, ve, abt]
    if rng == "torch":                          # the device generator's randn stream reproduced in-kernel
        from lanpaint_amd import LanPaint
        d.rng_kind = _cabi.LP_RNG_TORCH
        d.rng_bg, d.rng_inc = LanPaint._randn_policy(dev, n_el)
        d.rng_seed = 1234

    def launches(n):
        s = torch.cuda.current_stream().cuda_stream
        for k in range(n):
            d.rng_offset = k if rng != "torch" else 2 * k * d.rng_inc
            _cabi.check(lib.lp_step(ctypes.byref(d), s))

    launches(5)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph, stream=side):
        launches(reps)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rep = 20
    for _ in range(n_rep):
        graph.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (n_rep * reps) * 1e6
    # reference point: a plain torch elementwise kernel of the same size, same harness
    g2 = torch.cuda.CUDAGraph()
    a = bufs["x"]
    with torch.cuda.graph(g2, stream=side):
        for _ in range(reps):
            b = a * 0.9
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rep):
        g2.replay()
    torch.cuda.synchronize()
    us_mul = (time.perf_counter() - t0) / (n_rep * reps) * 1e6
    bytes_ = ({"steady": 36, "first": 32, "last": 36, "replace": 24, "replacec": 24}[phase] - (6 if half and phase != "replace" else 0)) * n_el
    env = {k: v for k, v in os.environ.items() if k.startswith("LANPAINT_AMD_TUNE")}
    print(f"{wl} {phase} n_el={n_el} us/launch={us:.3f} ({bytes_ / us / 1e3:.0f} GB/s algorithmic) "
          f"heads={'bf16' if half else 'fp32'} rng={rng} mask={mask_kind or 'default'}/{mask_format}"
          f"{' soft' if os.environ.get('LANPAINT_AMD_BENCH_SOFT') else ''}{' hostxi' if os.environ.get('LANPAINT_AMD_BENCH_HOSTXI') else ''}"
          f"{' av' if os.environ.get('LANPAINT_AMD_BENCH_AV') else ''} region_skip={0 if os.environ.get('LANPAINT_AMD_NO_REGION_SKIP') else 1} torch_mul={us_mul:.3f}us finite={bool(torch.isfinite(bufs['x_t']).all())} env={env}", flush=True)


if __name__ == "__main__":
    main()
