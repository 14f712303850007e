#!/usr/bin/env python3
"""Where a think-step launch spends its time ON the chip: shader-clock stamps (s_memtime) taken inside the kernel by
thread 0 of the first and of the last block, in a library built with -DLP_SHADER_CLOCK (python -m lanpaint_amd.build
--shader-clock, done here; the product library carries none of it).  Complements rocprofv3, which only sees a
launch from outside (begin/end).  Prints, for the plain steady launch and for the early-stop (gated, folded) launch of
one workload, the time from kernel entry to each stamp next to the wall time per launch of the replayed graph.

    python scripts/shader_clock.py [workload=c2_sdxl]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lanpaint_amd import build as lpbuild            # noqa: E402

os.environ["LANPAINT_AMD_LIB"] = lpbuild.build(shader_clock=True, verbose=False)

import torch                                         # noqa: E402
import bench                                         # noqa: E402
from lanpaint_amd import _cabi                       # noqa: E402
from lanpaint_amd.lanpaint import _DeviceStop        # noqa: E402

STAMPS = ["kernel entry", "operand loads issued", "noise generated", "operands arrived", "stop verdict formed",
          "arithmetic done", "stores issued", "block sums written"]


def run(wl, early_stop, half=False, rng="philox"):
    dev = torch.device("cuda", 0)
    lib = _cabi.load()
    steady = _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT
    d, keep, n_el = bench.standalone_step(_cabi, wl, dev, steady, model_dtype=torch.bfloat16 if half else None)
    clk = torch.zeros(32, dtype=torch.float64, device=dev)
    d.clk_out = clk.data_ptr()
    if early_stop:
        ds = _DeviceStop(keep[0]["x_t"], 64)
        d.flags |= _cabi.LP_FL_ES | _cabi.LP_FL_ES_GATED
        d.es, d.es_partials, d.es_host = ds.state.data_ptr(), ds.partials.data_ptr(), None
        d.es_xte = ds.x_te.data_ptr()
        for k in range(3):
            d.es_x0s[k] = ds.x0s[k].data_ptr()
        d.es_threshold, d.es_patience_eff, d.es_index, d.es_n_steps = 1e-30, 2, 1, 64
    reps = 50
    if rng == "torch":                 # the device generator's randn stream reproduced in the kernel (the engine's default)
        from lanpaint_amd import LanPaint
        d.rng_kind, d.rng_seed = _cabi.LP_RNG_TORCH, 1234
        d.rng_bg, d.rng_inc = LanPaint._randn_policy(dev, n_el)

    def launches():
        st = torch.cuda.current_stream(dev).cuda_stream
        for k in range(reps):
            d.rng_offset = k if rng != "torch" else 2 * k * d.rng_inc
            _cabi.check(lib.lp_step(ctypes.byref(d), st))

    launches()
    torch.cuda.synchronize()
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(g, stream=side):
        launches()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (40 * reps) * 1e6
    h = clk.cpu().numpy()
    label = ("early-stop launch (gated, verdict folded in)" if early_stop else "plain steady launch") + (", bf16 heads in / bf16 x_in out" if half else "") + \
        (", reference noise stream (Philox4x32 + library Box-Muller)" if rng == "torch" else "")
    print(f"\n{wl}, {label}: {us:.2f} us per launch in a replayed graph of {reps} (instrumented build)")
    print(f"  the last block entered {(h[30] - h[14]) * 10.0:.0f} ns after the first one (s_memrealtime, 10 ns ticks): the time the "
          f"dispatcher needs to start the launch's {d.el_per_row * d.rows // (4 if n_el > 512 * 1024 else 1) // 256} blocks")
    for name, o in (("first block", h[0:16]), ("last block", h[16:32])):
        last = max(o[:len(STAMPS)])
        ns_per_tick = o[15] * 10.0 / last if last else 0.0          # s_memrealtime counts 100 MHz
        print(f"  {name}: {ns_per_tick:.3f} ns per shader-clock tick")
        for k in range(1, len(STAMPS)):
            if o[k]:
                print(f"    {STAMPS[k]:24s} {o[k] * ns_per_tick:7.0f} ns after entry")


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2_sdxl"
    if len(sys.argv) > 2 and sys.argv[2] == "bf16":          # fp32 heads against bf16 heads, plain launch
        run(wl, False)
        run(wl, False, half=True)
    elif len(sys.argv) > 2 and sys.argv[2] == "torch":       # in-kernel Philox2x32 against the reference's noise stream, plain launch
        run(wl, False)
        run(wl, False, rng="torch")
    else:
        run(wl, False)
        run(wl, True)
