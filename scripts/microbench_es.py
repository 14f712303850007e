#!/usr/bin/env python3
"""In-graph cost of one early-stop iteration: LP_FL_ES step launch + its one-block decide kernel, against the plain
steady launch, on the same buffers (50 launches per graph, replayed).

    python scripts/microbench_es.py [workload=c2_sdxl]
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import _cabi                  # noqa: E402
from lanpaint_amd.lanpaint import _DeviceStop   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_sdxl"
dev = torch.device("cuda", 0)
lib = _cabi.load()
steady = _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT


def run(label, setup):
    d, keep, n_el = bench.standalone_step(_cabi, wl, dev, steady)
    extra = setup(d, keep)
    reps = 50

    def launches():
        st = torch.cuda.current_stream(dev).cuda_stream
        for k in range(reps):
            d.rng_offset = k
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
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"{wl} {label}: {(time.perf_counter() - t0) / (20 * reps) * 1e6:.2f} us per iteration", flush=True)
    del extra


def es_setup(gated, host=True, index=1):
    def f(d, keep):
        ds = _DeviceStop(keep[0]["x_t"], 64)
        d.flags |= _cabi.LP_FL_ES | (_cabi.LP_FL_ES_GATED if gated else 0)
        d.es, d.es_partials, d.es_host = ds.state.data_ptr(), ds.partials.data_ptr(), ds.mailbox.data_ptr() if host else None
        d.es_xte = ds.x_te.data_ptr()
        for k in range(3):
            d.es_x0s[k] = ds.x0s[k].data_ptr()
        d.es_threshold, d.es_patience_eff, d.es_index, d.es_n_steps = 1e-30, 2, index, 64
        return ds
    return f


run("plain steady launch", lambda d, keep: None)
run("ES step + decide (gated)", es_setup(True))
run("ES step + decide (watched: mailbox fence per iteration)", es_setup(False))
run("ES step + decide (gated, no mailbox at all)", es_setup(True, host=False))


def tuned(tune, **kw):
    inner = es_setup(True, **kw)

    def f(d, keep):
        ds = inner(d, keep)
        d.tune = tune
        return ds
    return f


# what a deferred-verdict loop would launch per iteration: the early-stop step with its sums and atomics but
# nothing that waits for a verdict -- here the unfolded launch without its decision kernel (run-time-phase kernel: an upper bound)
run("ES step, unfolded, NO decision kernel (upper bound of a deferred-verdict launch)",
    tuned(_cabi.LP_TUNE_ES_NO_FOLD | _cabi.LP_TUNE_ES_NO_DECIDE))
run("ES step, unfolded, no decision kernel, no atomics",
    tuned(_cabi.LP_TUNE_ES_NO_FOLD | _cabi.LP_TUNE_ES_NO_DECIDE | _cabi.LP_TUNE_ES_NO_ATOMICS))
run("ES step, unfolded, WITH decision kernel", tuned(_cabi.LP_TUNE_ES_NO_FOLD))
