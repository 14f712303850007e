#!/usr/bin/env python3
"""Is the default bench host-bound or GPU-bound?  Times (a) the host enqueue time of a schedule pass with no
synchronisation inside, (b) the wall time including the final sync, (c) a cProfile of the host side.

    python scripts/host_vs_gpu.py [workload] [graph 0|1]
"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import LanPaint               # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_sdxl"
graph = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
shape, flow, n_sig, n_think = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
sig_np = bench.flow_sigmas(n_sig) if flow else bench.karras_sigmas(n_sig)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
mode = sys.argv[3] if len(sys.argv) > 3 else "philox"
if mode == "philox":
    mask = bench.attach_mask_format(mask, "bits")
sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
ratios = bench.euler_ratios(sig_list, len(shape))
eng = (LanPaint(bench.StubBackbone(flow), n_think, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, rng="philox", graph=graph) if mode == "philox"
       else LanPaint(bench.StubBackbone(flow), n_think, 15.0, 5.0, 1.0, 0.2, False, flow))       # the drop-in engine
for _ in range(3):
    bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
calls = reps * n_sig
print(f"{wl} graph={graph} mode={mode}: host enqueue {t_host / calls * 1e6:.1f} us per sigma call, wall {t_all / calls * 1e6:.1f} us per sigma call "
      f"({'host' if t_host > 0.9 * t_all else 'GPU'}-bound)")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])

# host cost of ONE sigma call with an empty queue (sync before each call): no back-pressure possible
x = x0.clone()
ts = []
for i in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng(x, y, noise, sig_list[i % n_sig], mask, times_list[i % n_sig], None, 0, n_steps=n_think)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
ts = np.asarray(ts[10:]) * 1e6
print(f"host cost of one sigma call on an idle GPU: median {np.median(ts):.1f} us, min {ts.min():.1f} us")
t0 = time.perf_counter()
for i in range(200):
    torch.empty_like(x)
print(f"torch.empty_like: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us")
