#!/usr/bin/env python3
"""Where the host time of one sigma call on the node path goes (C2, node defaults, drop-in engine): wall per call by inner-step
count, time inside lp_node_call (enqueue + wait for the device's verdict), Python before / after it, the sampler's update."""
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from benchkit.extras import build_node_sampler                 # noqa: E402
from lanpaint_amd import _cabi                                 # noqa: E402

dev = torch.device("cuda", 0)
args = bench.parse_args([])
k, node_pass, n_sig = build_node_sampler(args, dev)
for _ in range(8):
    node_pass()
torch.cuda.synchronize()
lib = _cabi.load()
real = lib.lp_node_call
seg = defaultdict(list)
state = {}


def timed_node_call(nd, stream):
    t0 = time.perf_counter_ns()
    rc = real(nd, stream)
    state["c"] = time.perf_counter_ns() - t0
    state["t_in"] = t0
    return rc


pm = k.PaintMethod
pm._lib.lp_node_call = timed_node_call            # (the engine calls self._lib.lp_node_call)
orig_call = k.__class__.__call__


def timed_call(self, x, sigma, denoise_mask, **kw):
    t0 = time.perf_counter_ns()
    out = orig_call(self, x, sigma, denoise_mask, **kw)
    t1 = time.perf_counter_ns()
    n = int(self._node_desc.n_eff)
    seg[n].append((state["t_in"] - t0, state["c"], t1 - state["t_in"] - state["c"], t1 - t0))
    return out


k.__class__.__call__ = timed_call
for _ in range(40):
    node_pass()
torch.cuda.synchronize()
print("inner steps | calls | python before lp_node_call | inside lp_node_call (enqueue + wait) | python after | whole __call__   (us, median)")
for n in sorted(seg):
    a = np.asarray(seg[n], dtype=np.float64) / 1e3
    print(f"{n:11d} | {len(a):5d} | {np.median(a[:, 0]):6.1f} | {np.median(a[:, 1]):6.1f} | {np.median(a[:, 2]):6.1f} | {np.median(a[:, 3]):6.1f}")
k.__class__.__call__ = orig_call
t0 = time.perf_counter()
for _ in range(40):
    node_pass()
torch.cuda.synchronize()
print(f"wall per sigma call (untimed loop): {(time.perf_counter() - t0) / (40 * n_sig) * 1e6:.1f} us")
