#!/usr/bin/env python3
"""Host-side cost of the launch primitives a sigma call is made of (MI355X box): an eager kernel launch through
ctypes, hipGraphLaunch as a function of node count, torch's small ops, stream lookup.  Feeds DESIGN.md section 6."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import _cabi                  # noqa: E402

dev = torch.device("cuda", 0)
lib = _cabi.load()
d, keep, n_el = bench.standalone_step(_cabi, "c2_sdxl", dev)
st = torch.cuda.current_stream(dev).cuda_stream


def host_us(fn, n=200, batch=20):
    """mean host time of fn() with the queue drained every `batch` calls (no back-pressure)"""
    for _ in range(5):
        fn()                   # lazy module load / first-call compilation is not launch cost
    tot = 0.0
    for _ in range(n // batch):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    return tot / (n // batch * batch) * 1e6


print(f"lp_step eager launch via ctypes: {host_us(lambda: lib.lp_step(ctypes.byref(d), st)):.2f} us")
x = keep[0]["x"]
print(f"torch x*0.9: {host_us(lambda: x * 0.9):.2f} us")
r = torch.full((1, 1, 1, 1), 0.5, device=dev)
print(f"torch.lerp(a,b,w): {host_us(lambda: torch.lerp(x, x, r)):.2f} us")
print(f"torch.empty_like: {host_us(lambda: torch.empty_like(x)):.2f} us")
print(f"torch.cuda.current_stream(dev).cuda_stream: {host_us(lambda: torch.cuda.current_stream(dev).cuda_stream):.2f} us")
print(f"torch._C._cuda_getCurrentRawStream(0): {host_us(lambda: torch._C._cuda_getCurrentRawStream(0)):.2f} us")
print(f"x.data_ptr(): {host_us(lambda: x.data_ptr()):.3f} us")
for nodes in (1, 6, 12, 13, 24):
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(g, stream=side):
        s2 = torch.cuda.current_stream(dev).cuda_stream
        for k in range(nodes):
            d.rng_offset = k
            lib.lp_step(ctypes.byref(d), s2)
    raw = int(g.raw_cuda_graph_exec())
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    t_replay = host_us(g.replay)
    t_raw = host_us(lambda: hip.hipGraphLaunch(raw, st))
    # GPU-side: wall per replay when pipelined
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        hip.hipGraphLaunch(raw, st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e6
    print(f"graph of {nodes:2d} step launches: torch replay() {t_replay:.2f} us host, raw hipGraphLaunch {t_raw:.2f} us host, "
          f"pipelined wall {wall:.2f} us per launch of the graph ({wall / nodes:.2f} us/node)")

# pipelined wall of eager launches (host-paced) for comparison
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    lib.lp_step(ctypes.byref(d), st)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"2000 eager lp_step launches: host {t_host / 2000 * 1e6:.2f} us each, wall {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us each")
