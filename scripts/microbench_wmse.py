#!/usr/bin/env python3
"""lp_wmse_pair (the host stopper's metric: two kernels, partial sums + total) per call, 100 calls in a replayed graph.

    python scripts/microbench_wmse.py [workload=c5_wan]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import _cabi                  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5_wan"
shape = bench.WORKLOADS[wl][0]
dev = torch.device("cuda", 0)
lib = _cabi.load()
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(shape, device=dev, generator=g)
b = torch.randn(shape, device=dev, generator=g)
mask = (torch.rand(shape, device=dev, generator=g) > 0.5).float()
ring = (torch.rand(shape, device=dev, generator=g) > 0.9).float()
acc = torch.zeros(4, dtype=torch.float64, device=dev)
scratch = torch.zeros(4 * 1024, dtype=torch.float64, device=dev)
n = a.numel()


def calls(k):
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(k):
        _cabi.check(lib.lp_wmse_pair(a.data_ptr(), b.data_ptr(), mask.data_ptr(), ring.data_ptr(), n, acc.data_ptr(),
                                     scratch.data_ptr(), 1024, st))


calls(3)
torch.cuda.synchronize()
want = [float(((a - b) ** 2 * (1 - mask)).double().sum()), float((1 - mask).double().sum()),
        float(((a - b) ** 2 * ring).double().sum()), float(ring.double().sum())]
got = acc.tolist()
assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(got, want)), (got, want)
gr, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.graph(gr, stream=side):
    calls(100)
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gr.replay()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 1000 * 1e6
print(f"{wl} lp_wmse_pair n_el={n}: {us:.2f} us per call ({16 * n / us / 1e3:.0f} GB/s of the 16 B/element it reads)")
