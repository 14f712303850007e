#!/usr/bin/env python3
"""cProfile of the sampler-facing path (KSamplerX0Inpaint -> engine) on the C2 shape with node defaults.

    python scripts/prof_node.py
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
from lanpaint_amd import nodes as lpn           # noqa: E402

dev = torch.device("cuda", 0)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
shape, flow, n_sig, n_think = bench.WORKLOADS["c2_sdxl"]
sig_np = bench.karras_sigmas(n_sig)
x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
ratios = bench.euler_ratios(sig_list, 4)
model = bench.StubBackbone(flow)
model.model_type = "EPS"
k = lpn.KSamplerX0Inpaint(model, torch.cat([tt(sig_np), torch.zeros(1, device=dev)]))
k.latent_image, k.noise = y, noise
H = bench.HYPER
k.PaintMethod = LanPaint(model, n_think, H["Friction"], H["Lambda"], H["Beta"], H["StepSize"], MinStepFrac=1.0,
                         rng="philox", philox_seed=0, graph=True)
k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
denoise_mask = 1.0 - mask
model_options = {}


def node_pass():
    x = x0.clone()
    for i in range(n_sig):
        den = k(x, sig_list[i], denoise_mask, model_options=model_options, seed=0)
        if i + 1 < n_sig:
            x = torch.lerp(den, x, ratios[i])
    return x


for _ in range(3):
    node_pass()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    node_pass()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / (10 * n_sig) * 1e6:.1f} us per sigma call")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    node_pass()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4500])
