#!/usr/bin/env python3
"""Same-process A/B of the node path (KSamplerX0Inpaint at C2, node defaults): wait-then-launch / speculated count / speculated +
sigma algebra folded into the replace launch, interleaved blocks of 40 schedule passes, 6 rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from lanpaint_amd import LanPaint, nodes as lpn
dev = torch.device("cuda", 0)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
shape, flow, n_sig, n_think = bench.WORKLOADS["c2_sdxl"]
sig_np = bench.karras_sigmas(n_sig)
x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
ratios = bench.euler_ratios(sig_list, 4)
model = bench.StubBackbone(flow); model.model_type = "EPS"
k = lpn.KSamplerX0Inpaint(model, torch.cat([tt(sig_np), torch.zeros(1, device=dev)]))
k.latent_image, k.noise = y, noise
k.PaintMethod = LanPaint(model, n_think, 15.0, 5.0, 1.0, 0.2, MinStepFrac=1.0, rng="philox", philox_seed=0, graph=True)
k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
dm, mo = 1.0 - mask, {}
def node_pass():
    x = x0.clone()
    for i in range(n_sig):
        den = k(x, sig_list[i], dm, model_options=mo, seed=0)
        if i + 1 < n_sig:
            x = torch.lerp(den, x, ratios[i])
for _ in range(5): node_pass()
torch.cuda.synchronize()
res = {}
for rnd in range(6):
    for name, spec, fold in (("wait", "0", 0), ("speculate", "1", 0), ("speculate+fold", "1", 1)):
        k._speculate = spec == "1"
        k._node_desc.fold_sigma = fold
        k._spec_misses, k._spec_hist = 0, []
        node_pass(); torch.cuda.synchronize()
        it0, t0 = k.PaintMethod.iterations_run, time.perf_counter()
        for _ in range(40): node_pass()
        torch.cuda.synchronize()
        res.setdefault(name, []).append((k.PaintMethod.iterations_run - it0) / (time.perf_counter() - t0))
for name, v in res.items():
    print(f"{name:16s} median {np.median(v)/1e3:.1f} k it/s   " + " ".join(f"{a/1e3:.1f}" for a in v))
