#!/usr/bin/env python3
"""A few sigma calls of the HIP engine in front of the SDXL-shaped bf16 stand-in (tests/sdxl_standin.py) at BASELINE configs[1]'s
shape, eager launches -- the workload of the MFMA-busy PMC pass and of the kernel-trace time split (scripts/gpu_profile_mfma.sh).

    python scripts/sdxl_pass.py [sigma calls] [graph: 0 | 1]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from lanpaint_amd import LanPaint                        # noqa: E402
from tests.sdxl_standin import SDXLShapedBackbone        # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    graph = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    shape, flow, n_sig, n_think = bench.WORKLOADS["c2_sdxl"]
    sig_np = bench.karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
    mask = bench.attach_mask_format(mask, "bits")
    net = SDXLShapedBackbone(dev)
    eng = LanPaint(net, n_think, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=0, graph=graph, model_dtype=torch.bfloat16)
    x = x0.clone()
    for k in range(n_calls):
        s = sig_np[k % n_sig]
        sg = torch.full((1,), float(s), device=dev)
        den = eng(x, y, noise, sg, mask, bench.times_from_sigma(sg, flow), None, 0)
        x = torch.lerp(den, x, 0.9)
    torch.cuda.synchronize()
    print("ok", bool(torch.isfinite(den).all()), net.calls, "backbone calls", eng.iterations_run, "think iterations")


if __name__ == "__main__":
    main()
