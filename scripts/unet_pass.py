#!/usr/bin/env python3
"""A few sigma calls of the HIP engine in front of the random-init dummy UNet (bf16), eager launches:
the workload the MFMA-busy PMC pass profiles (scripts/gpu_profile.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from lanpaint_amd import LanPaint                        # noqa: E402
from tests.dummy_unet import DummyUNetBackbone           # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    shape, flow, n_sig, n_think = bench.WORKLOADS["c1_sd15"]
    sig_np = bench.karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
    net = DummyUNetBackbone(dev)
    eng = LanPaint(net, n_think, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=0)
    x = x0.clone()
    for s in sig_np[: int(sys.argv[1]) if len(sys.argv) > 1 else 4]:
        sg = torch.full((1,), float(s), device=dev)
        den = eng(x, y, noise, sg, mask, bench.times_from_sigma(sg, flow), None, 0)
    torch.cuda.synchronize()
    print("ok", bool(torch.isfinite(den).all()), net.calls, "backbone calls")


if __name__ == "__main__":
    main()
