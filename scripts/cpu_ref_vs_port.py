#!/usr/bin/env python3
"""How fast is the UNMODIFIED reference engine (/root/reference/src/LanPaint/lanpaint.py) next to the oracle port
bench.py times as `cpu_baseline`?  Runs only where /root/reference exists (the build container; the GPU box has no
copy), same inputs / stub backbone / schedule as bench.py's workload, one thread, interleaved passes, median of 5.
Writes profiles/rNN_cpu_reference_vs_port.json, which bench.py quotes next to its own port timing.

    python scripts/cpu_ref_vs_port.py [round=02] [workload=c2_sdxl]
"""
import json
import os
import platform
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from oracle.lanpaint_oracle import OracleLanPaint, TorchBackend  # noqa: E402

REF = "/root/reference"


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "02"
    wl = sys.argv[2] if len(sys.argv) > 2 else "c2_sdxl"
    sys.path.insert(0, REF)
    from src.LanPaint.lanpaint import LanPaint as RefLanPaint    # the reference, unmodified
    shape, flow, n_sig, n_think = bench.WORKLOADS[wl]
    sig_np = bench.flow_sigmas(n_sig) if flow else bench.karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))     # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, "cpu", tt)
    sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32) for s in sig_np]
    times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
    ratios = bench.euler_ratios(sig_list, len(shape))
    h = bench.HYPER
    model = bench.StubBackbone(flow)
    ref = RefLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=flow,
                      MinStepFrac=h["MinStepFrac"])
    port = OracleLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], is_flow=flow,
                          min_step_frac=h["MinStepFrac"], backend=TorchBackend())
    torch.set_num_threads(1)
    its = n_sig * n_think
    for eng in (ref, port):                                      # warm-up
        bench.schedule_pass(eng, x0, y, noise, mask, sig_list[:3], times_list[:3], ratios[:2], n_think)
    res = {"reference": [], "port": []}
    for _ in range(5):
        for name, eng in (("reference", ref), ("port", port)):
            t0 = time.perf_counter()
            bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
            res[name].append(its / (time.perf_counter() - t0))
    r, p = statistics.median(res["reference"]), statistics.median(res["port"])
    cpu = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), platform.processor())
    out = {"workload": wl, "threads": 1, "cpu": cpu, "host_cpus": os.cpu_count(), "torch": torch.__version__,
           "reference_it_per_s": r, "port_it_per_s": p, "port_over_reference": p / r,
           "reference_runs": res["reference"], "port_runs": res["port"],
           "note": "reference = /root/reference/src/LanPaint/lanpaint.py imported unmodified; port = oracle/lanpaint_oracle.py "
                   "on torch-CPU tensors (what bench.py's cpu_baseline times on the GPU box, where the reference is absent). "
                   "Full passes of the schedule, interleaved, median of 5."}
    path = os.path.join(ROOT, "profiles", f"r{rnd}_cpu_reference_vs_port.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
