"""cpu_baseline: the CPU port of the reference engine (oracle/lanpaint_oracle.py on torch-CPU fp32 tensors: the reference's
own eager ATen call structure, op for op) timed on THIS box's host cores over a bounded sample of the same workload.

The reference itself is pure Python and does not travel to the GPU box in any form; how the port's speed relates to the
unmodified reference is measured in the build container, where /root/reference exists, by scripts/cpu_ref_vs_port.py
(profiles/r*_cpu_reference_vs_port.json; quoted in the block as `port_over_reference_build_container`).  It is a reported
baseline, not the target."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .roofline import _latest_profile_json
from .workloads import HYPER, WORKLOADS, StubBackbone, Job, schedule_pass


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def usable_cpus():
    """CPUs this process can really use: its affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(workload, budget_s):
    """think-iterations/s of the port at 1 thread and at one multi-thread setting (the CPUs this process may use, stepping down
    a ladder when a team that large is pathological on the host); per setting one pass discarded, median of up to 5 timed
    passes.  Bounded: when the passes of the whole schedule would not fit `budget_s` / 2 per setting the sample is the first
    sigma calls of the schedule, and says so.  `value` / `cores`: the better of the settings."""
    from oracle.lanpaint_oracle import OracleLanPaint, TorchBackend
    shape, flow, n_sig, n_think = WORKLOADS[workload]
    job = Job(workload, "cpu")

    def make():
        return OracleLanPaint(StubBackbone(flow), HYPER["NSteps"], HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"],
                              HYPER["StepSize"], is_flow=flow, min_step_frac=HYPER["MinStepFrac"], backend=TorchBackend())

    def one_pass(eng, n_sigmas):
        t0 = time.perf_counter()
        schedule_pass(eng, job.x0, job.y, job.noise, job.mask, job.sig_list[:n_sigmas], job.times_list[:n_sigmas],
                      job.ratios[:max(0, n_sigmas - 1)], n_think)
        return time.perf_counter() - t0

    saved = torch.get_num_threads()
    n_cpu, n_usable = os.cpu_count() or 1, usable_cpus()
    per_threads, skipped, sample_sigmas = {}, {}, n_sig
    t_leg = time.perf_counter()
    ladder = [1] + [t for t in dict.fromkeys([n_usable, 64, 16, 8]) if 1 < t <= n_usable]
    try:
        for threads in ladder:
            if threads != 1 and any(t != 1 for t in per_threads):
                break                                           # one multi-thread setting has been sampled
            torch.set_num_threads(threads)
            share = budget_s / 2
            # what does ONE small elementwise op cost at this thread count?  (a sigma call is ~170 of them per think iteration;
            # on a many-core host the fork / join of a large team can cost more than the op)
            for _ in range(5):
                a = job.x0 * 0.9 + job.y
            t0 = time.perf_counter()
            for _ in range(20):
                a = job.x0 * 0.9 + job.y             # noqa: F841
            per_op = (time.perf_counter() - t0) / 40
            est_sigma = per_op * 170 * n_think
            if threads != 1 and 3 * est_sigma > share:
                skipped[str(threads)] = {"per_small_op_us": 1e6 * per_op, "estimated_s_per_sigma_call": est_sigma}
                continue
            eng = make()
            per_sigma = min(one_pass(eng, 1), one_pass(eng, 1))  # (the first also wakes the thread pool up)
            n_s = n_sig if 6 * n_sig * per_sigma <= share else max(1, int(share / (6 * per_sigma)))
            passes = 5 if 6 * n_s * per_sigma <= 2 * share else max(1, min(5, int(2 * share / (n_s * per_sigma)) - 1))
            sample_sigmas = min(sample_sigmas, n_s)
            one_pass(eng, n_s)                                  # discarded
            vals = [n_s * n_think / one_pass(eng, n_s) for _ in range(passes)]
            per_threads[threads] = {"median_it_s": float(np.median(vals)), "min_it_s": min(vals), "max_it_s": max(vals),
                                    "passes": len(vals), "sigma_calls_per_pass": n_s, "per_small_op_us": 1e6 * per_op}
    finally:
        torch.set_num_threads(saved)
    best = max(per_threads, key=lambda t: per_threads[t]["median_it_s"])
    whole = sample_sigmas == n_sig
    ref, ref_file = _latest_profile_json("r*_cpu_reference_vs_port.json")
    out = {"value": per_threads[best]["median_it_s"], "unit": "think-iterations/s", "cores": best, "kind": "port",
           "threads": {str(t): round(v["median_it_s"], 1) for t, v in sorted(per_threads.items())},
           "threads_detail": {str(t): v for t, v in sorted(per_threads.items())}, "threads_not_sampled": skipped or None,
           "host_cpus": n_cpu, "usable_cpus": n_usable, "cpu_model": cpu_model(), "leg_seconds": time.perf_counter() - t_leg,
           "sample": f"{'whole passes' if whole else f'the first {sample_sigmas} sigma calls'} of the {workload} schedule "
                     f"({n_sig} sigmas x {n_think}), stub backbone, oracle/lanpaint_oracle.py (op-for-op CPU port of the reference "
                     f"engine) on torch-CPU fp32 tensors; one pass discarded, median of <= 5"}
    if ref and ref.get("workload") == workload:
        out["port_over_reference_build_container"] = ref.get("port_over_reference")
        out["port_over_reference_source"] = ref_file
    return out


def compact_cpu(c):
    if not isinstance(c, dict) or "value" not in c:
        return c
    keys = ("value", "unit", "cores", "kind", "sample", "threads", "cpu_model", "host_cpus", "usable_cpus",
            "port_over_reference_build_container")
    return {k: c.get(k) for k in keys}
