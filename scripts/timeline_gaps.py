#!/usr/bin/env python3
"""Where does the GPU idle between the kernels of a sigma call?  Reads a rocprofv3 --kernel-trace result
(rocpd .db or kernel_trace.csv), orders the dispatches by start time and reports, per (previous kernel ->
next kernel) pair, the mean gap between the end of one and the start of the next, next to the mean
durations -- the launch-latency budget of a latency-bound workload (C2: 2.4 MB per launch).

    python scripts/timeline_gaps.py <results.db | kernel_trace.csv> [max_gap_us=200]
"""
import statistics
import sys
from collections import defaultdict

from rocprof_summary import rows_from_csv, rows_from_db


def tag(name):
    if "lp_step_kernel" in name:
        args = name.split("lp_step_kernel<")[1].split(">")[0].replace(" ", "")
        ph = args.split(",")[2]
        return {"17u": "replace", "49u": "replace+coef", "28u": "step_steady", "26u": "step_first", "20u": "step_last",
                "18u": "step_only", "0u": "step_generic"}.get(ph, "step<" + ph + ">")
    for key, t in (("lp_finalize", "finalize"), ("lp_sigma_times", "sigma_times"), ("lerp", "lerp(sampler)"),
                   ("MulFunctor", "stub_model"), ("lp_early", "earlystop"), ("lp_wmse", "wmse"), ("lp_ptr", "set_ptrs")):
        if key in name:
            return t
    return name.replace("void ", "")[:40]


def main():
    path = sys.argv[1]
    max_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
    rows = sorted(rows_from_db(path) if path.endswith(".db") else rows_from_csv(path), key=lambda r: r[1])
    gaps, durs = defaultdict(list), defaultdict(list)
    busy = idle = 0
    for prev, cur in zip(rows, rows[1:]):
        g = cur[1] - prev[2]
        if g > max_gap:          # a host-side pause (sync, set-up), not part of the steady pipeline
            continue
        gaps[(tag(prev[0]), tag(cur[0]))].append(g)
        durs[tag(cur[0])].append(cur[2] - cur[1])
        busy += cur[2] - cur[1]
        idle += max(g, 0)
    print(f"dispatches: {len(rows)}   busy {busy / 1e6:.2f} ms   idle between dispatches {idle / 1e6:.2f} ms "
          f"({100.0 * idle / max(1, busy + idle):.1f} % of busy+idle; gaps > {max_gap / 1e3:.0f} us excluded)\n")
    print("| kernel | dispatches | mean us | median us |")
    print("|---|---|---|---|")
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        print(f"| {k} | {len(v)} | {statistics.mean(v) / 1e3:.2f} | {statistics.median(v) / 1e3:.2f} |")
    print("\n| previous -> next | count | mean gap us | median gap us | total gap ms |")
    print("|---|---|---|---|---|")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 5:
            continue
        print(f"| {a} -> {b} | {len(v)} | {statistics.mean(v) / 1e3:.2f} | {statistics.median(v) / 1e3:.2f} | {sum(v) / 1e6:.3f} |")


if __name__ == "__main__":
    main()
