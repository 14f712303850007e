#!/usr/bin/env python3
"""Copy the summaries scripts/gpu_profile.sh left under gpurun_out/profiles/ into profiles/rNN_* and
derive profiles/rNN_pmc_traffic.json (per-launch HBM-side bytes of the steady-state lp_step kernel).

    python scripts/collect_profiles.py 01
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles")
DST = os.path.join(ROOT, "profiles")
N_EL = {"c1": ("c1_sd15", 16384), "c2": ("c2_sdxl_torch", 65536), "c2_philox": ("c2_sdxl", 65536), "c3": ("c3_sdxl_b4", 262144), "c4": ("c4_flux", 65536),
        "c5": ("c5_wan", 2096640), "xwanb16": ("x_wan_b16", 33546240), "xwanb16_noskip": ("x_wan_b16_every_stream", 33546240),
        "c5_bf16": ("c5_wan_bf16", 2096640), "xwanb16_bf16": ("x_wan_b16_bf16", 33546240), "c5_torch": ("c5_wan_torch", 2096640)}
BYTES_PER_EL = {"c5_wan_bf16": 30, "x_wan_b16_bf16": 30}          # bf16 x0, x0_BIG in, bf16 x_in out; 36 otherwise


def steady_duration(path):
    """(mean us, dispatches) of the steady lp_step kernel in a kernel-trace summary (scripts/rocprof_summary.py)."""
    for line in open(path):
        m = re.match(r"\| `lp::lp_step_kernel<\d, \w+, 28u[^`]*` \| [^|]* \| \d+ \| (\d+) \| [0-9.]+ \| ([0-9.]+) \|", line)
        if m:
            return float(m.group(2)), int(m.group(1))
    return None


def steady_mean(path):
    """mean-per-dispatch (KB) of the steady lp_step kernel (PH = 28) in a --pmc summary: the instantiation with the MOST
    dispatches (a drop-in job's first sigma call still streams the fp32 mask through another instantiation before the engine
    packs it)."""
    best = None
    for line in open(path):
        m = re.match(r"\| `lp::lp_step_kernel<(\d), \w+, 28u[^>]*>.*?` \| (\w+) \| (\d+) \| ([0-9.]+) \|", line)
        if m and (best is None or int(m.group(3)) > best[2]):
            best = (int(m.group(1)), float(m.group(4)), int(m.group(3)))
    return best


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
    os.makedirs(DST, exist_ok=True)
    for f in sorted(os.listdir(SRC)):
        if f.endswith(".md"):
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f"r{rnd}_{f}"))
        elif f.endswith("_under_rocprof.json.log"):
            lines = [ln for ln in open(os.path.join(SRC, f)) if ln.startswith("{")]
            if lines:
                open(os.path.join(DST, f"r{rnd}_{f[:-4]}"), "w").write(lines[-1])
    traffic = {"_doc": "HBM-side bytes per launch of the steady-state lp_step kernel from rocprofv3 PMC passes "
                       "(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, profiles/r%s_*_pmc_*.md; KB per dispatch). "
                       "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read "
                       "(MI355X_MICROARCH.md, HBM section; re-checked in the same runs on torch's fp32 x*0.9 kernel, "
                       "which reads N*4 B and writes N*4 B) -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024." % rnd}
    for tag, (wl, n_el) in N_EL.items():
        fe = steady_mean(os.path.join(SRC, f"{tag}_pmc_FETCH_SIZE.md")) if os.path.exists(os.path.join(SRC, f"{tag}_pmc_FETCH_SIZE.md")) else None
        wr = steady_mean(os.path.join(SRC, f"{tag}_pmc_WRITE_SIZE.md")) if os.path.exists(os.path.join(SRC, f"{tag}_pmc_WRITE_SIZE.md")) else None
        if not fe or not wr:
            continue
        tb = int(round((2 * fe[1] + wr[1]) * 1024))
        traffic[wl] = {"kernel": f"lp::lp_step_kernel<VEC={fe[0]},MODE_HARD,POST_STEADY|PRE_HALF|EMIT,X0W=4,RNG={'torch' if wl.endswith('_torch') else 'philox'}>", "FETCH_SIZE_KB": fe[1], "WRITE_SIZE_KB": wr[1],
                       "dispatches": fe[2], "traffic_bytes_per_launch": tb,
                       "algorithmic_bytes_per_launch": BYTES_PER_EL.get(wl, 36) * n_el,
                       "traffic_over_algorithmic": round(tb / (BYTES_PER_EL.get(wl, 36) * n_el), 4)}
    if "c2_sdxl_torch" in traffic and "c2_sdxl" not in traffic:       # (the noise generator does not change the bytes a launch moves)
        traffic["c2_sdxl"] = dict(traffic["c2_sdxl_torch"], note="measured on the torch-stream launch of the same shape")
    json.dump(traffic, open(os.path.join(DST, f"r{rnd}_pmc_traffic.json"), "w"), indent=1)
    durations = {"_doc": "mean per-dispatch duration of the steady-state lp_step kernel as rocprofv3 --kernel-trace measured it "
                         "(profiles/r%s_*_kernel_trace.md); bench.py quotes it (committed_profile) next to its own live event timing" % rnd}
    for tag, (wl, _n) in N_EL.items():
        path = os.path.join(SRC, f"{tag}_kernel_trace.md")
        got = steady_duration(path) if os.path.exists(path) else None
        if got:
            durations[wl] = {"mean_us": got[0], "dispatches": got[1], "source": f"r{rnd}_{tag}_kernel_trace.md"}
    json.dump(durations, open(os.path.join(DST, f"r{rnd}_kernel_durations.json"), "w"), indent=1)
    print(json.dumps({k: v.get("traffic_over_algorithmic") for k, v in traffic.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
