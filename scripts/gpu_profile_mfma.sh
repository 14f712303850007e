#!/bin/bash
# MFMA-busy evidence for a stand-in backbone next to the lp:: kernels (kept separate from gpu_profile.sh: MIOpen under
# --pmc FETCH_SIZE crashed rocprofv3 once).  Counters only with --kernel-trace, no other trace domain.
# Round 5: the SDXL-shaped bf16 stand-in at BASELINE configs[1]'s shape (tests/sdxl_standin.py) -- where MFMA is busy, that it is
# never busy in an lp:: kernel, and how a sigma call's GPU time splits between the backbone's kernels and the Langevin path's.
set -u
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o t -- python $R/scripts/sdxl_pass.py 6 > $OUT/sdxl_standin_pmc_mfma.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_mfma/t_results.db --pmc 2>&1 | grep -A600 "counter | dispatches" > $OUT/sdxl_standin_pmc_mfma_full.md
python - $OUT/sdxl_standin_pmc_mfma_full.md > $OUT/sdxl_standin_pmc_mfma.md <<'PY'
import re, sys, collections
rows = collections.defaultdict(dict)
for ln in open(sys.argv[1]):
    c = [x.strip() for x in ln.strip().strip("|").split("|")]
    if len(c) < 5 or c[1] not in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
        continue
    try:
        rows[c[0]][c[1]] = (int(c[2]), float(c[4]))
    except ValueError:
        pass
print("# MFMA busy per kernel, SDXL-shaped bf16 stand-in + LanPaint engine at 1x4x128x128 (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES)\n")
print("| kernel | dispatches | sum SQ_VALU_MFMA_BUSY_CYCLES | sum SQ_BUSY_CYCLES | MFMA busy / SQ busy |\n|---|---|---|---|---|")
fam = collections.defaultdict(lambda: [0.0, 0.0])
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]):
    m, b = v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0)), v.get("SQ_BUSY_CYCLES", (0, 0.0))
    f = "lp:: (Langevin path)" if "lp::" in k else "backbone / torch"
    fam[f][0] += m[1]; fam[f][1] += b[1]
    print(f"| {k[:110]} | {m[0]} | {m[1]:.0f} | {b[1]:.0f} | {m[1] / b[1] if b[1] else 0:.3f} |")
print("\n| family | sum SQ_VALU_MFMA_BUSY_CYCLES | sum SQ_BUSY_CYCLES |\n|---|---|---|")
for f, (m, b) in fam.items():
    print(f"| {f} | {m:.0f} | {b:.0f} |")
PY
rm -rf /tmp/p_mfma
# kernel-trace: where a sigma call's GPU time goes (eager launches, 6 sigma calls = 30 think iterations + 36 backbone passes)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o t -- python $R/scripts/sdxl_pass.py 6 > $OUT/sdxl_standin_kernel_trace.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_kt/t_results.db 2>&1 > $OUT/sdxl_standin_kernel_trace_full.md
python - $OUT/sdxl_standin_kernel_trace_full.md > $OUT/sdxl_standin_time_split.md <<'PY'
import sys
tot = {"lp:: (Langevin path)": [0.0, 0], "backbone / torch": [0.0, 0]}
for ln in open(sys.argv[1]):
    c = [x.strip() for x in ln.strip().strip("|").split("|")]
    if len(c) < 6 or not c[0].startswith("`"):
        continue
    try:
        calls, total_ms = int(c[3]), float(c[4])
    except ValueError:
        continue
    f = "lp:: (Langevin path)" if "lp::" in c[0] else "backbone / torch"
    tot[f][0] += total_ms; tot[f][1] += calls
s = sum(v[0] for v in tot.values())
print("# GPU time of 6 sigma calls (eager), SDXL-shaped bf16 stand-in + LanPaint engine at 1x4x128x128: kernel-trace totals by family\n")
print("| family | dispatches | total ms | share |\n|---|---|---|---|")
for f, (ms, n) in tot.items():
    print(f"| {f} | {n} | {ms:.3f} | {ms / s if s else 0:.4f} |")
PY
head -30 $OUT/sdxl_standin_kernel_trace_full.md > $OUT/sdxl_standin_kernel_trace.md
rm -rf /tmp/p_kt
cd $R
cat $OUT/sdxl_standin_time_split.md; tail -6 $OUT/sdxl_standin_pmc_mfma.md; grep "lp::" $OUT/sdxl_standin_pmc_mfma.md | cut -c1-170 | head
