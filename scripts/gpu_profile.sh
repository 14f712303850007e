#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace + PMC passes of bench.py and of the
# c5_wan micro-benchmark; leaves only small text summaries under gpurun_out/ (the rocpd .db
# files are deleted: gpurun copies back at most 64 MiB).
set -u
R=$PWD
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
summ() { python $R/scripts/rocprof_summary.py "$@"; }

rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-shape --extras 0 > $OUT/c2_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c2/t_results.db > $OUT/c2_kernel_trace.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o t -- python $R/scripts/microbench_step.py c5_wan steady 50 > $OUT/c5_microbench_under_rocprof.log 2>&1
summ /tmp/p_c5/t_results.db > $OUT/c5_kernel_trace.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_c5b -o t -- python $R/bench.py --workload c5_wan --steps 4 --warmup 2 --no-cpu-baseline --extras 0 > $OUT/c5_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c5b/t_results.db > $OUT/c5_bench_kernel_trace.md 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_c2_$ctr -o t -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-large-shape --extras 0 > $OUT/c2_pmc_$ctr.log 2>&1
  summ /tmp/p_pmc_c2_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter" > $OUT/c2_pmc_$ctr.md
  for wl in c1:c1_sd15 c3:c3_sdxl_b4 c4:c4_flux c5:c5_wan; do
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_${wl%%:*}_$ctr -o t -- python $R/scripts/microbench_step.py ${wl#*:} steady 20 > $OUT/${wl%%:*}_pmc_$ctr.log 2>&1
    summ /tmp/p_pmc_${wl%%:*}_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter\|Mul" > $OUT/${wl%%:*}_pmc_$ctr.md
  done
done
rm -rf /tmp/p_*
cd $R
ls -la $OUT
# MFMA-busy evidence for the stand-in backbone (kept separate: MIOpen under --pmc FETCH_SIZE crashed rocprofv3 once)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o t -- python $R/scripts/unet_pass.py 4 > $OUT/unet_pmc_mfma.log 2>&1
summ /tmp/p_mfma/t_results.db --pmc 2>&1 | grep -A400 "counter | dispatches" > $OUT/unet_pmc_mfma.md
summ /tmp/p_mfma/t_results.db 2>&1 | head -25 > $OUT/unet_kernel_trace.md
rm -rf /tmp/p_mfma
cd $R
