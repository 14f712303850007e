#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace + PMC passes of bench.py and of the steady-kernel
# micro-benchmark; leaves only small text summaries under gpurun_out/profiles/ (the rocpd .db files are deleted:
# gpurun copies back at most 64 MiB).  scripts/collect_profiles.py NN then files them under profiles/rNN_*.
set -u
R=$PWD
OUT=$R/gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
summ() { python $R/scripts/rocprof_summary.py "$@"; }

# ---- kernel traces (durations) -----------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o t -- python $R/bench.py --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline --no-summary > $OUT/c2_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c2/t_results.db > $OUT/c2_kernel_trace.md 2>&1
python $R/scripts/timeline_gaps.py /tmp/p_c2/t_results.db > $OUT/c2_timeline_gaps.md 2>&1
# (the default bench is the drop-in engine: the reference's noise stream; the same trace with the Philox2x32 engine beside it)
rocprofv3 --kernel-trace --stats -d /tmp/p_c2p -o t -- python $R/bench.py --rng philox --graph 1 --mask-format bits --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline --no-summary > $OUT/c2_philox_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c2p/t_results.db > $OUT/c2_philox_kernel_trace.md 2>&1
for wl in c3_sdxl_b4:c3 c5_wan:c5 x_wan_b16:xwanb16; do
  rocprofv3 --kernel-trace --stats -d /tmp/p_${wl#*:} -o t -- python $R/scripts/microbench_step.py ${wl%%:*} steady 50 > $OUT/${wl#*:}_microbench_under_rocprof.log 2>&1
  summ /tmp/p_${wl#*:}/t_results.db > $OUT/${wl#*:}_kernel_trace.md 2>&1
done
# the reference's noise stream (rng="torch", the engine default) at the video latent: ATen-ordered generation, LDS transpose (round 5)
rocprofv3 --kernel-trace --stats -d /tmp/p_c5_torch -o t -- python $R/scripts/microbench_step.py c5_wan steady 50 torch > $OUT/c5_torch_microbench_under_rocprof.log 2>&1
summ /tmp/p_c5_torch/t_results.db > $OUT/c5_torch_kernel_trace.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_c5b -o t -- python $R/bench.py --workload c5_wan --steps 4 --warmup 2 --repeats 0 --no-cpu-baseline --no-summary > $OUT/c5_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c5b/t_results.db > $OUT/c5_bench_kernel_trace.md 2>&1
# the same past-L3 launch with every operand streamed regardless of the mask (region-aware streams off)
LANPAINT_AMD_NO_REGION_SKIP=1 rocprofv3 --kernel-trace --stats -d /tmp/p_xnoskip -o t -- python $R/scripts/microbench_step.py x_wan_b16 steady 50 > $OUT/xwanb16_noskip_microbench_under_rocprof.log 2>&1
summ /tmp/p_xnoskip/t_results.db > $OUT/xwanb16_noskip_kernel_trace.md 2>&1

# bf16 heads in / bf16 x_in out (30 B / element): the production storage widths at the two bandwidth-bound shapes
for wl in c5_wan:c5_bf16 x_wan_b16:xwanb16_bf16; do
  LANPAINT_AMD_BENCH_DTYPE=bf16 rocprofv3 --kernel-trace --stats -d /tmp/p_${wl#*:} -o t -- python $R/scripts/microbench_step.py ${wl%%:*} steady 50 > $OUT/${wl#*:}_microbench_under_rocprof.log 2>&1
  summ /tmp/p_${wl#*:}/t_results.db > $OUT/${wl#*:}_kernel_trace.md 2>&1
done

# ---- HBM-side traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), no other trace domain ----------
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_c2_$ctr -o t -- python $R/bench.py --steps 2 --warmup 1 --repeats 0 --graph 0 --no-cpu-baseline --no-summary > $OUT/c2_pmc_$ctr.log 2>&1
  summ /tmp/p_pmc_c2_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter" > $OUT/c2_pmc_$ctr.md
  for wl in c1:c1_sd15 c3:c3_sdxl_b4 c4:c4_flux c5:c5_wan xwanb16:x_wan_b16; do
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_${wl%%:*}_$ctr -o t -- python $R/scripts/microbench_step.py ${wl#*:} steady 20 > $OUT/${wl%%:*}_pmc_$ctr.log 2>&1
    summ /tmp/p_pmc_${wl%%:*}_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter\|Mul" > $OUT/${wl%%:*}_pmc_$ctr.md
  done
  for wl in c5_bf16:c5_wan xwanb16_bf16:x_wan_b16; do
    LANPAINT_AMD_BENCH_DTYPE=bf16 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_${wl%%:*}_$ctr -o t -- python $R/scripts/microbench_step.py ${wl#*:} steady 20 > $OUT/${wl%%:*}_pmc_$ctr.log 2>&1
    summ /tmp/p_pmc_${wl%%:*}_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter\|Mul" > $OUT/${wl%%:*}_pmc_$ctr.md
  done
  LANPAINT_AMD_NO_REGION_SKIP=1 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_xnoskip_$ctr -o t -- python $R/scripts/microbench_step.py x_wan_b16 steady 20 > $OUT/xwanb16_noskip_pmc_$ctr.log 2>&1
  summ /tmp/p_pmc_xnoskip_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter\|Mul" > $OUT/xwanb16_noskip_pmc_$ctr.md
done
rm -rf /tmp/p_*
cd $R
ls -la $OUT | head -60
