#!/bin/bash
# SQ counters of the steady streaming launch on the final build (VALU instructions per wave after the one-multiply Philox round)
R=$PWD; O=$R/gpurun_out/r04_sq; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"
for spec in c5_wan:fp32:philox c5_wan:bf16:philox c5_wan:fp32:torch; do
  wl=${spec%%:*}; r=${spec#*:}; dt=${r%%:*}; rng=${r#*:}
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_sq_${wl}_${dt}_$rng -o t -- python $R/scripts/microbench_step.py $wl steady 20 $rng > $O/sq_${wl}_${dt}_$rng.log 2>&1
  python $R/scripts/rocprof_summary.py /tmp/p_sq_${wl}_${dt}_$rng/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp_step_kernel" > $O/sq_${wl}_${dt}_$rng.md
  echo "== $wl $dt $rng"; awk -F"|" "{print \$3, \$5}" $O/sq_${wl}_${dt}_$rng.md
done
rm -rf /tmp/p_sq_*
