#!/bin/bash
# round 4, GPU session 3: is the streaming step kernel VALU-bound?  SQ counters of the steady launch, fp32 and bf16 heads
R=$PWD; O=$R/gpurun_out/r04_s3; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
summ() { python $R/scripts/rocprof_summary.py "$@"; }
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"
for spec in c5_wan:fp32 c5_wan:bf16 x_wan_b16:fp32 x_wan_b16:bf16; do
  wl=${spec%%:*}; dt=${spec#*:}
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_sq_${wl}_$dt -o t -- python $R/scripts/microbench_step.py $wl steady 20 > $O/sq_${wl}_$dt.log 2>&1
  summ /tmp/p_sq_${wl}_$dt/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp::\|counter" > $O/sq_${wl}_$dt.md
  echo "== $wl $dt"; cat $O/sq_${wl}_$dt.md | cut -c1-260
done
rm -rf /tmp/p_sq_*
