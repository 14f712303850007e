#!/bin/bash
# round 4, GPU session 4: uniform-wave arithmetic + scalar Philox key -- parity, then the streaming launch fp32 / bf16
R=$PWD; O=$R/gpurun_out/r04_s4; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_bench_selfspawn.py --ignore=tests/test_gpu_two_ranks.py 2>&1 | grep -E "passed|failed|Error|assert" | head -8
for wl in c5_wan x_wan_b16 c3_sdxl_b4; do for dt in fp32 bf16; do
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 120 python scripts/microbench_step.py $wl steady 2>&1 | grep -v amdgpu.ids | tail -1
done; done | tee $O/microbench_uni.log
export TMPDIR=/tmp; cd /tmp
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"
for spec in c5_wan:fp32 x_wan_b16:fp32 x_wan_b16:bf16; do
  wl=${spec%%:*}; dt=${spec#*:}
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_sq_${wl}_$dt -o t -- python $R/scripts/microbench_step.py $wl steady 20 > $O/sq_${wl}_$dt.log 2>&1
  python $R/scripts/rocprof_summary.py /tmp/p_sq_${wl}_$dt/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp_step_kernel" > $O/sq_${wl}_$dt.md
  echo "== $wl $dt"; awk -F"|" "{print \$3, \$5}" $O/sq_${wl}_$dt.md
done
rm -rf /tmp/p_sq_*
