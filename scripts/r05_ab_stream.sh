#!/bin/bash
# Round 5, VERDICT r04 next #4: same-box A/B of the streaming step launches, the round-4 tree (build/r04_tree, scripts/export_r04_tree.sh)
# against this one, alternately inside ONE gpurun call; then the SQ counters of the new steady launches (VALU instructions per wave).
R=$PWD; O=$R/gpurun_out/r05_ab; mkdir -p $O
run() {  # tree, args...
  local tree=$1; shift
  if [ $tree = r04 ]; then cd $R/build/r04_tree; else cd $R; fi
  "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-72
  cd $R
}
for round in 1 2; do
  for spec in "c5_wan steady" "c5_wan first" "c5_wan last" "c5_wan steady 200 torch" "c5_wan steady 200 philox box" "c5_wan steady 200 philox blob" "x_wan_b4 steady 100" "c2_sdxl steady" "c3_sdxl_b4 steady"; do
    for tree in r04 r05; do
      echo "round $round $tree [$spec] $(run $tree timeout 120 python scripts/microbench_step.py $spec)"
    done
  done
  for tree in r04 r05; do
    echo "round $round $tree [c5_wan steady bf16] $(LANPAINT_AMD_BENCH_DTYPE=bf16 run $tree timeout 120 python scripts/microbench_step.py c5_wan steady)"
    echo "round $round $tree [x_wan_b16 steady] $(run $tree timeout 120 python scripts/microbench_step.py x_wan_b16 steady 50)"
  done
done | tee $O/ab_stream.log
# ---- SQ counters of the new build's steady launches
export TMPDIR=/tmp; cd /tmp
CTRS="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"
for spec in c5_wan:fp32:philox:temporal c5_wan:bf16:philox:temporal c5_wan:fp32:torch:temporal c5_wan:fp32:philox:box; do
  IFS=: read wl dt rng mk <<< "$spec"
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_sq_$$ -o t -- python $R/scripts/microbench_step.py $wl steady 20 $rng $mk > $O/sq_${wl}_${dt}_${rng}_$mk.log 2>&1
  python $R/scripts/rocprof_summary.py /tmp/p_sq_$$/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "lp_step_kernel\|counter" > $O/sq_${wl}_${dt}_${rng}_$mk.md
  rm -rf /tmp/p_sq_$$
  echo "== $wl $dt $rng $mk"; cat $O/sq_${wl}_${dt}_${rng}_$mk.md | cut -c1-200
done 2>&1 | tee $O/sq.log
cd $R
