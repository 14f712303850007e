#!/bin/bash
# Round 5: the reference's noise stream at streaming sizes, round-4 tree (ATen-strided lanes: four 4-byte streams per tensor) against
# this one (ATen-ordered generation, LDS transpose, 16 bytes per lane), same box: rocprofv3 durations, graph-burst cost, SQ counters.
R=$PWD; O=$R/gpurun_out/r05_ab_torch; mkdir -p $O
export TMPDIR=/tmp
inr() { if [ $1 = r04 ]; then cd $R/build/r04_tree; else cd $R; fi; }
for round in 1 2; do
 for tree in r04 r05; do
  for spec in "c5_wan steady 50 torch" "c5_wan first 50 torch" "x_wan_b4 steady 30 torch" "x_wan_b16 steady 20 torch" "c5_wan steady 50 torch box"; do
    inr $tree; T=$PWD
    b=$(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-64)
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt_$$ -o t -- python $T/scripts/microbench_step.py $spec > /dev/null 2>&1
    echo "round $round $tree [$spec]: burst $b | rocprofv3 $(python $R/scripts/rocprof_summary.py /tmp/p_kt_$$/t_results.db 2>&1 | grep -i 'lp_step_kernel' | head -1 | cut -c1-48,130-220)"
    rm -rf /tmp/p_kt_$$; cd $R
  done
 done
done | tee $O/ab_torch.log
cd /tmp
CTRS="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS"
timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_f_$$ -o t -- python $R/scripts/microbench_step.py c5_wan steady 20 torch > /dev/null 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_f_$$/t_results.db --pmc 2>&1 | grep -i "lp_step_kernel" | cut -c1-60,130-220 | tee $O/sq_torch.md
rm -rf /tmp/p_f_$$; cd $R
for round in 1 2; do
  for tree in r04 r05; do
      inr $tree
      timeout 200 python bench.py --workload c5_wan --rng torch --steps 40 --warmup 5 --repeats 1 --extras 0 --no-large-shape --no-cpu-baseline > $O/line_${tree}_c5_torch.json 2>/dev/null
      python -c "
import json; l=json.load(open('$O/line_${tree}_c5_torch.json')); print('round $round $tree c5_wan rng=torch value', round(l['value']), 'repeat', [round(v) for v in (l.get('repeats') or {}).get('values', [])], 'parity', l['parity_check']['ok'])"
      cd $R
  done
done | tee $O/ab_lines.log
