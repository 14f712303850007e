#!/bin/bash
# Round 5: second same-box A/B of the streaming launches (round-4 tree against this one): rocprofv3 kernel durations of the steady
# launch, the graph-burst cost three times over, and the C5 / C3 bench lines (engine-driven: replayed graphs, device-side RNG state)
R=$PWD; O=$R/gpurun_out/r05_ab2; mkdir -p $O
export TMPDIR=/tmp
inr() { if [ $1 = r04 ]; then cd $R/build/r04_tree; else cd $R; fi; }
for round in 1 2 3; do
  for spec in "c5_wan steady" "c5_wan steady 200 torch"; do
    for tree in r04 r05; do
      inr $tree; echo "round $round $tree [$spec] $(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-72)"; cd $R
    done
  done
  for tree in r04 r05; do
    inr $tree; echo "round $round $tree [c5_wan steady bf16] $(LANPAINT_AMD_BENCH_DTYPE=bf16 timeout 120 python scripts/microbench_step.py c5_wan steady 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-72)"; cd $R
  done
done | tee $O/ab_burst.log
for tree in r04 r05; do
  for spec in "fp32:philox" "bf16:philox" "fp32:torch"; do
    dt=${spec%%:*}; rng=${spec#*:}
    inr $tree; T=$PWD; cd /tmp
    LANPAINT_AMD_BENCH_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt_$$ -o t -- python $T/scripts/microbench_step.py c5_wan steady 50 $rng > /dev/null 2>&1
    echo "$tree c5_wan steady $dt $rng: $(python $R/scripts/rocprof_summary.py /tmp/p_kt_$$/t_results.db 2>&1 | grep -i 'lp_step_kernel' | head -1 | cut -c1-60,130-220)"
    rm -rf /tmp/p_kt_$$; cd $R
  done
done | tee $O/ab_rocprof.log
for round in 1 2; do
  for tree in r04 r05; do
    for wl in c5_wan c3_sdxl_b4; do
      inr $tree
      timeout 200 python bench.py --workload $wl --steps 40 --warmup 5 --repeats 1 --extras 0 --no-large-shape --no-cpu-baseline > $O/line_${tree}_$wl.json 2>/dev/null
      python -c "
import json; l=json.load(open('$O/line_${tree}_$wl.json')); print('round $round $tree $wl value', round(l['value']), 'repeat', [round(v) for v in (l.get('repeats') or {}).get('values', [])], 'steady_launch_us', round(l['roofline']['mean_launch_us'],3))"
      cd $R
    done
  done
done | tee $O/ab_lines.log
