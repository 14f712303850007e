#!/bin/bash
# Round 5: the C2 steady launch under rocprofv3 in the bench's own graph replays, round-4 tree against this one on ONE box (the filed
# kernel-trace passes of the two rounds come from different boxes: 3.16 us then, 3.34 / 3.45 us now); and the wall time of the default
# bench line in the driver's form.
R=$PWD; O=$R/gpurun_out/r05_ab_c2; mkdir -p $O
export TMPDIR=/tmp
for round in 1 2; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then T=$R/build/r04_tree; else T=$R; fi
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_c2_$$ -o t -- python $T/bench.py --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline --no-large-shape --extras 0 > /dev/null 2>&1
    python $R/scripts/rocprof_summary.py /tmp/p_c2_$$/t_results.db > $O/c2_kernel_trace_${tree}_tree_round$round.md 2>&1
    echo "round $round $tree C2 bench under rocprofv3: $(grep 'lp_step_kernel<1, 2, 28u' $O/c2_kernel_trace_${tree}_tree_round$round.md | head -1 | cut -c1-50,128-215)"
    rm -rf /tmp/p_c2_$$; cd $R
  done
done | tee $O/ab_c2_rocprof.log
cd $R
s=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err; e=$(date +%s.%N)
echo "default bench line, driver form: wall $(python -c "print(round($e - $s, 1))") s, rc=$?" | tee -a $O/ab_c2_rocprof.log
python -c "
import json; l=json.load(open('$O/driver_form.json')); print('value', round(l['value']), 'blocks', [k for k in ('reference_gpu_eager','rccl_single_rank_selftest','sdxl_shaped_backbone','cpu_baseline') if isinstance(l.get(k), dict) and 'error' not in l[k]])" | tee -a $O/ab_c2_rocprof.log
