#!/bin/bash
# round 3, GPU session 7: the one-call node path
O=gpurun_out/r03_s7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-large-shape --no-cpu-baseline > $O/bench_c2_$i.json 2> $O/bench_c2_$i.err
done
timeout 120 python scripts/prof_node.py > $O/prof_node.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_s7/bench*.json')):
    try:
        d=json.load(open(f)); ex={k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), ex)
    except Exception as e: print(f, 'ERR', e)
PY
head -30 $O/prof_node.log
