#!/bin/bash
# round 3, GPU session 1: SetParams semantics, GPU tests, bench A/B of the one-launch layout, self-spawned 2-rank line
O=gpurun_out/r03_s1; mkdir -p $O
timeout 120 build/graph_setparams 12 > $O/setparams.log 2>&1; echo "setparams rc=$?" >> $O/setparams.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py --steps 200 > $O/bench_c2.json 2> $O/bench_c2.err
LANPAINT_AMD_REPLACE_IN_GRAPH=0 timeout 300 python bench.py --steps 200 --no-large-shape --no-cpu-baseline > $O/bench_c2_round2layout.json 2> $O/bench_c2_round2layout.err
timeout 300 python bench.py --steps 200 --no-large-shape --no-cpu-baseline --extras 0 > $O/bench_c2_b.json 2> $O/bench_c2_b.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-large-shape > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_s1/bench*.json')):
    try:
        d=json.load(open(f)); ex={k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v}
        print(f.split('/')[-1], round(d['value']), d['ms_per_step'], ex, (d.get('dist') or {}).get('backend'))
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/setparams.log
