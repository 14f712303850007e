#!/bin/bash
# round 3, GPU session 3: the accumulator-set early stop, coverage of every instantiation, runtime dispatch mode A/B
O=gpurun_out/r03_s3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
rm -f $O/trace.txt
LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_trace.log 2>&1
echo "trace pytest rc=$? lines=$(sort -u $O/trace.txt | wc -l)"
for wl in c2_sdxl c3_sdxl_b4 c5_wan; do timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids >> $O/microbench_es.log; done
cat $O/microbench_es.log
B="--no-large-shape --no-cpu-baseline --steps 200"
timeout 300 python bench.py $B > $O/bench_default.json 2> $O/bench_default.err
AMD_DIRECT_DISPATCH=0 timeout 300 python bench.py $B > $O/bench_dd0.json 2> $O/bench_dd0.err
AMD_DIRECT_DISPATCH=0 ROC_SYSTEM_SCOPE_SIGNAL=0 timeout 300 python bench.py $B > $O/bench_dd0_sss0.json 2> $O/bench_dd0_sss0.err
ROC_SYSTEM_SCOPE_SIGNAL=0 timeout 300 python bench.py $B --extras 0 > $O/bench_sss0.json 2> $O/bench_sss0.err
timeout 300 python bench.py $B --graph 0 --extras 0 > $O/bench_eager_default.json 2> $O/bench_eager_default.err
AMD_DIRECT_DISPATCH=0 timeout 300 python bench.py $B --graph 0 --extras 0 > $O/bench_eager_dd0.json 2> $O/bench_eager_dd0.err
timeout 300 python bench.py $B --workload c5_wan --extras 1 > $O/bench_c5.json 2> $O/bench_c5.err
AMD_DIRECT_DISPATCH=0 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_dd0.log 2>&1; echo "pytest dd0 rc=$?" >> $O/pytest_gpu_dd0.log
grep -E "passed|failed|rc=" $O/pytest_gpu_dd0.log | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_s3/bench*.json')):
    try:
        d=json.load(open(f)); ex={k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), ex)
    except Exception as e: print(f, 'ERR', e)
PY
