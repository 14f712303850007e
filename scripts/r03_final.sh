#!/bin/bash
# round 3, final GPU session: the suite under every mode switch, coverage trace, the rocprofv3 profile set, bench lines
O=gpurun_out/r03_final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "default rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
for env in "LANPAINT_AMD_GRAPH=1" "LANPAINT_AMD_GRAPH=0" "LANPAINT_AMD_REPLACE_IN_GRAPH=0" "LANPAINT_AMD_SPECULATE=0" "LANPAINT_AMD_FOLD_SIGMA=0"; do
  env $env timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_${env%%=*}_${env##*=}.log 2>&1; echo "$env rc=$?"; grep -E "passed|failed" $O/pytest_gpu_${env%%=*}_${env##*=}.log | tail -1
done
rm -f $O/trace.txt
LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_trace.log 2>&1
echo "trace rc=$? lines=$(sort -u $O/trace.txt | wc -l)"
timeout 1500 bash scripts/gpu_profile.sh > $O/gpu_profile.log 2>&1; echo "profile rc=$?"
timeout 600 bash scripts/gpu_profile_mfma.sh > $O/mfma.log 2>&1; echo "mfma rc=$?"
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-large-shape > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "2rank rc=$?"
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
timeout 400 python bench.py --workload c5_wan --steps 100 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python - <<'PY'
import json
for f in ('bench_c2','bench_c5'):
    d=json.load(open(f'gpurun_out/r03_final/{f}.json'))
    print(f, round(d['value']), d['ms_per_step'], {k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
PY
