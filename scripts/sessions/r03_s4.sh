#!/bin/bash
# round 3, GPU session 4: final library -- tests (product + coverage build), early-stop micro-benchmark, the rocprofv3 profile set
O=gpurun_out/r03_s4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
rm -f $O/trace.txt
LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_trace.log 2>&1
echo "trace pytest rc=$? lines=$(sort -u $O/trace.txt | wc -l)"
for wl in c2_sdxl c3_sdxl_b4 c5_wan; do timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids >> $O/microbench_es.log; done
cat $O/microbench_es.log
timeout 1500 bash scripts/gpu_profile.sh > $O/gpu_profile.log 2>&1; echo "profile rc=$?"
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-large-shape > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "2rank rc=$?"
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s4/bench_c2.json'))
print(round(d['value']), d['ms_per_step'], {k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
PY
ls gpurun_out/profiles | wc -l
