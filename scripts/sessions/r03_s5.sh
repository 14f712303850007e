#!/bin/bash
# round 3, GPU session 5: early-stop history reads (two instead of three at streaming sizes), non-temporal half-width streams,
# the suite under forced graph / forced eager, MFMA-busy evidence
O=gpurun_out/r03_s5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
for wl in c2_sdxl c3_sdxl_b4 c5_wan; do timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids >> $O/microbench_es.log; done
cat $O/microbench_es.log
for wl in c5_wan x_wan_b16; do
  timeout 120 python scripts/microbench_step.py $wl steady 50 2>&1 | grep -v amdgpu.ids >> $O/microbench_bf16.log
  LANPAINT_AMD_BENCH_DTYPE=bf16 timeout 120 python scripts/microbench_step.py $wl steady 50 2>&1 | grep -v amdgpu.ids >> $O/microbench_bf16.log
done
cat $O/microbench_bf16.log
LANPAINT_AMD_GRAPH=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_graph1.log 2>&1; echo "graph=1 rc=$?"; grep -E "passed|failed" $O/pytest_gpu_graph1.log | tail -2
LANPAINT_AMD_GRAPH=0 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_graph0.log 2>&1; echo "graph=0 rc=$?"; grep -E "passed|failed" $O/pytest_gpu_graph0.log | tail -2
timeout 600 bash scripts/gpu_profile_mfma.sh > $O/mfma.log 2>&1; echo "mfma rc=$?"
timeout 300 python bench.py --workload c5_wan --no-large-shape --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s5/bench_c5.json'))
print('c5', round(d['value']), {k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
PY
