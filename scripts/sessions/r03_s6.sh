#!/bin/bash
# round 3, GPU session 6: full-schedule reference fixtures on the GPU, early-stop launch width experiment, fallback graph layout
O=gpurun_out/r03_s6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
for wl in c2_sdxl c3_sdxl_b4; do
  echo "--- default width" >> $O/microbench_es_vec.log
  timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids >> $O/microbench_es_vec.log
  echo "--- LANPAINT_AMD_TUNE_VEC=4" >> $O/microbench_es_vec.log
  LANPAINT_AMD_TUNE_VEC=4 timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids >> $O/microbench_es_vec.log
done
cat $O/microbench_es_vec.log
LANPAINT_AMD_REPLACE_IN_GRAPH=0 LANPAINT_AMD_GRAPH=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_round2layout.log 2>&1; echo "round-2 layout rc=$?"; grep -E "passed|failed" $O/pytest_gpu_round2layout.log | tail -2
LANPAINT_AMD_GRAPH=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_graph1.log 2>&1; echo "graph=1 rc=$?"; grep -E "passed|failed" $O/pytest_gpu_graph1.log | tail -2
