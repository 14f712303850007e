#!/bin/bash
O=gpurun_out/r03_s10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
rm -f $O/trace.txt
LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_trace.log 2>&1
echo "trace pytest rc=$? lines=$(sort -u $O/trace.txt | wc -l)"
for i in 1 2 3 4; do for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; LANPAINT_AMD_SPECULATE=$1 LANPAINT_AMD_FOLD_SIGMA=$2 timeout 200 python bench.py --no-large-shape --no-cpu-baseline --steps 160 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(\"speculate=$1 fold_sigma=$2\", round(d[\"value\"]), \"node\", round(d[\"node_default_schedule\"][\"value\"]))"; done; done
