#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s9; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep "passed\|failed\|rc=" $OUT/pytest_gpu.log
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_es -o t -- python $R/scripts/_es_debug.py > $OUT/es_debug_under_rocprof.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_es/t_results.db > $OUT/es_kernel_trace.md 2>&1
head -14 $OUT/es_kernel_trace.md | cut -c1-230
