#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s10; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_es -o t -- python $R/scripts/_es_debug.py > $OUT/es_debug_under_rocprof.log 2>&1
python $R/scripts/timeline_gaps.py /tmp/p_es/t_results.db 60 > $OUT/es_timeline_gaps.md 2>&1
cat $OUT/es_timeline_gaps.md | cut -c1-150
