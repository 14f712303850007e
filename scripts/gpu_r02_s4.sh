#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s4; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl"
python bench.py --no-cpu-baseline --no-large-shape --steps 100 --repeats 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python -c "
import json; l=json.load(open('$OUT/bench_c2.json'))
print('bench c2 it/s', round(l['value']), 'repeats', [round(v) for v in l['repeats']['values']])
print('node_default_schedule', l.get('node_default_schedule'))
print('reference_noise_stream', l.get('reference_noise_stream'))
"
