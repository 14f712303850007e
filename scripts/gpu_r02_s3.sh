#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s3; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
python bench.py --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python -c "import json; l=json.load(open('$OUT/bench_c2.json')); print('bench c2 it/s', round(l['value']), 'ms/step', l['ms_per_step'])"
python scripts/host_vs_gpu.py c2_sdxl 1 2>&1 | grep -v amdgpu | head -4
python scripts/host_vs_gpu.py c2_sdxl 1 2>&1 | grep "host cost"
