#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s5; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl"
