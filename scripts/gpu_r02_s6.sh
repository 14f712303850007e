#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s6; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl"
{
for wl in c5_wan x_wan_b4 x_wan_b16; do
  for mk in temporal box blob; do
    python scripts/microbench_step.py $wl steady 100 philox $mk
    LANPAINT_AMD_NO_REGION_SKIP=1 python scripts/microbench_step.py $wl steady 100 philox $mk
  done
done
for wl in c2_sdxl c3_sdxl_b4 c5_wan x_wan_b4; do
  python scripts/microbench_step.py $wl steady 100 torch
  python scripts/microbench_step.py $wl steady 100 philox
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/microbench.log
