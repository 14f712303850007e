#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s7; mkdir -p $OUT
{
for wl in c5_wan x_wan_b4 x_wan_b16; do
  for mk in temporal blob; do
    python scripts/microbench_step.py $wl steady 100 philox $mk
    LANPAINT_AMD_NO_REGION_SKIP=1 python scripts/microbench_step.py $wl steady 100 philox $mk
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/microbench_region_skip.log
