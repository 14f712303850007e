#!/bin/bash
# same-box A/B, round-3 tree vs current: the ATen-strided kernels (rng="torch" past the grid cap) at the streaming shapes
R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2; do
  for spec in "c5_wan steady 200 torch" "c5_wan last 200 torch" "c5_wan first 200 torch" "x_wan_b4 steady 100 torch" "x_wan_b16 steady 50 torch"; do
    for tree in r03 r04; do
      if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
      line=$(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-66)
      echo "round $round $tree [$spec] $line"
    done
  done
done | tee $O/ab_st.log
