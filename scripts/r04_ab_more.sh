#!/bin/bash
# same-box A/B, round-3 tree vs current: the other phases, the torch noise stream, the early-stop launch
R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2; do
  for spec in "c2_sdxl first" "c2_sdxl last" "c2_sdxl replace" "c2_sdxl steady 200 torch" "c5_wan first" "c5_wan steady 200 torch" "c4_flux steady"; do
    for tree in r03 r04; do
      if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
      line=$(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-62)
      echo "round $round $tree [$spec] $line"
    done
  done
  for wl in c2_sdxl c3_sdxl_b4; do
    for tree in r03 r04; do
      if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
      line=$(timeout 120 python scripts/microbench_es.py $wl 2>&1 | grep -v amdgpu.ids | sed -n 2p)
      echo "round $round $tree $line"
    done
  done
done | tee $O/ab_more.log
