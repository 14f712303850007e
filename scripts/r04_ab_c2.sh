#!/bin/bash
# same-box A/B at the latency-bound shapes: round-3 tree vs current, steady launch in a replayed graph, interleaved
R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2 3; do
  for wl in c2_sdxl c3_sdxl_b4 c1_sd15; do
    for tree in r03 r04; do
      if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
      line=$(timeout 120 python scripts/microbench_step.py $wl steady 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-70)
      echo "round $round $tree $line"
    done
  done
done | tee $O/ab_c2.log
