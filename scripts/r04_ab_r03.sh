#!/bin/bash
# same-box A/B: the round-3 tree (build/r03_tree, exported from commit 9c1b296 and built in the build container) against the
# current one, the steady launch in a replayed graph (scripts/microbench_step.py of each tree), interleaved, three rounds
O=$PWD/gpurun_out/r04_ab; mkdir -p $O
R=$PWD
for round in 1 2 3; do
  for spec in c5_wan:fp32 c5_wan:bf16 x_wan_b16:fp32 x_wan_b16:bf16 c2_sdxl:fp32; do
    wl=${spec%%:*}; dt=${spec#*:}
    for tree in r03 r04; do
      if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
      line=$(LANPAINT_AMD_BENCH_DTYPE=$dt timeout 120 python scripts/microbench_step.py $wl steady 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-90)
      echo "round $round $tree $line"
    done
  done
done | tee $O/ab_r03_r04.log
