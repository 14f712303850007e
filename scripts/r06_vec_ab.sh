#!/bin/bash
# steady step launch at the latency-bound / L2-resident sizes: one element per lane against four (LP_TUNE_VEC1 / VEC4), both generators
for wl in c2_sdxl c3_sdxl_b4; do for rng in philox torch; do for vec in 1 4; do
  LANPAINT_AMD_TUNE_VEC=$vec python scripts/microbench_step.py $wl steady 200 $rng 2>/dev/null | cut -c1-150
done; done; done
