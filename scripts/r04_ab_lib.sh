#!/bin/bash
# same-box A/B of two builds of THIS tree: build/liblanpaint_hip_prev.so (the commit before, built by hand from a stash) against
# the product library; the streaming launches, then the whole GPU suite on the product library
R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2 3; do
  for spec in "c5_wan steady" "c5_wan first" "c5_wan last" "c5_wan steady 200 torch" "x_wan_b4 steady 100"; do
    for lib in prev new; do
      if [ $lib = new ]; then unset LANPAINT_AMD_LIB; else export LANPAINT_AMD_LIB=$R/build/liblanpaint_hip_$lib.so; fi
      a=$(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-64)
      echo "round $round $lib [$spec] $a"
    done
  done
done | tee $O/ab_lib.log
for lib in prev new; do
  if [ $lib = new ]; then unset LANPAINT_AMD_LIB; else export LANPAINT_AMD_LIB=$R/build/liblanpaint_hip_$lib.so; fi
  echo "$lib bf16: $(LANPAINT_AMD_BENCH_DTYPE=bf16 timeout 120 python scripts/microbench_step.py c5_wan steady 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-64)"
  echo "$lib es: $(timeout 100 python scripts/microbench_es.py c5_wan 2>&1 | grep -v amdgpu.ids | sed -n 2p)"
done | tee -a $O/ab_lib.log
unset LANPAINT_AMD_LIB
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest6.log 2>&1; grep -n "passed\|failed" $O/pytest6.log | tail -2
