R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2 3; do
  for spec in "c5_wan steady" "x_wan_b4 steady 100" "c3_sdxl_b4 steady" "c2_sdxl steady"; do
    for lib in nopre new; do
      if [ $lib = new ]; then unset LANPAINT_AMD_LIB; else export LANPAINT_AMD_LIB=$R/build/liblanpaint_hip_$lib.so; fi
      a=$(timeout 120 python scripts/microbench_step.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-64)
      echo "round $round $lib [$spec] $a"
    done
  done
done | tee $O/ab_preload.log
