#!/bin/bash
# Round 5, VERDICT r04 next #4 (last part) + weak #6: what the run-time-phase (PH = 0) fallback kernels cost at the video-latent size,
# next to the hot kernel: graph-burst us per launch, rocprofv3 mean per dispatch and SQ VALU instructions per wave.
# Then the torch-stream A/B against the round-4 tree (rocprofv3 durations, same box).
R=$PWD; O=$R/gpurun_out/r05_fallbacks; mkdir -p $O
export TMPDIR=/tmp
CTRS="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
for spec in "hot:" "u8:LANPAINT_AMD_BENCH_MASK_FORMAT=u8" "f32mask:LANPAINT_AMD_BENCH_MASK_FORMAT=f32" \
            "soft:LANPAINT_AMD_BENCH_MASK_FORMAT=f32 LANPAINT_AMD_BENCH_SOFT=1" "hostxi:LANPAINT_AMD_BENCH_HOSTXI=1" \
            "av:LANPAINT_AMD_BENCH_MASK_FORMAT=f32 LANPAINT_AMD_BENCH_AV=1"; do
  name=${spec%%:*}; envs=${spec#*:}
  b=$(env $envs timeout 120 python scripts/microbench_step.py c5_wan steady 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)
  cd /tmp
  env $envs timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_f_$$ -o t -- python $R/scripts/microbench_step.py c5_wan steady 50 > /dev/null 2>&1
  k=$(python $R/scripts/rocprof_summary.py /tmp/p_f_$$/t_results.db 2>&1 | grep -i 'lp_step_kernel' | head -1 | cut -c1-50,130-215)
  rm -rf /tmp/p_f_$$
  env $envs timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_f_$$ -o t -- python $R/scripts/microbench_step.py c5_wan steady 20 > /dev/null 2>&1
  q=$(python $R/scripts/rocprof_summary.py /tmp/p_f_$$/t_results.db --pmc 2>&1 | grep -i "lp_step_kernel" | grep "SQ_INSTS_VALU\|SQ_WAVES" | awk -F"|" '{print $3, $5}' | tr '\n' ' ')
  rm -rf /tmp/p_f_$$
  cd $R
  echo "== $name"; echo "   burst: $b"; echo "   rocprofv3: $k"; echo "   sq: $q"
done 2>&1 | tee $O/fallbacks.log
inr() { if [ $1 = r04 ]; then cd $R/build/r04_tree; else cd $R; fi; }
for round in 1 2; do
 for tree in r04 r05; do
  for spec in "c5_wan steady 50 torch" "x_wan_b4 steady 30 torch"; do
    inr $tree; T=$PWD; cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt_$$ -o t -- python $T/scripts/microbench_step.py $spec > /dev/null 2>&1
    echo "round $round $tree [$spec]: $(python $R/scripts/rocprof_summary.py /tmp/p_kt_$$/t_results.db 2>&1 | grep -i 'lp_step_kernel' | head -1 | cut -c1-60,130-220)"
    rm -rf /tmp/p_kt_$$; cd $R
  done
 done
done | tee $O/ab_torch.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/p_f_$$ -o t -- python $R/scripts/microbench_step.py c5_wan steady 20 torch > /dev/null 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_f_$$/t_results.db --pmc 2>&1 | grep -i "lp_step_kernel" | cut -c1-60,130-220 | tee $O/sq_torch.md
rm -rf /tmp/p_f_$$; cd $R
