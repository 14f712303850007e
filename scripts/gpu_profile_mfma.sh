#!/bin/bash
# MFMA-busy evidence for the stand-in backbone next to the lp:: kernels (kept separate from gpu_profile.sh: MIOpen under
# --pmc FETCH_SIZE crashed rocprofv3 once).  Counters only with --kernel-trace, no other trace domain.
set -u
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o t -- python $R/scripts/unet_pass.py 4 > $OUT/unet_pmc_mfma.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_mfma/t_results.db --pmc 2>&1 | grep -A400 "counter | dispatches" > $OUT/unet_pmc_mfma.md
python $R/scripts/rocprof_summary.py /tmp/p_mfma/t_results.db 2>&1 | head -25 > $OUT/unet_kernel_trace.md
rm -rf /tmp/p_mfma
grep -c MFMA $OUT/unet_pmc_mfma.md; grep "lp::" $OUT/unet_pmc_mfma.md | grep MFMA | cut -c1-160 | head
