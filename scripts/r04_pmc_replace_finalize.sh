#!/bin/bash
# HBM-side traffic of the replace and finalize launches at the video latent (region-aware streams, as shipped), FETCH_SIZE and
# WRITE_SIZE in separate passes, kernel-trace only next to them.  Every-stream bytes for comparison: 20.1 B / element each.
R=$PWD; O=$R/gpurun_out/r04_pmc_rf; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_rf_$ctr -o t -- python $R/bench.py --workload c5_wan --steps 2 --warmup 1 --repeats 0 --graph 0 --no-cpu-baseline --no-large-shape --extras 0 --no-parity-check > $O/run_$ctr.log 2>&1
  python $R/scripts/rocprof_summary.py /tmp/p_rf_$ctr/t_results.db --pmc 2>&1 | grep -A200 "counter | dispatches" | grep -i "finalize\|49u\|28u" | cut -c1-60,140-220 > $O/c5_$ctr.md
  echo "== $ctr"; cat $O/c5_$ctr.md
done
rm -rf /tmp/p_rf_*
