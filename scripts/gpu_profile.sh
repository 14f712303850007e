#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace + PMC passes of bench.py and of the
# c5_wan micro-benchmark; leaves only small text summaries under gpurun_out/ (the rocpd .db
# files are deleted: gpurun copies back at most 64 MiB).
set -u
R=$PWD
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
summ() { python $R/scripts/rocprof_summary.py "$@"; }

rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-shape > $OUT/c2_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c2/t_results.db > $OUT/c2_kernel_trace.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o t -- python $R/scripts/microbench_step.py c5_wan steady 50 > $OUT/c5_microbench_under_rocprof.log 2>&1
summ /tmp/p_c5/t_results.db > $OUT/c5_kernel_trace.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_c5b -o t -- python $R/bench.py --workload c5_wan --steps 4 --warmup 2 --no-cpu-baseline > $OUT/c5_bench_under_rocprof.json.log 2>&1
summ /tmp/p_c5b/t_results.db > $OUT/c5_bench_kernel_trace.md 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_c2_$ctr -o t -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-large-shape > $OUT/c2_pmc_$ctr.log 2>&1
  summ /tmp/p_pmc_c2_$ctr/t_results.db --pmc > $OUT/c2_pmc_$ctr.md 2>&1
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/p_pmc_c5_$ctr -o t -- python $R/scripts/microbench_step.py c5_wan steady 20 > $OUT/c5_pmc_$ctr.log 2>&1
  summ /tmp/p_pmc_c5_$ctr/t_results.db --pmc > $OUT/c5_pmc_$ctr.md 2>&1
done
python - <<'PY' > $OUT/pmc_schema.txt 2>&1
import sqlite3, glob
p = glob.glob('/tmp/p_pmc_c5_FETCH_SIZE/*.db')[0]
cur = sqlite3.connect(p).cursor()
for v in ['pmc_events', 'pmc_info', 'counters_collection', 'rocpd_pmc_event', 'rocpd_info_pmc']:
    try:
        print(v, [d[1] for d in cur.execute(f"pragma table_info({v})")])
        for r in list(cur.execute(f"select * from {v} limit 3")):
            print("   ", r)
    except Exception as e:
        print(v, "ERR", e)
PY
rm -rf /tmp/p_*
cd $R
ls -la $OUT
