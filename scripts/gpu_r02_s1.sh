#!/bin/bash
# round-2 session 1: sanity, C2 timeline (where the GPU idles inside a sigma call), host vs GPU, 2-rank rehearsal
set -u
R=$PWD
OUT=$R/gpurun_out/r02_s1
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python bench.py --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python scripts/host_vs_gpu.py c2_sdxl 1 > $OUT/host_vs_gpu.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-shape --extras 0 > $OUT/c2_bench_under_rocprof.log 2>&1
python $R/scripts/timeline_gaps.py /tmp/p_c2/t_results.db > $OUT/c2_timeline_gaps.md 2>&1
rm -rf /tmp/p_c2
cd $R
for be in gloo nccl; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --dist-backend $be --steps 20 --warmup 5 > $OUT/bench_2rank_$be.log 2>&1
  echo "rc=$?" >> $OUT/bench_2rank_$be.log
done
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench_c2.json | cut -c1-300; cat $OUT/host_vs_gpu.log | head -5; head -40 $OUT/c2_timeline_gaps.md; tail -5 $OUT/bench_2rank_gloo.log | cut -c1-400; tail -8 $OUT/bench_2rank_nccl.log | cut -c1-400
