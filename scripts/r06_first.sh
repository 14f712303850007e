#!/bin/bash
# Round 6, first GPU contact of the rewritten bench: the GPU suite, the default bench line (size + time), the driver form.
set -u
R=$PWD; O=$R/gpurun_out/r06_first; rm -rf $O; mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $O/driver_form_extras.json ) > $O/driver_form.out 2> $O/driver_form.err
tail -1 $O/driver_form.out > $O/driver_form.json; wc -c $O/driver_form.json; tail -4 $O/driver_form.err
( time python bench.py --sidecar $O/default_extras.json ) > $O/default.out 2> $O/default.err
tail -1 $O/default.out > $O/default.json; wc -c $O/default.json; tail -4 $O/default.err
for m in philox dropin; do python scripts/host_vs_gpu.py c2_sdxl 1 $m > $O/host_vs_gpu_$m.log 2>&1; grep "host enqueue\|idle GPU" $O/host_vs_gpu_$m.log; done
