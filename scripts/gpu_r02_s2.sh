#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s2; mkdir -p $OUT
run() { tag=$1; shift; echo "=== $tag" ; env "$@" python scripts/host_cost_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^torch\|x.data_ptr" ; env "$@" python bench.py --no-cpu-baseline --extras 0 --no-large-shape 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('bench c2 it/s', round(l['value']), 'graph_burst_us', l['roofline']['graph_burst_us_per_launch'], 'mean_launch_us', l['roofline']['mean_launch_us'])"; }
run default X=1
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run pkt_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pkt_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_batch DEBUG_HIP_GRAPH_BATCH_SIZE=64
run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run active_wait ROC_ACTIVE_WAIT_TIMEOUT=100
