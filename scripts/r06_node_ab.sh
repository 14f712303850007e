#!/bin/bash
# node path: the speculated call as ONE graph launch (sigma-folded root, lp_graph_clone_sigma_root) against the eager replace launch
# in front of the tail graph (LANPAINT_AMD_NODE_ONE_LAUNCH=0), same box
set -u
R=$PWD; O=$R/gpurun_out/r06_node_ab; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_api.py tests/test_gpu_state_machine.py tests/test_gpu_ksampler_glue.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "node or specul or ksampler or sampler or sigma or random_event" > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in 1 0 1 0; do
  LANPAINT_AMD_NODE_ONE_LAUNCH=$v python scripts/node_path_probe.py 2>&1 | grep -v amdgpu > $O/probe_$v.log
  echo "ONE_LAUNCH=$v: $(tail -1 $O/probe_$v.log)"
done
cat $O/probe_1.log
python scripts/launch_floor.py 2>/dev/null | sed -n '/node-default schedule/,/cProfile: host side of the replayed sigma call, engine-direct/p' | head -20
