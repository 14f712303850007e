#!/bin/bash
# The GPU suite on a fresh box, every failure listed (no -x); tail to gpurun_out/r06_suite/.
set -u
R=$PWD; O=$R/gpurun_out/r06_suite; rm -rf $O; mkdir -p $O
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -25 $O/gpu_tests.log
