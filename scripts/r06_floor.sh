#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/r06_floor; rm -rf $O; mkdir -p $O
python scripts/launch_floor.py > $O/host_sigma_call.md 2> $O/host_sigma_call.err; tail -3 $O/host_sigma_call.err; head -70 $O/host_sigma_call.md
( time python bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $O/driver_form_extras.json ) > $O/driver_form.out 2> $O/driver_form.err
tail -1 $O/driver_form.out > $O/driver_form.json; wc -c $O/driver_form.json; tail -4 $O/driver_form.err
