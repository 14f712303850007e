#!/bin/bash
# soak of the capture / replay plumbing after the round-6 split and the one-launch node path: 4 x 1200 random event sequences
# with fresh randomness, the random-configuration fuzz, the default soak script
set -u
O=gpurun_out/r06_soak; rm -rf $O; mkdir -p $O
( time LP_FUZZ_EXAMPLES=1200 LP_FUZZ_RANDOM=1 timeout 2400 python -m pytest tests/test_gpu_state_machine.py -q -x -p no:cacheprovider -k "random_event" ) > $O/state_machine_soak.log 2>&1; tail -6 $O/state_machine_soak.log
( time timeout 900 python scripts/soak.py 3000 ) > $O/soak.log 2>&1; tail -5 $O/soak.log
