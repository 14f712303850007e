#!/bin/bash
# round 4, GPU session 1: the whole GPU suite (incl. the new published-configuration parity tests and the 8-rank rehearsals)
# and the default bench line
O=gpurun_out/r04_s1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04_s1/bench_c2.json"))
print(l["value"], l["parity_check"], l["cpu_baseline"], l["collective"])
print({k:l["roofline"][k] for k in ("frac","frac_counter","mean_launch_us")})
PY
