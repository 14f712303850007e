#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r02_s8; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl"
python bench.py --no-cpu-baseline --steps 100 --repeats 1 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r02_s8/bench_c2.json"))
print("c2", round(l["value"]))
for k in ("node_default_schedule", "reference_noise_stream", "inner_early_stop_armed"):
    print("   ", k, l[k].get("value") and round(l[k]["value"]), l[k].get("error", ""))
v = l["roofline_hbm_past_l3"]
print({q: v.get(q) for q in ("achieved", "frac", "mean_launch_us", "rocprofv3_mean_launch_us", "hbm_side_GBps", "error")})
print(v.get("region_aware_streams"))
PY
