#!/bin/bash
# Un-profiled bench.py lines of the current build for every BASELINE config (+ eager launches, the reference's
# noise stream, a 2-rank rehearsal on the one GPU of the box); results under gpurun_out/bench/.
set -u
R=$PWD; OUT=$R/gpurun_out/bench; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --workload c1_sd15 --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c1.json 2>/dev/null
python bench.py --workload c3_sdxl_b4 --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c3.json 2>/dev/null
python bench.py --workload c4_flux --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c4.json 2>/dev/null
python bench.py --workload c5_wan --steps 100 --no-cpu-baseline --extras 0 > $OUT/bench_c5.json 2>/dev/null
python bench.py --graph 0 --steps 100 --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c2_eager.json 2>/dev/null
python bench.py --rng torch --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c2_torchrng.json 2>/dev/null
python bench.py --rng torch --workload c5_wan --steps 100 --no-cpu-baseline --extras 0 --no-large-shape > $OUT/bench_c5_torchrng.json 2>/dev/null
python scripts/host_vs_gpu.py c2_sdxl 1 > $OUT/host_vs_gpu_c2.log 2>&1
python scripts/host_cost_probe.py > $OUT/host_cost_probe.log 2>&1
for be in gloo nccl; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --dist-backend $be --steps 100 --warmup 10 --repeats 0 > $OUT/bench_2rank_$be.log 2>&1
  echo "rc=$?" >> $OUT/bench_2rank_$be.log
done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
r = l.get("roofline") or {}
print(sys.argv[1].split("/")[-1], round(l["value"]), "it/s", "repeats", [round(v) for v in (l.get("repeats") or {}).get("values", [])],
      "frac", round(r.get("frac", 0), 4), "mean_us", round(r.get("mean_launch_us", 0), 2))
for k in ("node_default_schedule", "reference_noise_stream", "inner_early_stop_armed", "with_backbone"):
    if k in l: print("   ", k, l[k].get("value") and round(l[k]["value"]), l[k].get("error", ""))
for k in ("roofline_hbm_bound_shape", "roofline_hbm_past_l3"):
    v = l.get(k)
    if v: print("   ", k, {q: (round(v[q], 3) if isinstance(v.get(q), float) else v.get(q)) for q in ("achieved", "frac", "mean_launch_us", "rocprofv3_mean_launch_us", "traffic", "hbm_side_GBps", "error")}, v.get("region_aware_streams"))
if l.get("cpu_baseline"): print("   cpu", {k: v for k, v in l["cpu_baseline"].items() if k in ("value", "cores", "kind", "port_over_reference", "reference_estimate")})
PY
done
grep -h "^{" $OUT/bench_2rank_gloo.log | cut -c1-200; tail -3 $OUT/bench_2rank_nccl.log | cut -c1-200; grep -h "Duplicate GPU" $OUT/bench_2rank_nccl.log | head -2
head -3 $OUT/host_vs_gpu_c2.log | grep -v amdgpu
