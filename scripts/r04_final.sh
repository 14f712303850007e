#!/bin/bash
# round 4, final GPU session: the whole GPU suite, the default bench line, the other BASELINE workloads, the 8-rank rehearsals
O=gpurun_out/r04_final; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
for wl in c1_sd15 c3_sdxl_b4 c4_flux c5_wan; do
  timeout 200 python bench.py --workload $wl --steps 40 --warmup 5 --repeats 2 --extras 0 --no-large-shape --cpu-seconds 6 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
done
for wl in c3_sdxl_b4 c5_wan; do
  timeout 200 python bench.py --gpus 8 --dist-backend gloo --workload $wl --steps 20 --warmup 3 --repeats 1 --no-large-shape --extras 0 --cpu-seconds 4 > $O/bench_8rank_gloo_$wl.json 2> $O/bench_8rank_gloo_$wl.err; echo "8rank $wl rc=$?"
done
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04_final/bench_c2.json"))
print(l["value"], l["ms_per_step"], l["parity_check"]["mse_x"], l["cpu_baseline"]["value"], l["cpu_baseline"]["kind"], l["cpu_baseline"]["threads"].keys())
print({k:l["roofline"][k] for k in ("frac","frac_counter","mean_launch_us")})
for k in ("engine_defaults","node_default_schedule","inner_early_stop_armed","reference_noise_stream","with_backbone"):
    print(k, l.get(k,{}).get("value"))
print({k:(v.get("mean_launch_us"),v.get("frac_counter"),v.get("counter_side_GBps")) for k,v in l["bf16_heads"].items() if isinstance(v,dict)})
h=l["roofline_hbm_bound_shape"]; p=l["roofline_hbm_past_l3"]
print("c5", h["mean_launch_us"], h["frac_counter"], "pastL3 every", p["mean_launch_us"], p["frac_counter"], "RA", p["region_aware_streams"]["event_mean_us"], p["region_aware_streams"]["frac_counter"])
for wl in ("c1_sd15","c3_sdxl_b4","c4_flux","c5_wan"):
    l=json.load(open("gpurun_out/r04_final/bench_%s.json"%wl)); print(wl, l["value"], l["parity_check"]["ok"], l["parity_check"]["sigmas_checked"], l["cpu_baseline"]["value"])
for wl in ("c3_sdxl_b4","c5_wan"):
    l=json.load(open("gpurun_out/r04_final/bench_8rank_gloo_%s.json"%wl)); print("8rank", wl, l["value"], l["collective"], l["parity_check"]["ok"])
PY
