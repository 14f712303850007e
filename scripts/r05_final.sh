#!/bin/bash
# round 5, final GPU session: the whole GPU suite against the coverage build (instantiation trace), the default bench line in the
# driver's form and in full, the other BASELINE workloads, the reference's noise stream at C2 / C5, the 8-rank rehearsals, smoke()
O=gpurun_out/r05_final; mkdir -p $O
LANPAINT_AMD_LIB=$PWD/build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 700 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > $O/pytest_gpu_trace_build.log 2>&1; echo "pytest(trace build) rc=$?"; tail -4 $O/pytest_gpu_trace_build.log
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_form.json 2> $O/bench_c2_driver_form.err; echo "driver-form rc=$?"
for wl in c1_sd15 c3_sdxl_b4 c4_flux c5_wan; do
  timeout 250 python bench.py --workload $wl --steps 40 --warmup 5 --repeats 2 --extras 0 --no-large-shape --cpu-seconds 6 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
done
timeout 250 python bench.py --workload c5_wan --rng torch --steps 40 --warmup 5 --repeats 2 --extras 0 --no-large-shape --no-cpu-baseline > $O/bench_c5_wan_torch.json 2>/dev/null; echo "c5 torch rc=$?"
for wl in c3_sdxl_b4 c5_wan; do
  timeout 250 python bench.py --gpus 8 --dist-backend gloo --workload $wl --steps 20 --warmup 3 --repeats 1 --no-large-shape --extras 0 --cpu-seconds 4 --parity-sigmas 2 > $O/bench_8rank_gloo_$wl.json 2> $O/bench_8rank_gloo_$wl.err; echo "8rank $wl rc=$?"
done
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<'PY'
import json
O="gpurun_out/r05_final/"
l=json.load(open(O+"bench_c2.json"))
print(l["value"], l["ms_per_step"], l["parity_check"]["mse_x"], l["cpu_baseline"]["value"], l["cpu_baseline"]["kind"])
print({k:l["roofline"].get(k) for k in ("frac","frac_every_stream","frac_counter","frac_rocprofv3","mean_launch_us")})
print("reference_gpu_eager", l["reference_gpu_eager"].get("value"), l["reference_gpu_eager"].get("product_over_reference_same_gpu"), "rccl", (l.get("rccl_single_rank_selftest") or {}).get("ok"))
for k in ("engine_defaults","node_default_schedule","inner_early_stop_armed","reference_noise_stream","bf16_backbone","with_backbone","sdxl_shaped_backbone"):
    print(k, l.get(k,{}).get("value"), l.get(k,{}).get("error",""))
print({k:(v.get("mean_launch_us"),v.get("frac"),v.get("frac_counter")) for k,v in l["bf16_heads"].items() if isinstance(v,dict)})
h=l["roofline_hbm_bound_shape"]; p=l["roofline_hbm_past_l3"]
print("c5", h["mean_launch_us"], h["frac"], h["frac_counter"], "pastL3 every", p["mean_launch_us"], p["frac"], "RA", p["region_aware_streams"]["event_mean_us"], p["region_aware_streams"]["frac"])
d=json.load(open(O+"bench_c2_driver_form.json")); print("driver form", d["value"], d["ms_per_step"])
for wl in ("c1_sd15","c3_sdxl_b4","c4_flux","c5_wan"):
    l=json.load(open(O+"bench_%s.json"%wl)); print(wl, l["value"], l["parity_check"]["ok"], l["parity_check"]["sigmas_checked"], l["cpu_baseline"]["value"], l["roofline"]["frac"])
l=json.load(open(O+"bench_c5_wan_torch.json")); print("c5 torch", l["value"], l["parity_check"]["ok"])
for wl in ("c3_sdxl_b4","c5_wan"):
    l=json.load(open(O+"bench_8rank_gloo_%s.json"%wl)); print("8rank", wl, l["value"], l["collective"], l["parity_check"]["ok"], l["dist"]["parity_ok_all_ranks"], l["dist"]["own_it_s_spread"])
PY
