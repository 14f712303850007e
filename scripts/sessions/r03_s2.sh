#!/bin/bash
# round 3, GPU session 2: tests (product + coverage build), bench with the new roofline fields, runtime knob sweep on the
# per-graph-launch cost, early-stop micro-benchmark baseline
O=gpurun_out/r03_s2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
rm -f $O/trace.txt
LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=$PWD/$O/trace.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_trace.log 2>&1
echo "trace pytest rc=$? lines=$(sort -u $O/trace.txt | wc -l)"
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
for knob in "NONE=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_CLR_MAX_BATCH_SIZE=4096" "DEBUG_CLR_MAX_BATCH_SIZE=1" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=4096" \
    "DEBUG_HIP_BLOCK_SYNC=0" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" \
    "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_USE_FGS_KERNARG=0" "ROC_SKIP_KERNEL_ARG_COPY=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "GPU_FORCE_QUEUE_PROFILING=1" \
    "ROC_AQL_QUEUE_SIZE=65536" "ROC_SIGNAL_POOL_SIZE=4096" "HIP_LAUNCH_BLOCKING=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "GPU_STREAMOPS_CP_WAIT=1"; do
  echo "=== $knob" >> $O/knobs.log
  env "$knob" timeout 60 build/graph_setparams 12 2>&1 | grep -E "^B (graph only|SetParams|eager chain \+ SetParams)|HIP error" >> $O/knobs.log
done
for wl in c2_sdxl c3_sdxl_b4 c5_wan; do timeout 120 python scripts/microbench_es.py $wl >> $O/microbench_es.log 2>&1; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s2/bench_c2.json'))
print(round(d['value']), d['ms_per_step'], {k:round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
print('engine_defaults', d.get('engine_defaults',{}).get('detail'))
for k in ('roofline','roofline_hbm_bound_shape','roofline_hbm_past_l3'):
    r=d[k]; print(k, {kk:(round(r[kk],4) if isinstance(r[kk],float) else r[kk]) for kk in ('frac_algorithmic','frac_counter','mean_launch_us','graph_burst_us_per_launch','event_mean_us','rocprofv3_mean_launch_us','event_over_rocprofv3','graph_burst_over_rocprofv3') if kk in r})
ra=d['roofline_hbm_past_l3']['region_aware_streams']; print('region_aware', {kk:ra[kk] for kk in ('frac_algorithmic','frac_counter','event_mean_us','graph_burst_us_per_launch','rocprofv3_mean_launch_us','event_over_rocprofv3','graph_burst_over_rocprofv3','event_mean_us_per_round','graph_burst_us_per_round')})
print('bf16', {k:{kk:v[kk] for kk in ('mean_launch_us','frac_algorithmic','achieved')} for k,v in d.get('bf16_heads',{}).items() if isinstance(v,dict) and 'mean_launch_us' in v})
PY
cat $O/knobs.log; cat $O/microbench_es.log
