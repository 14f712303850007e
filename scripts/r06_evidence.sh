#!/bin/bash
# Round 6 evidence session on a GPU box: the driver-form and default bench lines (with --pmc: live traffic), the other BASELINE
# workloads, the extras side-car, eight-rank rehearsals over gloo, the launch-floor / host-cost report, smoke().
set -u
R=$PWD; O=$R/gpurun_out/r06_final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
run() { name=$1; shift; ( time timeout 600 python bench.py "$@" --sidecar $O/${name}_extras.json ) > $O/$name.out 2> $O/$name.err; echo "$name rc=$?"; tail -1 $O/$name.out > $O/$name.json; wc -c < $O/$name.json; grep real $O/$name.err; }
run bench_c2_driver_form --gpus 1 --steps 20 --warmup 5
run bench_c2
run bench_c2_live_pmc --pmc --no-summary --steps 40
run bench_c2_extras --extras 1
run bench_c2_philox --rng philox --graph 1 --mask-format bits --no-summary
for wl in c1_sd15 c3_sdxl_b4 c4_flux c5_wan; do run bench_$wl --workload $wl --steps 40 --warmup 5 --repeats 2 --no-summary --cpu-seconds 6; done
run bench_c5_wan_live_pmc --workload c5_wan --pmc --no-summary --steps 20 --no-cpu-baseline
for wl in c3_sdxl_b4 c5_wan; do run bench_8rank_gloo_$wl --gpus 8 --dist-backend gloo --workload $wl --steps 20 --warmup 3 --repeats 1 --cpu-seconds 4 --parity-sigmas 2; done
python scripts/launch_floor.py > $O/host_sigma_call.md 2> $O/host_sigma_call.err; echo "floor rc=$?"
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<'PY'
import json, glob, os
O = "gpurun_out/r06_final/"
for f in sorted(glob.glob(O + "bench_*.json")):
    if f.endswith("_extras.json"):
        continue
    try:
        l = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "UNPARSEABLE", e); continue
    r, c = l.get("roofline") or {}, l.get("cpu_baseline") or {}
    print(os.path.basename(f), os.path.getsize(f), "B value", l.get("value"), "ms", l.get("ms_per_step"), "parity", (l.get("parity_check") or {}).get("ok"),
          "frac", r.get("frac"), "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:14], "cpu", c.get("value"), c.get("kind"), "summary", l.get("summary"), l.get("error", ""))
PY
