#!/bin/bash
# same-box A/B of the default bench line: the round-3 tree (build/r03_tree, exported from commit 9c1b296) against this one
R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
for round in 1 2; do
  for tree in r03 r04; do
    if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
    timeout 250 python bench.py --no-cpu-baseline --no-large-shape --repeats 1 > $O/headline_$tree.json 2>/dev/null
    python - <<PY
import json
l = json.load(open("$O/headline_$tree.json"))
g = lambda k: round(l[k]["value"]) if isinstance(l.get(k), dict) and l[k].get("value") else None
print("round $round $tree value", round(l["value"]), "engine_defaults", g("engine_defaults"), "node_default_schedule", g("node_default_schedule"),
      "inner_early_stop_armed", g("inner_early_stop_armed"), "reference_noise_stream", g("reference_noise_stream"))
PY
  done
done | tee $O/ab_headline.log
for tree in r03 r04; do
  if [ $tree = r03 ]; then cd $R/build/r03_tree; else cd $R; fi
  for wl in c3_sdxl_b4 c5_wan; do
    timeout 200 python bench.py --workload $wl --steps 40 --warmup 5 --repeats 1 --extras 0 --no-large-shape --no-cpu-baseline > $O/headline_${tree}_$wl.json 2>/dev/null
    python -c "
import json; l=json.load(open('$O/headline_${tree}_$wl.json')); print('$tree $wl value', round(l['value']))"
  done
done | tee -a $O/ab_headline.log
