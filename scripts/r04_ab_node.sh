R=$PWD; O=$R/gpurun_out/r04_ab; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest5.log 2>&1; grep -n "passed\|failed" $O/pytest5.log | tail -2
for round in 1 2; do
  for lib in prev new; do
    if [ $lib = new ]; then unset LANPAINT_AMD_LIB; else export LANPAINT_AMD_LIB=$R/build/liblanpaint_hip_$lib.so; fi
    timeout 200 python bench.py --no-cpu-baseline --no-large-shape --repeats 1 > $O/node_$lib.json 2>/dev/null
    python -c "
import json; l=json.load(open('$O/node_$lib.json')); print('round $round $lib value', round(l['value']), 'node_default_schedule', round(l['node_default_schedule']['value']), 'engine_defaults', round(l['engine_defaults']['value']))"
  done
done | tee $O/ab_node.log
