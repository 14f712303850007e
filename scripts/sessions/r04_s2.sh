#!/bin/bash
# round 4, GPU session 2: the lane-pair half-width streams -- bitwise test, then bf16 vs fp32 heads at the two streaming shapes
O=gpurun_out/r04_s2; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "half or bf16 or fp16 or dtype or matrix or region" 2>&1 | tail -5
for wl in c5_wan x_wan_b16; do for dt in fp32 bf16; do
  LANPAINT_AMD_BENCH_DTYPE=$dt timeout 120 python scripts/microbench_step.py $wl steady 2>&1 | grep -v amdgpu.ids | tail -1
done; done | tee $O/microbench_bf16_pair.log
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_s2/microbench_bf16_pair.log
import torch, bench, json
from lanpaint_amd import _cabi
dev=torch.device("cuda",0)
for wl in ("c5_wan","x_wan_b16"):
    for dt in (None, torch.bfloat16):
        m=bench.measure_hbm_bound_shape(_cabi, dev, workload=wl, launches=100, model_dtype=dt)
        print(wl, "bf16" if dt else "fp32", "event mean us", round(m["mean_launch_us"],2), "alg GB/s", round(m["achieved"]), "counter frac", m["frac_counter"])
PY
