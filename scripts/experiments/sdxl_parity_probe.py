"""Debug probe: where does the engine-vs-oracle difference behind the SDXL-shaped stand-in come from?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from lanpaint_amd import LanPaint
from tests.sdxl_standin import SDXLShapedBackbone
dev = torch.device("cuda", 0)
shape, flow, n_sig, n_think = bench.WORKLOADS["c2_sdxl"]
sig_np = bench.karras_sigmas(n_sig)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
mask = bench.attach_mask_format(mask, "bits")
sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
ratios = bench.euler_ratios(sig_list, 4)
for dtype, fused, md in ((torch.float32, False, None), (torch.float32, True, None), (torch.bfloat16, True, torch.bfloat16), (torch.bfloat16, False, torch.bfloat16)):
    net = SDXLShapedBackbone(dev, dtype=dtype, fused=fused)
    # determinism of the module itself
    a = net.predict(x0, sig_list[0]); b = net.predict(x0, sig_list[0])
    det = float((a[0].float() - b[0].float()).abs().max())
    # sensitivity: one bf16 ulp on the input
    xe = x0.clone(); xe.view(-1)[::97] *= (1 + 2 ** -8)
    c = net.predict(xe, sig_list[0])
    sens = float((a[0].float() - c[0].float()).abs().max()), float(a[0].float().abs().max()), float((a[0].float()-a[1].float()).abs().max())
    for ms in (1, 2, 4):
        eng = LanPaint(net, n_think, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=3, graph=False, model_dtype=md)
        r = bench.parity_check(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=ms, oracle_model=net.as_oracle_model())
        print(dtype, "fused" if fused else "eager-cfg", "sigmas", ms, "mse_x %.3e mse_den %.3e" % (r["mse_x"], r["mse_denoised_max"]), "det", det, "sens", sens, flush=True)
