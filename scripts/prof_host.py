import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from lanpaint_amd import LanPaint
shape, flow, n_sig, n_think = bench.WORKLOADS["c2_sdxl"]
dev = torch.device("cuda", 0)
sig_np = bench.karras_sigmas(n_sig)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
times_list = [bench.times_from_sigma(s, flow) for s in sig_list]
ratios = bench.euler_ratios(sig_list, 4)
eng = LanPaint(bench.StubBackbone(flow), 5, 15.0, 5.0, 1.0, 0.2, rng="philox")
for _ in range(3): bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): bench.schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
