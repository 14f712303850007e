import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanpaint_amd import LanPaint, nodes
from tests import golden_cases as gc
from tests.test_gpu_api import _DummyModel, _DummySampling
DEV="cuda"
shape, n_think = (1, 4, 16, 16), 5
sig = gc.karras_sigmas(10, 0.05, 12.0)
rs = np.random.default_rng(4)
y = rs.standard_normal(shape, dtype=np.float32); noise = rs.standard_normal(shape, dtype=np.float32)
denoise_mask = (rs.random(shape) > 0.4).astype(np.float32)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
class M(_DummyModel):
    def __call__(self, x, sigma, model_options=None, seed=None):
        self.calls += 1
        return 0.9 * x, 0.8 * x
from tests.stubs import VESampling
model = M(VESampling()); model.model_type = "EPS"
k = nodes.KSamplerX0Inpaint(model, tt(sig)); k.latent_image, k.noise = tt(y), tt(noise)
k.PaintMethod = LanPaint(model, n_think, 15.0, 5.0, 1.0, 0.2, MinStepFrac=1.0, rng="torch", graph=True)
k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
nc = k.PaintMethod.node_call
def counted(*a, **kw):
    r = nc(*a, **kw)
    nd = a[-1]
    print("node_call ->", None if r is None else "ok", "guess", nd.guess, "spec", nd.speculated, "hit", nd.hit, "n_eff", nd.n_eff, "launched", nd.launched, "last_step", k._last_step, "valid_word", bool(nd.valid_word), "counts", nd.n_counts)
    return r
k.PaintMethod.node_call = counted
dm, mo = tt(denoise_mask), {}
x = tt(y + noise * sig[0])
for j in [0, 1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9, 0, 1, 2]:
    s = torch.full((1,), float(sig[j]), dtype=torch.float32, device=DEV)
    den = k(x, s, dm, model_options=mo, seed=0)
    pm = k.PaintMethod
    cap = pm._last_cap
    print("j", j, "last_cap", cap is not None, "graphs", len(pm._graphs), [ (c.fast, c.final_in_graph, c.binding is not None, c.tail is not None) for c in pm._graphs.values()], "blocked", pm._graph_blocked, "es_type", type(k.LanPaint_early_stop), type(pm.n_steps), s.device.index, torch.cuda.current_device())
    x = torch.lerp(den, x, 0.7)
torch.cuda.synchronize()
print("table", k._n_eff_table)
