#!/usr/bin/env python3
"""Micro-timing of the Python in front of lp_node_call (C2, node defaults): which statements the 9.6 us are made of."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from benchkit.extras import build_node_sampler                 # noqa: E402
from lanpaint_amd import _cabi                                 # noqa: E402
from lanpaint_amd._util import tensor_version                  # noqa: E402

dev = torch.device("cuda", 0)
args = bench.parse_args([])
k, node_pass, n_sig = build_node_sampler(args, dev)
for _ in range(8):
    node_pass()
torch.cuda.synchronize()
pm = k.PaintMethod
cap0 = pm._last_cap
x = torch.randn((1, 4, 128, 128), device=dev)
sigma = torch.full((1,), 1.5, device=dev)
dm = k._mask_cache[0]()
lm = k._latent_mask(dm)
times = k._times[2][0][:3]
mo = {}
N = 20000


def t(label, fn):
    for _ in range(200):
        fn()
    t0 = time.perf_counter_ns()
    for _ in range(N):
        fn()
    print(f"{label:58s} {(time.perf_counter_ns() - t0) / N / 1e3:6.2f} us")


t("empty lambda (loop + call overhead, subtract)", lambda: None)
t("model_type chain + two compares", lambda: (k.inner_model.inner_model.model_type == "FLUX", k.inner_model.inner_model.model_type in ("FLOW", None)))
t("sigma / schedule guards", lambda: (sigma.is_cuda and sigma.dtype == torch.float32 and sigma.ndim == 1 and k.sigmas.is_cuda and k.sigmas.dtype == torch.float32 and k.sigmas.device == sigma.device))
t("sigma.is_contiguous + shape[0]", lambda: (sigma.is_contiguous(), sigma.shape[0]))
t("torch.cuda.current_device()", torch.cuda.current_device)
t("static tuple build + compare", lambda: (1, k._sched[2], k._sched[3], False, pm.n_steps, k.LanPaint_early_stop, k._sched[4], getattr(k, "LanPaint_min_step_frac", 1.0), k._mailbox[0]) == k._node_static)
nd = k._node_desc
t("three ctypes field writes (nd.sigma, nd.seq, nd.times_out)", lambda: (setattr(nd, "sigma", sigma.data_ptr()), setattr(nd, "seq", 5), setattr(nd, "times_out", times[0].data_ptr())))
k._last_step = 3
t("_guess_inner_steps", lambda: k._guess_inner_steps(nd, 1, False))
t("_latent_mask(denoise_mask)", lambda: k._latent_mask(dm))
pm.latent_image, pm.noise = k.latent_image, k.noise
t("engine._noise_is_zero(noise)", lambda: pm._noise_is_zero(k.noise))
pm._noise_regenerated = False
pm.audio_indicator = pm.audio_correction = None
t("engine._same_call(...)", lambda: pm._same_call(cap0, x, sigma, lm, times, cap0.ident[4], k._node_static and cap0.ident[2], 0))
t("  of which _override_state()", pm._override_state)
t("  of which _hyper_key()", pm._hyper_key)
t("  of which _times_ok", lambda: pm._times_ok(cap0, times))
t("torch.empty_like(x)", lambda: torch.empty_like(x))
k0 = cap0.k0_desc
t("eight k0 field writes + data_ptr calls", lambda: (setattr(k0, "x", x.data_ptr()), setattr(k0, "noise", k.noise.data_ptr()), setattr(k0, "t_ve", times[0].data_ptr()),
                                                      setattr(k0, "t_abt", times[1].data_ptr()), setattr(k0, "t_rsig", sigma.data_ptr()), setattr(k0, "t_model", times[0].data_ptr()),
                                                      k0.io_table_val.__setitem__(0, 1), k0.io_table_val.__setitem__(1, 2)))
t("generator: _generator + get_offset + initial_seed + 2 writes", lambda: (lambda g: (g.get_offset(), g.initial_seed(), k0.rng_state_val.__setitem__(0, 4), k0.rng_state_val.__setitem__(1, 5)))(pm._generator(dev)))
t("engine._stream(device)", lambda: pm._stream(dev))
lib = _cabi.load()
t("a ctypes call into the library (lp_abi_version)", lib.lp_abi_version)
t("ctypes.byref(nd)", lambda: ctypes.byref(nd))
