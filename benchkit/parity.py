"""Does the engine about to be timed compute what the reference computes?  One schedule pass in lockstep with the CPU oracle
(oracle/lanpaint_oracle.py -- test infrastructure, used here as the CHECKER only, never as the thing measured)."""
from __future__ import annotations

import numpy as np
import torch

from .workloads import HYPER, StubBackbone, StubSampling

PARITY_TOL = 1e-5                # BASELINE.json north_star: output MSE vs the reference < 1e-5


def lerp_np(start, end, w):
    """torch.lerp(start, end, w) in numpy fp32 (ATen's two-sided formula)."""
    w = np.float32(w)
    d = (end - start).astype(np.float32)
    return (start + w * d).astype(np.float32) if w < 0.5 else (end - d * (np.float32(1) - w)).astype(np.float32)


def bf16_round(a):
    """numpy fp32 -> nearest-even bf16, returned as fp32 (what the kernels' v_cvt_pk_bf16_f32 and torch's .to(bfloat16) do)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return u.view(np.float32)


class Bf16StubOracle:
    """What StubBackbone computes when the engine hands it bf16 latents (model_dtype=torch.bfloat16), restated in numpy for the
    oracle side of parity_check: the input rounded to bf16 (the kernel emits x_in as bf16; the final call's x is cast), the
    two products formed in fp32 and rounded to bf16 -- with the scales as the stub holds them (a bf16 tensor on the one-launch
    path of latency-bound latents, Python scalars above)."""

    def __init__(self, flow, n_el):
        self.inner_model = self
        self.model_sampling = StubSampling(flow)
        small = n_el <= 512 * 1024
        self.s0, self.s1 = ((bf16_round(np.float32([0.9]))[0], bf16_round(np.float32([0.8]))[0]) if small
                            else (np.float32(0.9), np.float32(0.8)))

    def __call__(self, x, t, model_options=None, seed=None):
        xb = bf16_round(x)
        return bf16_round(xb * self.s0), bf16_round(xb * self.s1)


def parity_check(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=None, oracle_model=None):
    """ONE schedule pass of the engine that is about to be timed -- same object, same launch mode (graph replay / eager),
    same noise generator, same mask format -- in lockstep with the CPU oracle fed the very draws the engine's kernels generate:
      rng="philox": lp_philox_normal(seed, sequence number, slot) for the sequence numbers this call's launches use (the
                    device-side counter of a replayed loop / the host-side one of eager launches, read around the call:
                    LanPaint.rng_position); iteration i draws slot 0 of launch i for its POST half-step and slot 1 of launch
                    i - 1 for its PRE half-step (reference lanpaint.py:277,280,283);
      rng="torch":  what torch.randn returns from the generator state the call starts in (drawn first, state restored).
    Returns {"mse_x", "mse_denoised_max", ...}; MSE in float64 over all elements, final latent and every denoised."""
    from lanpaint_amd import _cabi
    from oracle.lanpaint_oracle import OracleLanPaint
    lib, dev = _cabi.load(), x0.device
    shape, n_el = tuple(x0.shape), x0.numel()
    ns = len(sig_list) if max_sigmas is None else max(1, min(len(sig_list), int(max_sigmas)))
    per_call = max(0, 2 * n_think - 1)
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream     # noqa: E731

    def philox(seq, slot):
        out = torch.empty(n_el, dtype=torch.float32, device=dev)
        _cabi.check(lib.lp_philox_normal(out.data_ptr(), n_el, seed, seq, slot, stream()), "lp_philox_normal")
        return out.cpu().numpy().reshape(shape)

    seed = int(engine.philox_seed if engine.philox_seed is not None else 0) & 0xFFFFFFFFFFFFFFFF
    draws = []
    if oracle_model is not None:          # a caller-supplied restatement of the backbone (e.g. SDXLShapedBackbone.as_oracle_model())
        model = oracle_model
    else:
        model = StubBackbone(flow) if engine.model_dtype is None else Bf16StubOracle(flow, n_el)
        assert engine.model_dtype in (None, torch.bfloat16), "parity_check restates the stub for fp32 and bf16 backbones"
    oracle = OracleLanPaint(model, n_think, HYPER["Friction"], float(engine.chara_lamb), float(engine.chara_beta),
                            float(engine.step_size), is_flow=flow, min_step_frac=float(engine.min_step_frac),
                            randn=lambda like: draws.pop(0))
    to_np = lambda t: t.detach().cpu().numpy()                      # noqa: E731
    y_n, noise_n, mask_n = to_np(y), to_np(noise), to_np(mask)
    xg, xo = x0.clone(), to_np(x0).copy()
    worst, modes, drawn = 0.0, [], 0
    for i in range(ns):
        if engine.rng == "philox":
            c0, p0 = engine.rng_position(dev)
        else:                      # the reference's own stream: draw what the call will draw, put the generator back
            state = torch.cuda.get_rng_state(dev)
            draws[:] = [to_np(torch.randn(shape, device=dev)) for _ in range(per_call)]
            after = engine.rng_position(dev)[0]
            torch.cuda.set_rng_state(state, dev)
        den_g = engine(xg, y, noise, sig_list[i], mask, times_list[i], None, 0, n_steps=n_think)
        if engine.rng == "philox":
            c1, p1 = engine.rng_position(dev)
            if c1 != c0:           # a replayed loop: launch k drew with sequence number c0 + k
                base, used, mode = c0, c1 - c0, "graph"
            else:                  # eager launches: 2^48 + the host-side launch count
                base, used, mode = (1 << 48) + p0, p1 - p0, "eager"
            assert used == n_think, f"sigma call {i}: {used} noise-drawing launches, expected {n_think} ({mode})"
            modes.append(mode)
            draws[:] = [philox(base + k // 2, k % 2) for k in range(per_call)]
        else:
            assert engine.rng_position(dev)[0] == after, "the engine did not leave torch's generator where the reference would"
            modes.append("torch")
        drawn += len(draws)
        den_o = oracle(xo, y_n, noise_n, to_np(sig_list[i]), mask_n, tuple(to_np(t) for t in times_list[i]), None, 0,
                       n_steps=n_think)
        assert not draws, "oracle and engine disagree on the number of draws of a sigma call"
        worst = max(worst, float(np.mean((to_np(den_g).astype(np.float64) - den_o) ** 2)))
        if i + 1 < len(sig_list):
            w = float(ratios[i].reshape(-1)[0])
            xg = torch.lerp(den_g, xg, ratios[i])
            xo = lerp_np(den_o, xo, w)
    mse_x = float(np.mean((to_np(xg).astype(np.float64) - xo) ** 2))
    ok = bool(np.isfinite(mse_x) and np.isfinite(worst) and mse_x < PARITY_TOL and worst < PARITY_TOL)
    return {"mse_x": mse_x, "mse_denoised_max": worst, "tolerance": PARITY_TOL, "ok": ok, "sigmas_checked": ns,
            "sigmas_in_schedule": len(sig_list), "think_iterations_checked": ns * n_think, "draws": drawn,
            "launch_modes": {m: modes.count(m) for m in sorted(set(modes))},
            "captured_calls": len(getattr(engine, "_graphs", ())),
            "checker": "oracle/lanpaint_oracle.py (numpy fp32 restatement of the reference, pinned to reference-generated "
                       "fixtures) on the draws the engine's own kernels generated, sigma call by sigma call, Euler update "
                       "between sigmas; the engine object, launch mode, generator and mask format are the timed ones"}


def check_job(job, engine, **kw):
    return parity_check(engine, job.x0, job.y, job.noise, job.mask, job.sig_list, job.times_list, job.ratios, job.n_think,
                        job.flow, **kw)
