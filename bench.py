#!/usr/bin/env python3
"""bench.py -- Langevin think-iterations/sec of the HIP path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over the whole sigma schedule of the workload
(C2: SDXL 1x4x128x128 latent per GPU, 30 Karras sigmas x 5 think iterations = 150
think iterations + 30 final denoise calls), engine driven directly,
stub backbone x -> (0.9x, 0.8x), synthetic inputs resident in HBM before the timed
region.  value = think iterations of ALL ranks / max-over-ranks wall time.
One process per GPU; ranks are independent replicas (mask / known latent are
broadcast from rank 0 over RCCL at setup; no collective inside the loop) -> weak scaling.

The same JSON line carries
  roofline     : the dominant kernel (steady-state fused lp_step) -- algorithmic bytes
                 per launch / mean launch duration measured with HIP events on the
                 launch stream in an instrumented replay of the timed region
  cpu_baseline : the CPU oracle (op-for-op port of the reference engine) on torch-CPU
                 tensors, timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

# RCCL / IPC between the ranks of one node needs dmabuf IPC on this driver stack; must be in the environment before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (per-GPU latent shape, flow?, n_sigmas, think iterations per sigma)
    "c1_sd15":  ((1, 4, 64, 64), False, 20, 5),
    "c2_sdxl":  ((1, 4, 128, 128), False, 30, 5),
    "c3_sdxl_b4": ((4, 4, 128, 128), False, 30, 5),
    "c4_flux":  ((1, 16, 64, 64), True, 28, 10),
    "c5_wan":   ((1, 16, 21, 60, 104), True, 30, 5),
    # not BASELINE configs: larger batches of the video latent, to see the kernel once the fixed launch cost is amortised
    "x_wan_b4": ((4, 16, 21, 60, 104), True, 30, 5),
    "x_wan_b16": ((16, 16, 21, 60, 104), True, 30, 5),
}
HYPER = dict(NSteps=5, Friction=15.0, Lambda=5.0, Beta=1.0, StepSize=0.2, MinStepFrac=0.0)
BYTES_PER_EL_STEADY = 36          # read x_t,x0,x0_BIG,y,m,C ; write x_t,C,x_in (fp32, in-kernel RNG)
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable copy)


def karras_sigmas(n, sigma_min=0.0292, sigma_max=14.6146, rho=7.0):
    ramp = np.linspace(0, 1, n, dtype=np.float64)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return ((hi + ramp * (lo - hi)) ** rho).astype(np.float32)


def flow_sigmas(n, shift=3.0):
    t = np.linspace(1.0, 0.0, n + 1, dtype=np.float64)[:-1]
    t = np.clip(shift * t / (1 + (shift - 1) * t), 0.0, 0.999)
    return t.astype(np.float32)


def times_from_sigma(s, flow):
    if flow:
        abt = (1 - s) ** 2 / ((1 - s) ** 2 + s ** 2)
        return s / (1 - s), abt, s
    abt = 1 / (1 + s ** 2)
    return s, abt, (1 - abt) ** 0.5 / ((1 - abt) ** 0.5 + abt ** 0.5)


class StubSampling:
    def __init__(self, flow):
        self.lanpaint_noise_scaling_kind = "flow" if flow else "ve"
        self.noise_scale = 1.0
        self.flow = flow

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        if self.flow:
            return sigma * noise + (1.0 - sigma) * latent_image
        return latent_image + noise * sigma


class StubBackbone:
    """x -> (0.9 x, 0.8 x): isolates the Langevin path (SURVEY.md 8d backbone stand-in (i)).  On latency-bound
    latents (<= 512 Ki elements) both heads come out of ONE broadcast launch (as the two heads of a batched CFG
    forward would) so the stand-in costs a single kernel per call; above that two vectorised launches are cheaper
    than torch's strided broadcast kernel (C5: 35 k vs 31 k it/s), so it stays two."""

    def __init__(self, flow):
        self.inner_model = self
        self.model_sampling = StubSampling(flow)
        self._scales = None

    def __call__(self, x, t, model_options=None, seed=None):
        if not (torch.is_tensor(x) and x.is_cuda) or x.numel() > 512 * 1024:
            return 0.9 * x, 0.8 * x
        s = self._scales
        if s is None or s.device != x.device or s.ndim != x.ndim + 1:
            s = self._scales = torch.tensor([0.9, 0.8], dtype=x.dtype, device=x.device).view(2, *([1] * x.ndim))
        heads = x.unsqueeze(0) * s
        return heads[0], heads[1]


def shared_conditioning(workload, device, seed=0):
    """Synthetic stand-ins, shape and dtype right, for the conditioning tensors every replica of a job shares and rank 0
    therefore broadcasts with the mask and the known latent at set-up: SDXL text states [1, 77, 2048] + pooled
    / ADM vector [1, 2816]; SD1.5 [1, 77, 768]; Flux T5 states [1, 512, 4096] + CLIP pooled [1, 768]; Wan UMT5 states
    [1, 512, 4096]; bf16.  The stub backbone does not read them -- they are there so that the one packed broadcast has the size
    and the layout of a real job's, and every rank reports a checksum of what it received."""
    shapes = {"c1_sd15": {"cond": (1, 77, 768)},
              "c2_sdxl": {"cond": (1, 77, 2048), "pooled": (1, 2816)}, "c3_sdxl_b4": {"cond": (1, 77, 2048), "pooled": (1, 2816)},
              "c4_flux": {"cond": (1, 512, 4096), "pooled": (1, 768)}}.get(workload, {"cond": (1, 512, 4096)})
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    return {k: torch.randn(v, generator=g).to(torch.bfloat16).to(device) for k, v in shapes.items()}


def temporal_known_frames(latent_frames):
    """an 81-frame video whose second half (pixel frames >= P // 2) is inpainted, brought to the
    latent grid as reshape_mask's video path does (nodes.py:100-122): nearest-exact frame index (ATen's fp32 formula)
    and a 5-tap temporal union of the inpaint region.  Returns the number of leading latent frames that stay known."""
    f = int(latent_frames)
    p = 4 * (f - 1) + 1
    scale = np.float32(p) / np.float32(f)
    src = np.minimum(np.floor((np.arange(f, dtype=np.float32) + np.float32(0.5)) * scale).astype(np.int64), p - 1)
    inpaint = src >= p // 2
    union = np.array([inpaint[max(0, t - 2): t + 3].any() for t in range(f)])
    return int(np.argmax(union)) if union.any() else f


def make_mask(shape, kind=None):
    """latent_mask (1 = known).  box: 50 % box over the last axis; temporal: the leading latent frames
    known (video latents, C5); blob: a centred disc of inpainting covering ~38 % of every plane."""
    kind = kind or ("temporal" if len(shape) == 5 else "box")
    mask = np.zeros(shape, dtype=np.float32)
    if kind == "box":
        mask[..., : shape[-1] // 2] = 1.0
    elif kind == "temporal":
        mask[:, :, : temporal_known_frames(shape[2])] = 1.0
    elif kind == "blob":
        h, w = shape[-2], shape[-1]
        yy, xx = np.mgrid[0:h, 0:w]
        mask[...] = (((yy - h / 2) ** 2 + (xx - w / 2) ** 2) > (0.35 * min(h, w)) ** 2).astype(np.float32)
    else:
        raise ValueError(kind)
    return mask


MASK_KIND = None              # set from --mask; None = the workload's default (box for image latents, temporal for video)


def make_inputs(shape, flow, sigma0, seed, device, xp):
    g = np.random.default_rng(seed)
    y = g.standard_normal(shape, dtype=np.float32)
    noise = g.standard_normal(shape, dtype=np.float32)
    x = (sigma0 * noise + (1 - sigma0) * y) if flow else (y + noise * sigma0)
    mask = make_mask(shape, MASK_KIND)
    return tuple(xp(a.astype(np.float32)) for a in (x, y, noise, mask))


def euler_ratios(sig_list, ndim):
    """1 + (sigma_{i+1} - sigma_i) / sigma_i = sigma_{i+1} / sigma_i as the lerp weight of the Euler update,
    broadcastable over the latent."""
    return [(1 + (sig_list[i + 1] - sig_list[i]) / sig_list[i]).reshape((-1,) + (1,) * (ndim - 1))
            for i in range(len(sig_list) - 1)]


def schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think):
    """One step of the bench: the whole sigma schedule, Euler update between sigmas
    (k-diffusion sample_euler form), x mutated in place by the engine each sigma."""
    x = x0.clone()
    ns = len(sig_list)
    for i in range(ns):
        den = engine(x, y, noise, sig_list[i], mask, times_list[i], None, 0, n_steps=n_think)
        if i + 1 < ns:
            x = torch.lerp(den, x, ratios[i])             # x + (x - den) * r, r = dsigma / sigma, in one launch
    return x


PARITY_TOL = 1e-5                # BASELINE.json north_star: output MSE vs the reference < 1e-5


def _lerp_np(start, end, w):
    """torch.lerp(start, end, w) in numpy fp32 (ATen's two-sided formula)."""
    w = np.float32(w)
    d = (end - start).astype(np.float32)
    return (start + w * d).astype(np.float32) if w < 0.5 else (end - d * (np.float32(1) - w)).astype(np.float32)


def _bf16_round(a):
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
        self.s0, self.s1 = ((_bf16_round(np.float32([0.9]))[0], _bf16_round(np.float32([0.8]))[0]) if small
                            else (np.float32(0.9), np.float32(0.8)))

    def __call__(self, x, t, model_options=None, seed=None):
        xb = _bf16_round(x)
        return _bf16_round(xb * self.s0), _bf16_round(xb * self.s1)


def parity_check(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=None, oracle_model=None):
    """ONE schedule pass of the engine that is about to be timed -- same object, same launch mode (graph replay / eager),
    same noise generator, same mask format -- in lockstep with the CPU oracle (oracle/lanpaint_oracle.py, the checker) fed
    the very draws the engine's kernels generate:
      rng="philox": lp_philox_normal(seed, sequence number, slot) for the sequence numbers this call's launches use (the
                    device-side counter of a replayed loop / the host-side one of eager launches, read around the call:
                    LanPaint.rng_position); iteration i draws slot 0 of launch i for its POST half-step and slot 1 of launch
                    i - 1 for its PRE half-step (lanpaint.py:277,280,283);
      rng="torch":  what torch.randn returns from the generator state the call starts in (drawn first, state restored).
    Returns {"mse_x", "mse_denoised_max", ...}; MSE in float64 over all elements, final latent and every denoised."""
    import ctypes
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
            xo = _lerp_np(den_o, xo, w)
    mse_x = float(np.mean((to_np(xg).astype(np.float64) - xo) ** 2))
    ok = bool(np.isfinite(mse_x) and np.isfinite(worst) and mse_x < PARITY_TOL and worst < PARITY_TOL)
    return {"mse_x": mse_x, "mse_denoised_max": worst, "tolerance": PARITY_TOL, "ok": ok, "sigmas_checked": ns,
            "sigmas_in_schedule": len(sig_list), "think_iterations_checked": ns * n_think, "draws": drawn,
            "launch_modes": {m: modes.count(m) for m in sorted(set(modes))},
            "checker": "oracle/lanpaint_oracle.py (numpy fp32 restatement of the reference, pinned to reference-generated "
                       "fixtures) on the draws the engine's own kernels generated, sigma call by sigma call, Euler update "
                       "between sigmas; the engine object, launch mode, generator and mask format are the timed ones"}


def run_gpu(args):
    import torch.distributed as dist
    from lanpaint_amd import LanPaint, _cabi
    from lanpaint_amd import distributed as lpd

    rank, world, local_rank = lpd.env_world()
    if args.gpus != world:
        # (main() starts the ranks itself when no launcher did; getting here means a launcher disagrees with --gpus)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = max(1, torch.cuda.device_count())
    dev_index = local_rank % n_dev                      # identity on an N-GPU node; lets a 1-GPU box rehearse N > 1
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.dist_backend
    if world > 1 and backend == "nccl" and n_dev < world:
        # RCCL refuses two ranks on one device ("Duplicate GPU detected"): a box with fewer GPUs than ranks can only
        # REHEARSE the N > 1 path over gloo, and only when asked to -- a run that was meant to measure RCCL scaling must not
        # quietly turn into something else (the line's top-level `collective` says which library carried the run).
        if not args.allow_gloo_fallback:
            if rank == 0:
                print(f"bench.py: --gpus {world} over RCCL needs {world} devices, this box has {n_dev}.  Pass --dist-backend gloo "
                      "(or --allow-gloo-fallback) to rehearse the multi-rank path on fewer devices; no line is printed.",
                      file=sys.stderr, flush=True)
            raise SystemExit(4)
        backend = "gloo"
    affinity0 = os.sched_getaffinity(0)
    numa = lpd.bind_to_device_numa(dev_index) if not args.no_numa_bind else None
    t_proc0 = time.perf_counter()

    # Per-dispatch event timing of the dominant kernel, taken FIRST in the process: the same burst repeated after
    # graph captures / other streams exist reads ~1.2 us higher at the video-latent size (11.4 vs 10.2 us) although
    # rocprofv3 shows the same 10.5-10.7 us per dispatch in both places -- event bookkeeping, not the kernel.
    # EVERY rank takes its own (round 5; rank 0 alone used to, with the others parked in a collective): the per-rank
    # launch duration is the first thing to look at when one rank of an 8-GPU run is slow (dist.per_rank[].steady_launch_us).
    # All of it -- and the parity pass below -- runs BEFORE the process group exists, so no rank ever waits in a collective
    # for another rank's local work (RCCL's watchdog is the first place an 8-GPU run can die).
    pre_busy = pre_large = pre_past = None
    try:
        if rank == 0 and world == 1 and not args.no_large_shape:
            # 1.2 GB per launch: nothing of it survives in the 256 MiB Infinity Cache between launches.  Every operand
            # streamed (what SURVEY.md 8d's 36 B / element describes) and as shipped (waves whose mask bits are uniform
            # skip the streams their region never reads), interleaved, three clocks each
            pre_past = measure_past_l3(_cabi, dev)
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and args.workload != "c5_wan" and not args.no_large_shape:
            pre_large = measure_hbm_bound_shape(_cabi, dev)      # the bandwidth-bound shape is the sensitive one
        pre_busy = measure_hbm_bound_shape(_cabi, dev, workload=args.workload, launches=120)
    except Exception as e:
        pre_busy = {"error": repr(e)}

    shape, flow, n_sig, n_think = WORKLOADS[args.workload]
    sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), lpd.replica_seed(args.seed, rank), dev, tt)
    t_init0 = time.perf_counter()
    lpd.init(backend, dev)                              # "nccl" IS RCCL on ROCm; no-op at world size 1
    t_init = time.perf_counter() - t_init0
    bcast, cond = {}, None
    if world > 1:      # all replicas inpaint the same image with the same mask under the same conditioning: ONE packed
        # broadcast from rank 0 at set-up -- mask, known latent, cond tensors -- and nothing afterwards
        job = lpd.broadcast_job(dict({"mask": mask, "y": y}, **shared_conditioning(args.workload, dev, args.seed)) if rank == 0 else None,
                                src=0, device=dev, stats=bcast)
        mask, y = job["mask"], job["y"]
        cond = {k: v for k, v in job.items() if k not in ("mask", "y")}
        x0 = (float(sig_np[0]) * noise + (1 - float(sig_np[0])) * y) if flow else (y + noise * float(sig_np[0]))
    if len(shape) == 5 and (args.mask or "temporal") == "temporal" and rank == 0:
        # job set-up as a workflow does it: the pixel-resolution video mask goes through reshape_mask's video path
        # (lp_reshape_mask: nearest-exact + 5-tap temporal union) and must give the latent mask used here
        from lanpaint_amd import nodes as lpn
        frames = 4 * (shape[2] - 1) + 1
        pix = torch.zeros((frames, shape[3] * 8, shape[4] * 8), device=dev)
        pix[frames // 2:] = 1.0                                    # ComfyUI denoise mask: 1 = inpaint
        lat = lpn.reshape_mask(pix, (1,) + tuple(shape[1:]), video_inpainting=True)
        assert torch.equal(1.0 - (lat > 0.5).float(), mask[:1]), "reshape_mask disagrees with the analytic temporal mask"
        del pix, lat
    mask = attach_mask_format(mask, args.mask_format)          # once per job, outside the timed region
    b = shape[0]
    sig_list = [torch.full((b,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
    times_list = [times_from_sigma(s, flow) for s in sig_list]
    ratios = euler_ratios(sig_list, len(shape))

    engine = LanPaint(StubBackbone(flow), HYPER["NSteps"], HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"],
                      HYPER["StepSize"], IS_FLOW=flow, MinStepFrac=HYPER["MinStepFrac"], rng=args.rng,
                      philox_seed=lpd.replica_seed(args.seed, rank), graph=bool(args.graph),
                      model_dtype=torch.bfloat16 if args.model_dtype == "bf16" else None)

    waits = []

    def barrier():
        """dist.barrier + device sync; remembers how long THIS rank waited in the collective (dist.per_rank[].barrier_wait_s:
        a rank that arrives early waits for the slowest one -- the first number to look at when scaling efficiency is off)."""
        if world > 1:
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            dist.barrier()
            waits.append(time.perf_counter() - t_b)
        torch.cuda.synchronize()

    # Before anything is timed: does THIS engine, in THIS configuration, compute what the reference computes?  One schedule
    # pass against the CPU oracle on the same draws (bounded on the video-latent shapes, where a numpy pass over the whole
    # schedule would take minutes).  No `value` is printed when the pass is off by more than the stated tolerance.  EVERY rank
    # checks its own replica (round 5): symmetric work, nobody sits in a collective while rank 0 runs numpy for seconds.
    parity = None
    if not args.no_parity_check:
        n_par = args.parity_sigmas if args.parity_sigmas > 0 else (n_sig if int(np.prod(shape)) <= 512 * 1024 else 2)
        try:
            parity = parity_check(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=n_par)
        except Exception as e:
            parity = {"ok": False, "error": repr(e)}

    # untimed set-up before the W warm-up steps: graph capture and lazy initialisation, then ~0.3 s of the very
    # workload so the clocks have ramped (a 1.3 ms step otherwise gets timed on a chip that is still waking up:
    # back-to-back default runs on one box read 100 k / 100 k / 112 k it/s without it)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
    t_setup = time.perf_counter() - t_proc0
    clocks_before = gpu_clocks(dev_index)
    barrier()
    first_wait = waits[-1] if waits else None
    it0 = engine.iterations_run
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x_last = schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0             # this rank's own K steps, before it waits for the others
    cpu_busy = time.process_time() - cpu0
    barrier()
    elapsed = time.perf_counter() - t0
    iters_local = engine.iterations_run - it0
    assert torch.isfinite(x_last).all(), "bench produced non-finite latents"

    tmax, iters_total = lpd.reduce_throughput(elapsed, iters_local, dev)
    # who took part: every rank reports its device, its own clock and its own iteration count (the evidence that the
    # collective backend really carried `world` ranks and that each of them ran the full K steps)
    dist_info = lpd.gather_rank_reports({
        "rank": rank, "device": f"cuda:{dev_index}", "device_name": torch.cuda.get_device_name(dev),
        "pci_bus_id": _pci_bus_id(dev_index), "numa": numa, "pid": os.getpid(), "steps": args.steps, "iterations": int(iters_local),
        "elapsed_s": elapsed, "it_s": iters_local / elapsed, "final_checksum": float(x_last.double().sum().item()),
        "rows": int(shape[0]),
        # diagnostics of a multi-rank run (round 5): this rank's own time for the K steps (before the closing barrier), the host
        # CPU time it burnt over them (process_time / elapsed ~ 1: the rank is host-bound and wants its own core), how long it
        # waited in the barrier in front of the timed region, its steady-launch duration measured alone before the group
        # existed, the device's clocks, and its own parity verdict
        "own_elapsed_s": own_elapsed, "own_it_s": iters_local / own_elapsed, "process_time_over_elapsed": cpu_busy / own_elapsed,
        "t_first_barrier_wait_s": first_wait, "closing_barrier_wait_s": (waits[-1] if len(waits) > 1 else None),
        "setup_s": t_setup, "init_process_group_s": t_init,
        "steady_launch_us": (pre_busy or {}).get("mean_launch_us"), "clocks_mhz": {"before": clocks_before, "after": gpu_clocks(dev_index)},
        "cpus_allowed": len(os.sched_getaffinity(0)),
        "parity_ok": (None if parity is None else bool(parity.get("ok"))), "parity_mse_x": (None if parity is None else parity.get("mse_x")),
        # what this rank holds of the shared job after the broadcast: equal on every rank, or the broadcast did not deliver
        "shared_checksum": (float(sum(t.double().sum().item() for t in (mask, y, *(cond or {}).values()))) if world > 1 else None)})
    if dist_info is not None and rank == 0:
        dist_info.update({"backend_requested": args.dist_backend, "broadcast_bytes": bcast.get("bytes"),
                          "broadcast_ms": bcast.get("ms"), "launcher": os.environ.get("LANPAINT_BENCH_LAUNCHER", "external"),
                          "collectives_in_timed_region": 0,
                          "shared_tensors": {k: list(v.shape) for k, v in dict({"mask": mask, "y": y}, **(cond or {})).items()},
                          "shared_checksums_equal": len({r.get("shared_checksum") for r in dist_info["per_rank"]}) == 1,
                          "global_rows": sum(r.get("rows", 0) for r in dist_info["per_rank"]),
                          "parity_ok_all_ranks": all(r.get("parity_ok") is not False for r in dist_info["per_rank"]),
                          "slowest_rank": max(dist_info["per_rank"], key=lambda r: r.get("own_elapsed_s") or 0.0).get("rank"),
                          "own_it_s_spread": [min(r.get("own_it_s") or 0.0 for r in dist_info["per_rank"]),
                                              max(r.get("own_it_s") or 0.0 for r in dist_info["per_rank"])],
                          "note": "weak scaling: every rank runs the whole workload on its own replica (seed + rank); one packed "
                                  "broadcast of mask + known latent at set-up, no collective inside the timed loop; value = "
                                  "sum of the ranks' iterations / slowest rank's time"})

    # run-to-run spread of the same measurement: further blocks of K steps, each bracketed like the timed region
    # (the headline `value` stays the first block -- exactly K steps, as the contract says)
    repeat_values = []
    for _ in range(max(0, args.repeats)):
        barrier()
        itr, tr = engine.iterations_run, time.perf_counter()
        for _ in range(args.steps):
            schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        barrier()
        t_r, n_r = lpd.reduce_throughput(time.perf_counter() - tr, engine.iterations_run - itr, dev)
        repeat_values.append(n_r / t_r)

    roofline = None
    if rank == 0:
        try:
            roofline = measure_roofline(engine, _cabi, x0, y, noise, mask, sig_list, times_list, ratios, n_think, args,
                                        pre_busy)
        except Exception as e:
            roofline = {"error": repr(e)}
    # the secondary measurements must never cost the headline line
    large, past_l3, extras, cpu = None, None, {}, None
    if rank == 0 and world == 1 and args.workload != "c5_wan" and not args.no_large_shape:
        try:
            large = pre_large if pre_large is not None else measure_hbm_bound_shape(_cabi, dev)
        except Exception as e:
            large = {"error": repr(e)}
    past_l3 = pre_past
    if rank == 0 and world == 1 and args.extras:
        try:
            extras = extra_lines(args, dev)
        except Exception as e:
            extras = {"extras_error": repr(e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if not args.no_cpu_baseline:     # rank 0 only, after the process group is gone: ~20 s of host time nobody waits for
        try:
            try:
                os.sched_setaffinity(0, affinity0)      # the baseline is the reference on the BOX's host cores, not on one NUMA node's
            except Exception:
                pass
            cpu = cpu_baseline(args.workload, args.cpu_seconds)
        except Exception as e:
            cpu = {"error": repr(e)}
    n_el = int(np.prod(shape))
    kind = args.mask or ("temporal" if len(shape) == 5 else "box")
    mask_desc = {"box": "50% box mask", "blob": "centred disc mask",
                 "temporal": f"temporal mask (second half of the video inpainted: latent frames >= "
                             f"{temporal_known_frames(shape[2]) if len(shape) == 5 else 0} after the 5-tap union)"}[kind]
    parity_failed = (parity is not None and not parity.get("ok")) or (dist_info is not None and not dist_info.get("parity_ok_all_ranks", True))
    ref_gpu = rccl = None
    if world == 1 and args.extras:
        try:
            ref_gpu = reference_gpu_eager(args.workload, dev, iters_total / tmax)
        except Exception as e:
            ref_gpu = {"error": repr(e)}
        if not args.no_rccl_selftest:
            rccl = rccl_single_rank_selftest()
    line = {
        "metric": "langevin_think_iterations_per_sec",
        "value": None if parity_failed else iters_total / tmax,
        "parity_check": parity,
        "unit": "think-iterations/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * tmax / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",      # the path's arithmetic and state (with --model-dtype bf16 only the backbone's
                                                  # input / outputs are bf16 storage: config.backbone_io)
        "config": {"workload": f"{args.workload}: latent {'x'.join(map(str, shape))} per GPU, {n_sig} sigmas x "
                               f"{n_think} think iterations, {mask_desc}, stub backbone x->(0.9x,0.8x), "
                               f"{'flow' if flow else 'VE/Karras'} schedule",
                   "rng": args.rng, "mask_format": args.mask_format, "backbone_io": args.model_dtype, "launch": "hipGraph replay per sigma call" if args.graph else "eager launches",
                   "replicas": args.gpus, "rows_per_gpu": b, "global_rows": b * args.gpus, "iterations_per_step": n_sig * n_think,
                   "latent_elements_per_gpu": n_el, "lambda": HYPER["Lambda"], "beta": HYPER["Beta"],
                   "step_size": HYPER["StepSize"]},
        "latent_rows_x_iterations_per_s": iters_total * b / tmax,
        # which library carried the ranks: "rccl" (torch.distributed "nccl" on ROCm), "gloo" (a rehearsal on fewer devices than
        # ranks: NOT a scaling measurement), None at N = 1
        "collective": (None if dist_info is None else ("rccl" if dist_info["backend"] == "nccl" else dist_info["backend"])),
        "distinct_devices": (1 if dist_info is None else dist_info["distinct_devices"]),
        "host_binding": numa,
        "repeats": ({"values": repeat_values, "median": float(np.median(repeat_values)), "min": min(repeat_values),
                     "max": max(repeat_values), "note": f"further timed blocks of {args.steps} steps each, same bracketing"}
                    if repeat_values else None),
        "roofline": roofline,
        "roofline_hbm_bound_shape": large,
        "roofline_hbm_past_l3": past_l3,
        "cpu_baseline": cpu,
        # the unmodified reference on THIS GPU (eager ATen launches) next to the product: same device, same schedule, same stub
        "reference_gpu_eager": ref_gpu,
        "dist": dist_info,
        # N = 1: a one-rank RCCL group in a child process pushes a job through the collectives of the N > 1 path
        "rccl_single_rank_selftest": rccl,
    }
    line.update(extras)
    if parity_failed:
        line["error"] = ("parity_check failed: the timed configuration does not reproduce the oracle within the stated tolerance; "
                         f"no value is reported (measured {iters_total / tmax:.1f} it/s is void)")
    emit_line(line)
    if parity_failed:
        raise SystemExit(3)


def _profile_order(path):
    """Sort key of a profiles/ file: round number, then rNN_ (the round's final pass) AFTER rNNa_, rNNb_ (its earlier passes,
    kept for the box-to-box spread) -- plain string order would put `r04a_` behind `r04_`."""
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), m.group(2) == "", m.group(2)) if m else (-1, False, "")


def _latest_profile_json(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=_profile_order)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_traffic(workload):
    """HBM-side bytes per steady-state launch measured with rocprofv3 PMC counters (FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 x2 read correction) -- a committed measurement
    (profiles/r*_pmc_traffic.json, produced by scripts/gpu_profile.sh), not something bench.py can
    collect on itself.  None when no profile covers this workload."""
    data, _f = _latest_profile_json("r*_pmc_traffic.json")
    try:
        entry = (data or {}).get(workload)
        return int(entry["traffic_bytes_per_launch"]) if entry else None
    except Exception:
        return None


def rocprof_duration(workload):
    """Mean per-dispatch duration (us) of the steady kernel on this workload's shape as rocprofv3 --kernel-trace
    measured it (profiles/r*_kernel_durations.json, written by scripts/collect_profiles.py from the committed
    kernel-trace summaries).  None when no profile covers the workload."""
    data, _f = _latest_profile_json("r*_kernel_durations.json")
    try:
        entry = (data or {}).get(workload)
        return float(entry["mean_us"]) if entry else None
    except Exception:
        return None


def committed_profile(workload):
    """What profiles/ holds for this shape, as a cross-reference NEXT TO this run's own measurement (never folded into
    it): the rocprofv3 --kernel-trace mean per dispatch and the PMC bytes per launch, with the files they come from."""
    dur, f_dur = _latest_profile_json("r*_kernel_durations.json")
    pmc, f_pmc = _latest_profile_json("r*_pmc_traffic.json")
    d, t = (dur or {}).get(workload), (pmc or {}).get(workload)
    return {"rocprofv3_mean_launch_us": d.get("mean_us") if d else None, "rocprofv3_source": d.get("source") if d else None,
            "kernel_durations_file": f_dur, "pmc_traffic_bytes_per_launch": t.get("traffic_bytes_per_launch") if t else None,
            "pmc_traffic_file": f_pmc,
            "note": "committed rocprofv3 measurements of the same launch (other box, under the profiler); this run's numbers "
                    "are the event-timer ones"}


def steady_bytes_per_launch(mask_np, mask_format, n_el, every_stream=False, model_dtype=None):
    """ALGORITHMIC bytes of one steady-state launch (POST_STEADY | PRE_HALF | EMIT), two figures:
      every_stream -- SURVEY.md 8(d)'s per-unit figure for this storage: read x_t, x0, x0_BIG, y, m, C and write x_t, C, x_in
                      for EVERY element (36 B with fp32 streams and the reference's fp32 mask; the mask's own width as used:
                      0.125 B bit-packed, 1 B as bytes; half-width heads / x_in with a bf16 backbone);
      required     -- the bytes THIS job needs, from the mask actually used: an inpaint element (m = 0) reads head 0 only, a
                      known one (m = 1) head 1 and y only (lanpaint.py:182-184 with m in {0, 1}).  It applies to the launches
                      that act on it -- the region-aware streaming kernels (bit-packed mask, 16 B per lane: more than 512 Ki
                      elements, not LP_FL_NO_REGION_SKIP); every other launch streams every operand and `required` equals
                      `every_stream`.
    `roofline.frac` is computed on `required`: a fraction of peak on bytes the kernel never has to move is not a bandwidth
    fraction (the round-4 C5 line printed 1.09 that way)."""
    half = model_dtype is not None
    head, xin = (2.0, 2.0) if half else (4.0, 4.0)
    m_b = {"bits": 0.125, "u8": 1.0}.get(mask_format, 4.0)
    every = 8.0 + 8.0 + xin + 2 * head + 4.0 + m_b            # x_t, C in; x_t, C out; x_in out; two heads; y; mask
    region_aware = mask_format == "bits" and n_el > 512 * 1024 and not every_stream
    required = every
    known_frac = None
    if mask_np is not None:
        known_frac = float(np.count_nonzero(np.asarray(mask_np) > 0.5)) / float(np.asarray(mask_np).size)
        if region_aware:
            required = 8.0 + 8.0 + xin + m_b + (1.0 - known_frac) * head + known_frac * (head + 4.0)
    return {"every_stream": every * n_el, "required": required * n_el, "bytes_per_element_every_stream": every,
            "bytes_per_element_required": required, "known_fraction": known_frac, "region_aware_launch": region_aware}


def roofline_fields(bytes_alg, duration_us, traffic, bytes_every_stream=None):
    """The bandwidth statement of one launch.  `bytes_alg`: the algorithmic bytes the launch has to move
    (steady_bytes_per_launch's `required`); `achieved` / `frac` = that / duration -- at most the rate the bytes really moved
    at, so a fraction of peak.  `frac_every_stream`: the same duration against SURVEY.md 8(d)'s every-operand figure (can
    exceed what HBM delivers when the kernel skips streams; quoted for comparison with earlier rounds, never as `frac`).
    `frac_counter`: min(algorithmic, PMC-measured) bytes / duration."""
    achieved = bytes_alg / (duration_us * 1e-6) / 1e9
    moved = min(bytes_alg, traffic) if traffic else None
    counter = (moved / (duration_us * 1e-6) / 1e9) if moved else None
    every = (bytes_every_stream / (duration_us * 1e-6) / 1e9) if bytes_every_stream else None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "frac_algorithmic": achieved / HBM_PEAK_GBPS, "frac_counter": (counter / HBM_PEAK_GBPS) if counter else None,
            "frac_every_stream": (every / HBM_PEAK_GBPS) if every else None, "every_stream_GBps": every,
            "every_stream_bytes_per_launch": bytes_every_stream,
            "counter_side_GBps": counter, "frac_counter_vs_6290": (counter / 6290.0) if counter else None,
            "traffic": traffic,
            "traffic_source": "committed_constant: profiles/r*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                              "of the same launch on another box), not a live counter of this run" if traffic else None,
            "algorithmic_bytes_per_launch": bytes_alg, "duration_used_us": duration_us}


def steady_kernel_name(workload, rng, mask_format):
    """The instantiation lp_step dispatches the steady-state launch of this workload to (step_kernel.hip)."""
    n_el = int(np.prod(WORKLOADS[workload][0]))
    vec = 4 if n_el > 512 * 1024 else 1
    strided = rng == "torch" and vec == 4 and WORKLOADS[workload][0][0] == 1
    return (f"lp::lp_step_kernel<{vec}, {2 if mask_format == 'bits' else 0}, 28u, 4, {1 if rng == 'torch' else 0}, "
            f"{'true' if strided else 'false'}, 0>  (VEC, MODE: 2 = bit-packed hard mask, PH 28 = POST_STEADY|PRE_HALF|EMIT, "
            "fp32 backbone outputs, RNG: 0 = Philox 1 = torch stream, ST: generation in ATen's thread order with the values transposed "
            "through LDS (round 5; the reference's stream past ATen's grid cap), early stop: 0 = off 1 = on "
            "2 = on with the verdict folded into the launch)")


def measure_roofline(engine, _cabi, x0, y, noise, mask, sig_list, times_list, ratios, n_think, args, busy=None):
    """Instrumented replay of the timed region: every steady-state lp_step launch
    (POST_STEADY|PRE_HALF|EMIT, the dominant kernel) goes through lp_step_timed, i.e.
    hipExtLaunchKernelGGL with a HIP start/stop event pair bound to that dispatch on the
    launch stream -- the kernel's own begin->end time, the quantity rocprofv3
    --kernel-trace reports (profiles/ holds the matching summary)."""
    import ctypes
    lib = _cabi.load()
    steady = _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT
    timers, used = [], []
    orig = engine._launch_step

    def timed_launch(stream):
        if engine._desc.phases == steady:
            if timers:
                t = timers.pop()
            else:
                t = ctypes.c_void_p()
                _cabi.check(lib.lp_timer_create(ctypes.byref(t)), "lp_timer_create")
            _cabi.check(lib.lp_step_timed(ctypes.byref(engine._desc), stream, t), "lp_step_timed")
            used.append(t)
        else:
            orig(stream)

    engine._launch_step = timed_launch
    graph_was, engine.graph = engine.graph, False      # per-dispatch timers need individual (eager) launches
    try:
        for _ in range(max(1, min(args.steps, 3))):
            schedule_pass(engine, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
    finally:
        engine._launch_step = orig
        engine.graph = graph_was
    if not used:
        return None
    durs = []
    for t in used:
        ns = ctypes.c_double()
        _cabi.check(lib.lp_timer_elapsed_ns(t, ctypes.byref(ns)), "lp_timer_elapsed_ns")
        durs.append(ns.value * 1e-9)
        lib.lp_timer_destroy(t)
    durs = np.asarray(durs)
    n_el = x0.numel()
    nbytes = steady_bytes_per_launch(make_mask(tuple(x0.shape), MASK_KIND), args.mask_format, n_el)
    bytes_per_launch = nbytes["required"]
    burst_us = graph_burst_us_per_launch(_cabi, args.workload, x0.device)
    # The timed region replays hipGraphs, inside which per-dispatch events cannot be recorded, and the eager
    # replay above leaves the GPU idle between dispatches (host-paced), which stretches each dispatch (7.8 us vs
    # 4.9 us in rocprofv3's trace of the graph replays at C2).  The figure that matches the timed region -- and
    # rocprofv3 -- is the same event pair per dispatch with the launches back to back: that one is `achieved`.
    if not busy or "error" in busy:
        busy = measure_hbm_bound_shape(_cabi, x0.device, workload=args.workload, launches=120)
    out = roofline_fields(bytes_per_launch, busy["mean_launch_us"], pmc_traffic(args.workload), nbytes["every_stream"])
    prof = committed_profile(args.workload)
    rp_us = prof.get("rocprofv3_mean_launch_us")
    out.update({
        "bytes_model": nbytes,
        # the same bytes over the rocprofv3 --kernel-trace mean per dispatch committed under profiles/ (another box, under
        # the profiler), printed beside the live figure so that nobody has to recompute it
        "frac_rocprofv3": (bytes_per_launch / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if rp_us else None,
        "frac_every_stream_rocprofv3": (nbytes["every_stream"] / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if rp_us else None,
        "regime": busy.get("regime"),
        "note": "latency-bound shape: 2.4 MB per launch is cache resident, the dispatch is launch latency "
                "(see roofline_hbm_bound_shape / roofline_hbm_past_l3 for the bandwidth-bound regime)"
                if n_el <= 512 * 1024 else None,
        "eager_replay_mean_us": float(durs.mean()) * 1e6, "eager_replay_median_us": float(np.median(durs)) * 1e6,
        "eager_replay_min_us": float(durs.min()) * 1e6, "eager_replay_launches_timed": len(durs),
        "graph_burst_us_per_launch": burst_us,
        "graph_burst_GBps": bytes_per_launch / burst_us / 1e3,
        "kernel": steady_kernel_name(args.workload, "philox", args.mask_format),
        "kernel_rng": "philox (the roofline launches are the Philox2x32 variant of the step kernel whatever --rng the "
                      "timed region ran with)",
        "storage": "fp32 x_t, C, x0, x0_BIG, y, x_in; mask: " + {"bits": "1 bit", "u8": "1 byte"}.get(args.mask_format, "4 bytes (fp32)") + " / element",
        "mean_launch_us": busy["mean_launch_us"], "median_launch_us": busy["median_launch_us"],
        "min_launch_us": busy["min_launch_us"], "launches_timed": busy["launches_timed"], "warm_burst_s": busy.get("warm_burst_s"),
        "committed_profile": prof,
        "timer": "hipExtLaunchKernelGGL start/stop events per dispatch (kernel begin->end) on the launch stream, THIS run; "
                 "mean over back-to-back launches of the steady kernel on buffers of this workload's shape after a warm burst "
                 "of the same launch"})
    return out


MASK_FORMAT = "bits"          # set from --mask-format; the standalone launches follow the headline's format


def attach_mask_format(mask, fmt):
    """The job-setup step that hands the kernels a compact copy of a binary mask."""
    if fmt == "bits":
        import lanpaint_amd
        return lanpaint_amd.pack_mask(mask)
    if fmt == "u8":
        mask._lp_u8 = mask.to(torch.uint8).contiguous()
    return mask


def standalone_step(_cabi, workload, dev, phase=None, model_dtype=None):
    """A self-contained steady-state lp_step launch on synthetic buffers of `workload`'s shape
    (used for the HBM-bound supplementary roofline and by scripts/microbench_step.py)."""
    import ctypes
    lib = _cabi.load()
    shape, flow, _, _ = WORKLOADS[workload]
    n_el, rows = int(np.prod(shape)), shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    bufs = {k: torch.randn(shape, device=dev, generator=g) for k in ("x", "y", "noise", "x_t", "C", "x0", "x0b", "x_in")}
    if model_dtype is not None:          # a half-precision backbone: its two heads arrive, and x_in leaves, in that dtype
        for k in ("x0", "x0b", "x_in"):
            bufs[k] = bufs[k].to(model_dtype)
    mask = torch.from_numpy(make_mask(shape, MASK_KIND)).to(dev)
    h = _cabi.LpHyper()
    h.lambda_, h.beta, h.step_size, h.min_step_frac = HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"], 0.0
    h.is_flow, h.one_plus_lambda = int(flow), 1.0 + HYPER["Lambda"]
    sig = torch.full((rows,), 0.7 if flow else 1.5, device=dev)
    ve, abt, _ = times_from_sigma(sig, flow)
    coef = torch.empty((rows, _cabi.LP_COEF_STRIDE), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _cabi.check(lib.lp_coeffs(ctypes.byref(h), ve.data_ptr(), 1, abt.data_ptr(), 1, sig.data_ptr(), 1, None, 0, None, 0, rows,
                              coef.data_ptr(), st))
    d = _cabi.LpStepDesc()
    d.n_el, d.el_per_row, d.rows = n_el, n_el // rows, rows
    d.phases = phase or (_cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT)
    mask = attach_mask_format(mask, MASK_FORMAT)
    d.flags = (_cabi.LP_FL_FLOW if flow else 0) | (_cabi.LP_FL_NO_REGION_SKIP if os.environ.get("LANPAINT_AMD_NO_REGION_SKIP") else 0)
    d.replace_kind, d.lambda_, d.one_plus_lambda, d.beta = _cabi.LP_REPLACE_VE, h.lambda_, h.one_plus_lambda, h.beta
    d.step_size, d.noise_scale = h.step_size, 1.0
    d.coef, d.x, d.noise, d.y, d.mask = (coef.data_ptr(), bufs["x"].data_ptr(), bufs["noise"].data_ptr(),
                                         bufs["y"].data_ptr(), mask.data_ptr())
    if MASK_FORMAT == "bits":
        d.mask, d.flags = mask._lp_bits.data_ptr(), d.flags | _cabi.LP_FL_MASK_BITS
    elif MASK_FORMAT == "u8":
        d.mask, d.flags = mask._lp_u8.data_ptr(), d.flags | _cabi.LP_FL_MASK_U8
    d.x_t, d.C, d.x0, d.x0_big, d.x_in = (bufs[k].data_ptr() for k in ("x_t", "C", "x0", "x0b", "x_in"))
    if model_dtype is not None:
        half = model_dtype == torch.bfloat16
        d.flags |= (_cabi.LP_FL_X0_BF16 | _cabi.LP_FL_XIN_BF16) if half else (_cabi.LP_FL_X0_F16 | _cabi.LP_FL_XIN_F16)
    d.rng_seed = 1
    d.tune = tune_from_env(_cabi)
    keep = (bufs, mask, coef, sig, ve, abt)
    return d, keep, n_el


def tune_from_env(_cabi):
    """lp_step_desc.tune for the micro-benchmark scripts (scripts/microbench_*.py): the A/B switches used to be
    LANPAINT_AMD_TUNE_* variables read INSIDE the library; the library no longer looks at the environment, the scripts
    translate the same variables into the descriptor field."""
    t = 0
    vec = os.environ.get("LANPAINT_AMD_TUNE_VEC")
    if vec == "1":
        t |= _cabi.LP_TUNE_VEC1
    elif vec == "4":
        t |= _cabi.LP_TUNE_VEC4
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_DECIDE"):
        t |= _cabi.LP_TUNE_ES_NO_DECIDE
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_FOLD"):
        t |= _cabi.LP_TUNE_ES_NO_FOLD
    if os.environ.get("LANPAINT_AMD_TUNE_ES_NO_ATOMICS"):
        t |= _cabi.LP_TUNE_ES_NO_ATOMICS
    return t


def graph_burst_us_per_launch(_cabi, workload, dev, reps=200, replays=20, every_stream=False, model_dtype=None):
    """Un-profiled steady-state cost of one launch: `reps` launches of the steady kernel captured in a
    hipGraph on synthetic buffers of the workload's shape, replayed; wall time / launches (kernel +
    the dependent-launch boundary; a bare torch elementwise kernel costs ~1.66 us this way)."""
    import ctypes
    lib = _cabi.load()
    d, keep, _n = standalone_step(_cabi, workload, dev, model_dtype=model_dtype)
    if every_stream:
        d.flags |= _cabi.LP_FL_NO_REGION_SKIP

    def launches(n):
        st = torch.cuda.current_stream(dev).cuda_stream
        for k in range(n):
            d.rng_offset = k
            _cabi.check(lib.lp_step(ctypes.byref(d), st))

    launches(5)
    torch.cuda.synchronize(dev)
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(graph, stream=side):
        launches(reps)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize(dev)
    us = (time.perf_counter() - t0) / (replays * reps) * 1e6
    del keep
    return us


def warm_burst(lib, d, stream, dev, seconds):
    """Untimed launches of the very launch about to be timed, for `seconds`: the first launches of a process (or after
    an idle gap) run while the chip's clocks are still ramping -- 24 launches timed first in the process read 8-21 %
    longer than rocprofv3's mean over a thousand at the 1.2 GB shape."""
    import ctypes
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(32):
            d.rng_offset = n
            n += 1
            _cabi_check(lib.lp_step(ctypes.byref(d), stream))
        torch.cuda.synchronize(dev)
    return n


def _cabi_check(rc):
    from lanpaint_amd import _cabi
    _cabi.check(rc)


def timed_burst(_cabi, lib, d, stream, dev, launches, lead=16):
    """Per-dispatch durations (s) of `launches` back-to-back launches from ONE host call (`lead` more in front, not
    counted: they start on an idle chip)."""
    import ctypes
    timers = (ctypes.c_void_p * (launches + lead))()
    for k in range(launches + lead):
        t = ctypes.c_void_p()
        _cabi.check(lib.lp_timer_create(ctypes.byref(t)))
        timers[k] = t
    d.rng_offset = 100
    torch.cuda.synchronize(dev)
    _cabi.check(lib.lp_step_timed_burst(ctypes.byref(d), stream, timers, launches + lead))   # one host call: GPU stays busy
    torch.cuda.synchronize(dev)
    durs = []
    for t in timers:
        ns = ctypes.c_double()
        _cabi.check(lib.lp_timer_elapsed_ns(ctypes.c_void_p(t), ctypes.byref(ns)))
        durs.append(ns.value * 1e-9)
        lib.lp_timer_destroy(ctypes.c_void_p(t))
    return np.asarray(durs[lead:])


def shape_regime(n_el, streams=9):
    working_set = streams * 4 * n_el                  # the nine fp32 streams of the steady launch
    regime = ("past the 256 MiB Infinity Cache: every byte comes from / goes to HBM" if working_set > 2 * 256 * 2 ** 20 else
              "L3-resident: the working set fits the 256 MiB Infinity Cache, the rates are fabric-side, not DRAM-side"
              if working_set > 32 * 2 ** 20 else "cache resident (L2): launch-latency bound")
    return working_set, regime


def measure_hbm_bound_shape(_cabi, dev, workload="c5_wan", launches=100, every_stream=False, warm_s=0.3, model_dtype=None):
    """Supplementary evidence: the same steady-state kernel on the video-latent shape (75 MB of
    algorithmic traffic per launch -- the regime where the kernel is bandwidth bound, not launch
    bound), launched back to back through lp_step_timed after a warm burst of the same launch."""
    lib = _cabi.load()
    d, keep, n_el = standalone_step(_cabi, workload, dev, model_dtype=model_dtype)
    if every_stream:
        d.flags |= _cabi.LP_FL_NO_REGION_SKIP
    st = torch.cuda.current_stream(dev).cuda_stream
    warmed = warm_burst(lib, d, st, dev, warm_s)
    durs = timed_burst(_cabi, lib, d, st, dev, launches)
    shape = WORKLOADS[workload][0]
    nbytes = steady_bytes_per_launch(make_mask(shape, MASK_KIND), MASK_FORMAT, n_el, every_stream=every_stream,
                                     model_dtype=model_dtype)
    bytes_per_launch = nbytes["required"]
    del keep
    prof_key = workload + ("_every_stream" if every_stream else "") + ("_bf16" if model_dtype is not None else "")
    event_us = float(durs.mean()) * 1e6
    working_set, regime = shape_regime(n_el)
    out = roofline_fields(bytes_per_launch, event_us, pmc_traffic(prof_key), nbytes["every_stream"])
    out.update({"bytes_model": nbytes, "workload": f"{workload}: latent {'x'.join(map(str, shape))}, steady-state lp_step back to back"
                            + (", every operand streamed (LP_FL_NO_REGION_SKIP)" if every_stream else "")
                            + (", bf16 heads in / bf16 x_in out" if model_dtype is not None else ""),
                "regime": regime, "working_set_bytes": working_set, "frac_vs_6290": out["achieved"] / 6290.0,
                "hbm_side_GBps": out["counter_side_GBps"],
                "mean_launch_us": event_us, "median_launch_us": float(np.median(durs)) * 1e6,
                "min_launch_us": float(durs.min()) * 1e6, "launches_timed": int(durs.size), "warm_burst_s": warm_s,
                "warm_burst_launches": warmed, "committed_profile": committed_profile(prof_key),
                "limiter_note": ("streaming sizes: the launch is co-limited by VALU issue, not by HBM alone -- SQ_INSTS_VALU / SQ_WAVES of this "
                                 "kernel: 538 (round 3) -> 326 (round 4) -> 216 per wave in round 5 with fp32 heads, 631 -> 375 -> 267 with "
                                 "bf16 heads, 585 -> 322 with the reference's noise stream (profiles/r05_sq_*.log / .md); rocprofv3 mean per "
                                 "dispatch, same box against the round-4 library: 8.45 -> 7.88 us, 8.54 -> 7.84, 11.6 -> 10.2 "
                                 "(profiles/r05_ab_r04_vs_r05_*.log)")
                if n_el > 512 * 1024 else None})
    return out


def measure_past_l3(_cabi, dev, workload="x_wan_b16", launches=100, warm_s=0.3, rounds=2):
    """The 1.2 GB point, both variants of the launch (every operand streamed / region-aware as shipped), three clocks
    each, INTERLEAVED on this box: per-dispatch event pairs after a warm burst, the un-profiled graph-burst cost per
    launch, and the committed rocprofv3 mean for reference."""
    res = {}
    for rnd in range(rounds):
        for every in (True, False):
            key = "every_stream" if every else "region_aware"
            m = measure_hbm_bound_shape(_cabi, dev, workload=workload, launches=launches, every_stream=every, warm_s=warm_s)
            g = graph_burst_us_per_launch(_cabi, workload, dev, reps=50, replays=20, every_stream=every)
            r = res.setdefault(key, {"event_mean_us": [], "graph_burst_us": [], "last": None})
            r["event_mean_us"].append(m["mean_launch_us"])
            r["graph_burst_us"].append(g)
            r["last"] = m
            torch.cuda.empty_cache()
    out = res["every_stream"]["last"]
    for key in ("every_stream", "region_aware"):
        r, m = res[key], res[key]["last"]
        ev, gb = float(np.mean(r["event_mean_us"])), float(np.mean(r["graph_burst_us"]))
        blk = roofline_fields(m["algorithmic_bytes_per_launch"], ev, m["traffic"], m["every_stream_bytes_per_launch"])
        blk["bytes_model"] = m["bytes_model"]
        blk.update({"event_mean_us_per_round": r["event_mean_us"], "graph_burst_us_per_round": r["graph_burst_us"],
                    "event_mean_us": ev, "graph_burst_us_per_launch": gb,
                    "rocprofv3_mean_launch_us": m["committed_profile"]["rocprofv3_mean_launch_us"],
                    "event_over_rocprofv3": (ev / m["committed_profile"]["rocprofv3_mean_launch_us"])
                    if m["committed_profile"]["rocprofv3_mean_launch_us"] else None,
                    "graph_burst_over_rocprofv3": (gb / m["committed_profile"]["rocprofv3_mean_launch_us"])
                    if m["committed_profile"]["rocprofv3_mean_launch_us"] else None})
        if key == "every_stream":
            out.update(blk)
            out["mean_launch_us"] = ev
        else:
            blk["note"] = ("same launch with the wave-uniform stream skipping on (the default): `frac` is on the bytes this mask "
                           "requires (bytes_model.required), `frac_every_stream` on the every-operand figure the launch no longer "
                           "has to move, `frac_counter` on the PMC bytes")
            out["region_aware_streams"] = blk
    out["interleaved"] = f"{rounds} rounds of [every-stream events, every-stream graph burst, region-aware events, region-aware graph burst]"
    return out


def extra_lines(args, dev):
    """Secondary numbers of the same build (N = 1 only; not the headline):
    node_default_schedule -- C2 driven through KSamplerX0Inpaint with the node defaults
        (MinStepFrac = 1.0 => n_eff = round(N (1 - abt)), last sigma skipped: SURVEY.md 8d second line);
    with_backbone -- BASELINE configs[0] shape (1x4x64x64, 20 sigmas x 5) in front of a random-init
        SD1.5-shaped dummy UNet in bf16 (tests/dummy_unet.py), the stand-in (ii) of SURVEY.md 8d."""
    from lanpaint_amd import LanPaint
    from lanpaint_amd import nodes as lpn
    out = {}
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    # ---- the headline workload with the engine's DEFAULT noise source (rng="torch": the reference's own
    # torch.randn_like stream, generated inside the kernel) when the headline ran with another one
    if args.rng != "torch":
        try:
            shape, flow, n_sig, n_think = WORKLOADS[args.workload]
            sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
            x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
            mask = attach_mask_format(mask, args.mask_format)
            sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
            times_list = [times_from_sigma(s, flow) for s in sig_list]
            ratios = euler_ratios(sig_list, len(shape))
            eng = LanPaint(StubBackbone(flow), n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                           IS_FLOW=flow, rng="torch", graph=bool(args.graph))
            for _ in range(5):
                schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            it0, t0, reps = eng.iterations_run, time.perf_counter(), max(5, args.steps // 2)
            for _ in range(reps):
                schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["reference_noise_stream"] = {
                "value": (eng.iterations_run - it0) / dt, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps,
                "note": f"{args.workload} as in the headline but rng='torch' (the engine default): the values "
                        "torch.randn_like(x_t) returns for the device generator, bit for bit, generated in-kernel"}
        except Exception as e:
            out["reference_noise_stream"] = {"error": repr(e)}
    # ---- what a drop-in user gets: the engine built with NO optional keyword (graph="auto", rng="torch"), called the way
    # the reference engine is called (plain fp32 mask, the positional signature of lanpaint.py:8 / :44)
    try:
        shape, flow, n_sig, n_think = WORKLOADS[args.workload]
        sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
        x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
        sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
        times_list = [times_from_sigma(s, flow) for s in sig_list]
        ratios = euler_ratios(sig_list, len(shape))
        res = {}
        for label, m in (("fp32_mask", mask), ("packed_mask", attach_mask_format(mask.clone(), "bits"))):
            eng = LanPaint(StubBackbone(flow), n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                           False, flow)
            for _ in range(5):
                schedule_pass(eng, x0, y, noise, m, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            it0, t0, reps = eng.iterations_run, time.perf_counter(), max(5, args.steps // 2)
            for _ in range(reps):
                schedule_pass(eng, x0, y, noise, m, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[label] = {"value": (eng.iterations_run - it0) / dt, "ms_per_step": 1e3 * dt / reps,
                          "captured_calls": len(eng._graphs), "graph_blocked": bool(eng._graph_blocked)}
        out["engine_defaults"] = {
            "value": res["fp32_mask"]["value"], "unit": "think-iterations/s", "ms_per_step": res["fp32_mask"]["ms_per_step"],
            "detail": res,
            "note": f"{args.workload}: LanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize, IS_FLUX, IS_FLOW) -- no optional "
                    "keyword: graph='auto' (captured from the job's second sigma call on, after a check against eager launches), "
                    "rng='torch' (the reference's noise stream, in-kernel), the reference's fp32 mask -- which the engine, seeing the "
                    "same binary mask tensor on the job's second sigma call, bit-packs by itself (round 4: auto_pack_mask; one host "
                    "read per mask tensor); packed_mask: the same engine with the mask packed by the caller before the first call "
                    "(lanpaint_amd.pack_mask, what KSamplerX0Inpaint does)"}
    except Exception as e:
        out["engine_defaults"] = {"error": repr(e)}
    # ---- the headline workload with the inner early stop armed but never firing (threshold far below any distance):
    # what the device-side stop rule costs per iteration (LP_FL_ES: three more streams, the block reduction, the decision)
    try:
        shape, flow, n_sig, n_think = WORKLOADS[args.workload]
        sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
        x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
        mask = attach_mask_format(mask, args.mask_format)
        sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
        times_list = [times_from_sigma(s, flow) for s in sig_list]
        ratios = euler_ratios(sig_list, len(shape))
        eng = LanPaint(StubBackbone(flow), n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                       IS_FLOW=flow, EarlyStopThreshold=1e-30, EarlyStopPatience=1, rng=args.rng, philox_seed=args.seed,
                       graph=bool(args.graph))
        for _ in range(5):
            schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        it0, t0, reps = eng.iterations_run, time.perf_counter(), max(5, args.steps // 4)
        for _ in range(reps):
            schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["inner_early_stop_armed"] = {
            "value": (eng.iterations_run - it0) / dt, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps,
            "note": f"{args.workload} as in the headline with EarlyStopThreshold > 0 (never reached): the stop rule is "
                    "evaluated on the device inside every replayed launch, no host read in the loop"}
    except Exception as e:
        out["inner_early_stop_armed"] = {"error": repr(e)}
    # ---- the headline workload behind a bf16 backbone (BASELINE configs[1] says bf16: that is the backbone's dtype): the kernels emit
    # x_in and read both heads as bf16; checked against the oracle with the stub restated in bf16 before it is timed
    if args.model_dtype == "f32":
        try:
            shape, flow, n_sig, n_think = WORKLOADS[args.workload]
            sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
            x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
            mask = attach_mask_format(mask, args.mask_format)
            sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
            times_list = [times_from_sigma(s, flow) for s in sig_list]
            ratios = euler_ratios(sig_list, len(shape))
            eng = LanPaint(StubBackbone(flow), n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                           IS_FLOW=flow, rng=args.rng, philox_seed=args.seed, graph=bool(args.graph), model_dtype=torch.bfloat16)
            par = parity_check(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow,
                               max_sigmas=None if int(np.prod(shape)) <= 512 * 1024 else 2)
            for _ in range(5):
                schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            it0, t0, reps = eng.iterations_run, time.perf_counter(), max(5, args.steps // 2)
            for _ in range(reps):
                schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["bf16_backbone"] = {
                "value": (eng.iterations_run - it0) / dt if par["ok"] else None, "unit": "think-iterations/s",
                "ms_per_step": 1e3 * dt / reps, "parity_check": {k: par[k] for k in ("mse_x", "mse_denoised_max", "ok", "sigmas_checked", "launch_modes")},
                "note": f"{args.workload} as in the headline with model_dtype=torch.bfloat16: x_in emitted and both heads read as bf16 "
                        "(state, arithmetic and the written-back x stay fp32); the oracle side of the check restates the stub in bf16"}
        except Exception as e:
            out["bf16_backbone"] = {"error": repr(e)}
    # ---- a half-precision backbone: both heads arrive as bf16 and x_in leaves as bf16 (30 B / element instead of 36);
    # the production storage widths -- BASELINE configs[1] says bf16 -- at the two bandwidth-bound shapes
    try:
        from lanpaint_amd import _cabi as _c
        out["bf16_heads"] = {
            wl: measure_hbm_bound_shape(_c, dev, workload=wl, launches=100, model_dtype=torch.bfloat16)
            for wl in ("c5_wan", "x_wan_b16")}
        out["bf16_heads"]["note"] = ("steady lp_step launch with LP_FL_X0_BF16 | LP_FL_XIN_BF16 (x0, x0_BIG read and x_in "
                                     "written as bf16; state x_t, C, y stay fp32): 30 algorithmic bytes per element")
        torch.cuda.empty_cache()
    except Exception as e:
        out["bf16_heads"] = {"error": repr(e)}
    # ---- node-default schedule through the sampler-facing callable
    shape, flow, n_sig, n_think = WORKLOADS["c2_sdxl"]
    sig_np = karras_sigmas(n_sig)
    x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
    sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
    ratios = euler_ratios(sig_list, 4)
    model = StubBackbone(flow)
    model.model_type = "EPS"
    k = lpn.KSamplerX0Inpaint(model, torch.cat([tt(sig_np), torch.zeros(1, device=dev)]))
    k.latent_image, k.noise = y, noise
    k.PaintMethod = LanPaint(model, n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                             MinStepFrac=1.0, rng=args.rng, philox_seed=args.seed, graph=bool(args.graph))
    k.LanPaint_early_stop, k.LanPaint_min_step_frac = 1, 1.0
    denoise_mask = 1.0 - mask
    model_options = {}                     # ComfyUI hands the SAME dict to every step

    def node_pass():
        x = x0.clone()
        for i in range(n_sig):
            den = k(x, sig_list[i], denoise_mask, model_options=model_options, seed=args.seed)
            if i + 1 < n_sig:
                x = torch.lerp(den, x, ratios[i])
        return x

    for _ in range(8):          # (captures for every inner-step count of the ramp, then a few steady passes: the first ones read 10-15 % low)
        node_pass()
    torch.cuda.synchronize()
    it0, t0, reps = k.PaintMethod.iterations_run, time.perf_counter(), max(20, args.steps // 8)   # (5 passes = 6 ms read +-10 %)
    for _ in range(reps):
        node_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iters = k.PaintMethod.iterations_run - it0
    out["node_default_schedule"] = {
        "value": iters / dt, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps,
        "iterations_per_step": iters // reps,
        "note": "C2 through KSamplerX0Inpaint, MinStepFrac=1.0, EarlyStop=1 (n_eff = round(5(1-abt)), last sigma 0); "
                "the n_eff rule needs sigma's position from the device once per sigma: one call into the library per sigma "
                "(lp_node_call) queues the call for a speculated count, the device checks the guess and voids a miss"}
    # ---- BASELINE configs[1] / [3] with a backbone that exercises what the path was built for (round 5): SDXL 1x4x128x128, a
    # random-init SDXL-SHAPED stand-in with bf16 weights (three levels of ResBlocks, self- + cross-attention at 32 x 32: MIOpen /
    # hipBLASLt / SDPA on the matrix cores), ONE batched cond + uncond pass per call handed over as FusedCFGHeads (the kernels form
    # both CFG heads), ComfyUI's data flow around the network (bf16 in, fp32 denoised predictions out).
    # Parity, two statements: (i) the SAME architecture with fp32 weights against the oracle driving that module -- the
    # Langevin path behind a real backbone, strict bound; (ii) the bf16 network against the oracle driving it, next to the
    # network's own run-to-run noise (two engine passes from one seed): bf16 kernels of MIOpen / hipBLASLt are not bitwise
    # reproducible and one flipped bf16 rounding of eps is amplified by sigma and the CFG scale, so (ii) is bounded by what
    # the backbone itself does, not by 1e-5.  Then it/s and where a sigma call's time goes: the same number of backbone passes
    # alone (one hipGraph of n + 1 forwards) against the whole call.
    try:
        from tests.sdxl_standin import SDXLShapedBackbone
        shape, flow, n_sig, n_think = WORKLOADS["c2_sdxl"]
        sig_np = karras_sigmas(n_sig)
        x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
        mask = attach_mask_format(mask, args.mask_format)
        sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
        times_list = [times_from_sigma(s, flow) for s in sig_list]
        ratios = euler_ratios(sig_list, 4)
        mk = lambda net, **kw: LanPaint(net, n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],   # noqa: E731
                                        philox_seed=args.seed, graph=bool(args.graph), **kw)
        net32 = SDXLShapedBackbone(dev, flow=flow, dtype=torch.float32)
        par32 = parity_check(mk(net32, rng=args.rng), x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=6,
                             oracle_model=net32.as_oracle_model())
        del net32
        torch.cuda.empty_cache()
        net = SDXLShapedBackbone(dev, flow=flow)
        eng = mk(net, rng=args.rng)
        par = parity_check(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think, flow, max_sigmas=6,
                           oracle_model=net.as_oracle_model())
        # the backbone's own noise floor: two passes of ONE engine from one torch seed over the same 6 sigma calls
        eng_t = mk(net, rng="torch")
        finals = []
        for _ in range(2):
            torch.manual_seed(1234)
            finals.append(schedule_pass(eng_t, x0, y, noise, mask, sig_list[:6], times_list[:6], ratios[:5], n_think).double())
        self_mse = float(((finals[0] - finals[1]) ** 2).mean())
        for _ in range(2):
            schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        it0, t0, reps = eng.iterations_run, time.perf_counter(), 3
        for _ in range(reps):
            xl = schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        call_ms = 1e3 * dt / (reps * n_sig)
        # the backbone alone: n_think + 1 forward passes per sigma call, captured like the engine captures them
        gb, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
        for _ in range(2):
            net.predict(x0, sig_list[3])
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.graph(gb, stream=side):
            for _ in range(n_think + 1):
                keep = net.predict(x0, sig_list[3])
        for _ in range(3):
            gb.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            gb.replay()
        torch.cuda.synchronize()
        bb_ms = 1e3 * (time.perf_counter() - t0) / 30
        del keep
        ok = bool(par32["ok"] and np.isfinite(par["mse_x"]) and par["mse_x"] <= max(PARITY_TOL, 20.0 * self_mse))
        out["sdxl_shaped_backbone"] = {
            "value": (eng.iterations_run - it0) / dt if ok else None, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps,
            "sigma_call_ms": call_ms, "backbone_only_ms_per_sigma_call": bb_ms, "backbone_passes_per_sigma_call": n_think + 1,
            # (the difference of two ~15 ms wall-clock figures: it comes out at +-0.02 ms, i.e. the Langevin path is below the
            # backbone's own run-to-run noise; the rocprofv3 kernel trace of the same workload puts the lp:: kernels at 0.17 % of
            # the GPU time -- profiles/r05_sdxl_standin_time_split.md)
            "langevin_path_ms_per_sigma_call": max(0.0, call_ms - bb_ms), "langevin_path_share_of_sigma_call": max(0.0, (call_ms - bb_ms) / call_ms),
            "langevin_path_share_kernel_trace": 0.0017, "langevin_path_share_source": "profiles/r05_sdxl_standin_time_split.md (committed rocprofv3 pass)",
            "finite": bool(torch.isfinite(xl).all()), "captured_calls": len(eng._graphs), "backbone_parameters": net.n_params,
            "parity_check_fp32_weights": {k: par32[k] for k in ("mse_x", "mse_denoised_max", "ok", "sigmas_checked", "launch_modes")},
            "parity_check_bf16_weights": {k: par[k] for k in ("mse_x", "mse_denoised_max", "sigmas_checked", "launch_modes")},
            "backbone_run_to_run_mse_bf16_weights": self_mse,
            "parity_note": "fp32 weights: strict (MSE < 1e-5; the Langevin path behind a real backbone).  bf16 weights: the network's "
                           "kernels are not bitwise reproducible and a flipped bf16 rounding of eps is amplified by sigma and the CFG "
                           "scale; the engine-vs-oracle MSE is to be read next to the backbone's own run-to-run MSE (two engine passes, "
                           "one seed, same 6 sigma calls)",
            "kernel_flags": "LP_FL_CFG_FUSED | LP_FL_MASK_BITS, fp32 predictions (ComfyUI forms the denoised in fp32)",
            "backbone": "tests/sdxl_standin.py: random-init SDXL-shaped UNet stand-in (128/256/512 channels at 128/64/32 px, GroupNorm-SiLU-conv "
                        "ResBlocks, self-attention over 1024 tokens + cross-attention to [77, 2048] text states, ADM vector [2816]), "
                        "bf16 weights and activations, one batched cond + uncond pass per call, latent 1x4x128x128, 30 sigmas x 5 "
                        "(BASELINE configs[1]); MFMA-busy of its kernels vs the lp:: kernels: profiles/r05_sdxl_standin_pmc_mfma.md"}
        del net, eng, eng_t
        torch.cuda.empty_cache()
    except Exception as e:                       # the stand-in is not a deliverable; never fail the bench on it
        out["sdxl_shaped_backbone"] = {"error": repr(e)}
    # ---- dummy UNet backbone on the SD1.5 shape
    try:
        from tests.dummy_unet import DummyUNetBackbone
        shape, flow, n_sig, n_think = WORKLOADS["c1_sd15"]
        sig_np = karras_sigmas(n_sig)
        x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), args.seed, dev, tt)
        sig_list = [torch.full((1,), float(s), dtype=torch.float32, device=dev) for s in sig_np]
        times_list = [times_from_sigma(s, flow) for s in sig_list]
        ratios = euler_ratios(sig_list, 4)
        net = DummyUNetBackbone(dev, flow=flow)
        eng = LanPaint(net, n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                       rng=args.rng, philox_seed=args.seed, graph=bool(args.graph))
        for _ in range(2):
            schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        it0, t0, reps = eng.iterations_run, time.perf_counter(), 3
        for _ in range(reps):
            xl = schedule_pass(eng, x0, y, noise, mask, sig_list, times_list, ratios, n_think)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["with_backbone"] = {
            "value": (eng.iterations_run - it0) / dt, "unit": "think-iterations/s", "ms_per_step": 1e3 * dt / reps,
            "finite": bool(torch.isfinite(xl).all()),
            "backbone": "random-init SD1.5-shaped dummy UNet (conv/GroupNorm/SiLU + 1 self-attention block, 1.3 M "
                        "params, bf16, dual-head output), latent 1x4x64x64, 20 sigmas x 5 (BASELINE configs[0] shape)"}
    except Exception as e:                       # the stand-in is not a deliverable; never fail the bench on it
        out["with_backbone"] = {"error": repr(e)}
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def _usable_cpus():
    """CPUs this process can really use: its affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(workload, budget_s):
    """The reference's CPU path next to the GPU number: the UNMODIFIED reference engine
    -- oracle/_ref, the reference's own lanpaint.py compiled to bytecode where it lies by oracle/build_ref.py -- driven over
    the same schedule with the same stub on this box's host cores, at 1 thread and at os.cpu_count() threads: one warm-up
    pass discarded, median of 5 timed passes (`kind: "reference"`).  Without oracle/_ref (a checkout that never saw
    /root/reference) the CPU port of the reference (oracle/lanpaint_oracle.py on torch-CPU tensors) stands in
    (`kind: "port"`).  Bounded: when five passes of the whole schedule would not fit `budget_s` per thread setting the
    sample is the first sigma calls of the schedule, and says so.  The port is timed beside the reference at 1 thread
    (median of 3) so the ratio between the two is a number of THIS run."""
    from oracle.lanpaint_oracle import OracleLanPaint, TorchBackend
    from oracle import ref_engine
    shape, flow, n_sig, n_think = WORKLOADS[workload]
    sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))   # noqa: E731
    x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), 0, "cpu", tt)
    b = shape[0]
    sig_list = [torch.full((b,), float(s), dtype=torch.float32) for s in sig_np]
    times_list = [times_from_sigma(s, flow) for s in sig_list]
    ratios = euler_ratios(sig_list, len(shape))
    ref_cls = ref_engine.load_reference()

    def make(kind):
        if kind == "reference":     # the reference's own constructor (lanpaint.py:8)
            return ref_cls(StubBackbone(flow), HYPER["NSteps"], HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                           IS_FLUX=False, IS_FLOW=flow, MinStepFrac=HYPER["MinStepFrac"])
        return OracleLanPaint(StubBackbone(flow), HYPER["NSteps"], HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"],
                              HYPER["StepSize"], is_flow=flow, min_step_frac=HYPER["MinStepFrac"], backend=TorchBackend())

    def timed(eng, n_sigmas, passes):
        """it/s of each of `passes` passes over the first n_sigmas sigma calls, after one discarded pass."""
        vals = []
        for k in range(passes + 1):
            t0 = time.perf_counter()
            schedule_pass(eng, x0, y, noise, mask, sig_list[:n_sigmas], times_list[:n_sigmas], ratios[:max(0, n_sigmas - 1)], n_think)
            dt = time.perf_counter() - t0
            if k:
                vals.append(n_sigmas * n_think / dt)
        return vals

    kind = "reference" if ref_cls is not None else "port"
    import warnings
    warnings.filterwarnings("ignore", message="In CPU autocast")      # (the reference wraps its loop in torch.autocast(fp32))
    saved = torch.get_num_threads()
    # "all threads" = the CPUs this process may run on (a rank pinned to its GPU's NUMA node must not start one thread per CPU
    # of the whole box: oversubscribed OpenMP teams turn every tiny op into scheduler round trips)
    n_cpu, n_usable = os.cpu_count() or 1, _usable_cpus()
    per_threads, skipped, port_vals, sample_sigmas = {}, {}, None, n_sig
    t_leg = time.perf_counter()
    # 1 thread, then "all threads"; when the all-threads team is pathological on this host (the per-op probe below: a
    # container may show 256 CPUs and schedule far fewer) the next smaller team of the ladder is tried instead
    ladder = [1] + [t for t in dict.fromkeys([n_usable, 64, 16, 8]) if 1 < t <= n_usable]
    try:
        for threads in ladder:
            if threads != 1 and any(t != 1 for t in per_threads):
                break                                           # one multi-thread setting has been sampled
            torch.set_num_threads(threads)
            share = budget_s / 2                                # per thread setting: 1 discarded + 5 timed passes
            # what does ONE small elementwise op cost at this thread count?  (a sigma call is ~170 of them per think iteration;
            # on a many-core host the fork / join of a large team can cost more than the op)
            a = x0 * 0.9
            for _ in range(5):
                a = x0 * 0.9 + y
            t0 = time.perf_counter()
            for _ in range(20):
                a = x0 * 0.9 + y
            per_op = (time.perf_counter() - t0) / 40
            est_sigma = per_op * 170 * n_think
            if threads != 1 and 3 * est_sigma > share:
                skipped[str(threads)] = {"per_small_op_us": 1e6 * per_op, "estimated_s_per_sigma_call": est_sigma,
                                         "note": f"not sampled: at {threads} threads one small elementwise op costs "
                                                 f"{1e6 * per_op:.0f} us on this host, a sigma call ~{est_sigma:.1f} s -- more than the "
                                                 f"{share:.1f} s this setting may take"}
                continue
            eng = make(kind)
            per_sigma = float("inf")                            # one sigma call, twice (the first also wakes the thread pool up)
            for _ in range(2):
                t0 = time.perf_counter()
                schedule_pass(eng, x0, y, noise, mask, sig_list[:1], times_list[:1], [], n_think)
                per_sigma = min(per_sigma, time.perf_counter() - t0)
            n_s = n_sig if 6 * n_sig * per_sigma <= share else max(1, int(share / (6 * per_sigma)))
            passes = 5 if 6 * n_s * per_sigma <= 2 * share else max(1, min(5, int(2 * share / (n_s * per_sigma)) - 1))
            sample_sigmas = min(sample_sigmas, n_s)
            vals = timed(eng, n_s, passes)
            per_threads[threads] = {"median_it_s": float(np.median(vals)), "min_it_s": min(vals), "max_it_s": max(vals),
                                    "passes": len(vals), "sigma_calls_per_pass": n_s, "per_small_op_us": 1e6 * per_op}
        if kind == "reference" and time.perf_counter() - t_leg < 2 * budget_s:      # the port beside it, 1 thread
            torch.set_num_threads(1)
            port_vals = timed(make("port"), per_threads[1]["sigma_calls_per_pass"], 3)
    finally:
        torch.set_num_threads(saved)
    best_threads = max(per_threads, key=lambda t: per_threads[t]["median_it_s"])
    whole = sample_sigmas == n_sig
    detail = "; ".join(f"{t} thread(s): median {v['median_it_s']:.1f} it/s of {v['passes']} passes "
                       f"({v['min_it_s']:.1f} .. {v['max_it_s']:.1f})" for t, v in sorted(per_threads.items()))
    engine_desc = ("the UNMODIFIED reference engine (oracle/_ref: /root/reference/src/LanPaint/lanpaint.py compiled to "
                   "bytecode by oracle/build_ref.py)" if kind == "reference"
                   else "oracle/lanpaint_oracle.py (CPU port of the reference) on torch-CPU fp32 tensors -- oracle/_ref is not "
                        "staged in this checkout")
    out = {"value": per_threads[best_threads]["median_it_s"], "unit": "think-iterations/s", "cores": best_threads, "kind": kind,
           "threads": {str(t): v for t, v in sorted(per_threads.items())}, "threads_not_sampled": skipped or None,
           "host_cpus": n_cpu, "usable_cpus": n_usable, "cpu_model": _cpu_model(), "leg_seconds": time.perf_counter() - t_leg,
           "sample": f"{'whole passes' if whole else f'the first {sample_sigmas} sigma calls'} of the {workload} schedule "
                     f"({n_sig} sigmas x {n_think}), stub backbone, {engine_desc}; one warm-up pass discarded, median of 5; {detail}"}
    if kind == "reference":
        m = ref_engine.manifest() or {}
        out["reference_source_sha256"] = {k: v.get("source_sha256") for k, v in m.get("modules", {}).items()}
    if port_vals:
        port = float(np.median(port_vals))
        out["port_1_thread_it_s"] = port
        out["port_over_reference"] = port / per_threads[1]["median_it_s"]
    return out


def reference_gpu_eager(workload, dev, product_it_s, budget_s=20.0, passes=5):
    """The UNMODIFIED reference engine (oracle/_ref) driven over the same schedule ON THIS GPU -- the same eager ATen
    launches (~164 per think iteration, its own torch.randn_like draws, one host sync per iteration) a ComfyUI user of the
    reference gets on this device -- so that the line carries product vs reference on the SAME device next to product vs
    CPU.  Same stub backbone object type, same inputs, fp32; one pass discarded, median of `passes`, each bracketed by
    torch.cuda.synchronize(); outside the timed region of the headline.  Bounded like cpu_baseline: when the passes would
    not fit `budget_s` the sample is the first sigma calls of the schedule, and says so."""
    from oracle import ref_engine
    ref_cls = ref_engine.load_reference()
    if ref_cls is None:
        return {"error": "oracle/_ref is not staged in this checkout (built from /root/reference by __graft_entry__.build())"}
    import warnings
    warnings.filterwarnings("ignore", message="In CUDA autocast")
    warnings.filterwarnings("ignore", message=".*autocast.*")
    shape, flow, n_sig, n_think = WORKLOADS[workload]
    sig_np = flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
    sig_list = [torch.full((shape[0],), float(s), dtype=torch.float32, device=dev) for s in sig_np]
    times_list = [times_from_sigma(s, flow) for s in sig_list]
    ratios = euler_ratios(sig_list, len(shape))
    eng = ref_cls(StubBackbone(flow), n_think, HYPER["Friction"], HYPER["Lambda"], HYPER["Beta"], HYPER["StepSize"],
                  IS_FLUX=False, IS_FLOW=flow, MinStepFrac=HYPER["MinStepFrac"])

    def one_pass(n_s):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        xl = schedule_pass(eng, x0, y, noise, mask, sig_list[:n_s], times_list[:n_s], ratios[:max(0, n_s - 1)], n_think)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, xl

    state = torch.cuda.get_rng_state(dev)
    try:
        torch.manual_seed(0)
        per_sigma, _ = one_pass(1)
        per_sigma = min(per_sigma, one_pass(1)[0])
        n_s = n_sig if (passes + 1) * n_sig * per_sigma <= budget_s else max(1, int(budget_s / ((passes + 1) * per_sigma)))
        one_pass(n_s)                                         # discarded
        vals, xl = [], None
        for _ in range(passes):
            dt, xl = one_pass(n_s)
            vals.append(n_s * n_think / dt)
        finite = bool(torch.isfinite(xl).all())
    finally:
        torch.cuda.set_rng_state(state, dev)
    med = float(np.median(vals))
    m = ref_engine.manifest() or {}
    return {"value": med, "unit": "think-iterations/s", "min_it_s": min(vals), "max_it_s": max(vals), "passes": len(vals),
            "sigma_calls_per_pass": n_s, "ms_per_sigma_call": 1e3 * n_think / med, "finite": finite,
            "device": torch.cuda.get_device_name(dev), "dtype": "f32", "launch": "eager ATen launches (the reference has no other mode)",
            "product_over_reference_same_gpu": (product_it_s / med) if (product_it_s and med > 0) else None,
            "reference_source_sha256": {k: v.get("source_sha256") for k, v in m.get("modules", {}).items()},
            "sample": f"{'whole passes' if n_s == n_sig else f'the first {n_s} sigma calls'} of the {workload} schedule ({n_sig} sigmas x "
                      f"{n_think}), stub backbone, the UNMODIFIED reference engine (oracle/_ref) on {dev}; one pass discarded, "
                      f"median of {len(vals)}"}


def gpu_clocks(dev_index):
    """Current shader / memory clock of the device (MHz) as the driver reports them in sysfs (pp_dpm_sclk / pp_dpm_mclk: the
    line marked '*'), best effort; None where the files are not readable."""
    bdf = _pci_bus_id(dev_index)
    out = {}
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        try:
            for ln in open(f"/sys/bus/pci/devices/{bdf}/{name}"):
                if ln.rstrip().endswith("*"):
                    out[key] = int(re.search(r"(\d+)\s*Mhz", ln, flags=re.I).group(1))
        except Exception:
            pass
    if not out:
        try:
            out["sclk_mhz"] = int(torch.cuda.clock_rate(dev_index))
        except Exception:
            pass
    return out or None


def rccl_single_rank_selftest(timeout_s=150):
    """First contact with RCCL at N = 1: a CHILD process brings up a ONE-rank "nccl" process group on this
    GPU and pushes a job through the very functions the N > 1 path uses -- lanpaint_amd.distributed.broadcast_job
    (broadcast_object_list + ONE packed uint8 device broadcast), reduce_throughput (two fp64 device all-reduces) and
    gather_rank_reports (all_gather_object) -- and checks the tensors come back byte-identical.  In a child with a time limit so
    that a library fault can never cost the headline line."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, "-c",
                            "import sys, json; sys.path.insert(0, %r); from lanpaint_amd import distributed as d; "
                            "print(json.dumps(d.single_rank_selftest()))" % ROOT],
                           env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"ok": False, "error": f"child exited {p.returncode}", "stderr_tail": p.stderr[-600:]}
        out = json.loads(lines[-1])
        out["wall_s"] = time.perf_counter() - t0
        return out
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": f"no answer within {timeout_s} s"}
    except Exception as e:
        return {"ok": False, "error": repr(e)}


_LINE_FD = None


def claim_stdout():
    """ONE JSON line on stdout, nothing else: libraries write to file descriptor 1 behind Python's back (gloo prints
    "[Gloo] Rank 0 is connected to ..." there), so fd 1 is pointed at stderr for the whole run and the line goes to the
    saved descriptor."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(line):
    data = (json.dumps(line) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


def _pci_bus_id(index):
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, port, base=None):
    """The environment of each of the `n` ranks `python bench.py --gpus n` starts when no launcher did: what
    torch.distributed.run would export (one process per GPU, rendezvous on 127.0.0.1)."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "LANPAINT_BENCH_LAUNCHER": "bench.py self-spawn"})
        envs.append(env)
    return envs


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves -- one
    child process per GPU running this very file, rank 0's stdout (the JSON line) passed through -- and wait for all
    of them.  A rank that fails takes the others down (by PID) and the exit code is its code."""
    import subprocess
    envs = rank_environments(n, _free_port())
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                              stdout=None if r == 0 else subprocess.DEVNULL) for r, env in enumerate(envs)]
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (C2: 400 x 1.26 ms = 0.5 s)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="further timed blocks of --steps steps, reported as `repeats` (spread)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed set-up (graph capture, lazy init, clock ramp) before the warm-up steps")
    ap.add_argument("--workload", default="c2_sdxl", choices=sorted(WORKLOADS))
    ap.add_argument("--rng", default="philox", choices=["philox", "torch", "torch-eager"],
                    help="philox: independent in-kernel stream; torch: the device generator's randn stream reproduced "
                         "inside the kernel (the engine's default); torch-eager: torch.randn_like tensors, one launch per draw")
    ap.add_argument("--graph", type=int, default=1, help="1: replay each sigma call as one hipGraph (default); 0: eager launches")
    ap.add_argument("--mask-format", default="bits", choices=["bits", "u8", "f32"],
                    help="how the (binary) latent mask is streamed by the kernels: bit-packed once per job by "
                         "lanpaint_amd.pack_mask (what KSamplerX0Inpaint does), one byte, or the reference's fp32")
    ap.add_argument("--mask", default=None, choices=["box", "temporal", "blob"],
                    help="synthetic mask; default: 50 %% box for image latents, second half of the video inpainted for video latents")
    ap.add_argument("--model-dtype", default="f32", choices=["f32", "bf16"],
                    help="storage of the latent handed to the backbone and of its two outputs (bf16: the kernels emit / read half "
                         "width, 30 instead of 36 B per element and iteration; state and arithmetic stay fp32)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="budget of the cpu_baseline leg (split between the 1-thread and the all-threads setting)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (default).  With fewer GPUs than ranks RCCL cannot run (duplicate device): the run exits "
                         "non-zero unless --allow-gloo-fallback (or gloo) is given; the line's top-level `collective` says which "
                         "library carried the ranks")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="with --dist-backend nccl on a box with fewer GPUs than ranks: carry the ranks over gloo instead of "
                         "exiting with an error (a rehearsal of the multi-rank path, not a scaling measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rccl-selftest", action="store_true",
                    help="N = 1: skip the one-rank RCCL group that exercises the collectives of the N > 1 path in a child process")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the schedule pass against the CPU oracle that precedes the timed region")
    ap.add_argument("--parity-sigmas", type=int, default=0,
                    help="sigma calls of the schedule the parity pass covers (0: all of them up to 512 Ki latent elements, "
                         "the first 2 above)")
    ap.add_argument("--no-numa-bind", action="store_true",
                    help="do not pin the process (each rank) to the CPU cores of its GPU's NUMA node")
    ap.add_argument("--extras", type=int, default=1, help="1: also report node_default_schedule and with_backbone (N=1)")
    ap.add_argument("--no-large-shape", action="store_true", help="skip the supplementary c5_wan-shape roofline")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher (the torch.distributed.run form keeps working -- it sets WORLD_SIZE)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    global MASK_FORMAT, MASK_KIND
    MASK_FORMAT, MASK_KIND = args.mask_format, args.mask
    claim_stdout()
    run_gpu(args)


if __name__ == "__main__":
    main()
