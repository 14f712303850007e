"""BASELINE.json's configurations as synthetic jobs: shapes, sigma schedules, the stub backbone, masks, inputs,
and ONE pass of the hot path over a workload's sigma schedule (`schedule_pass` = one bench step)."""
from __future__ import annotations

import numpy as np
import torch

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


def karras_sigmas(n, sigma_min=0.0292, sigma_max=14.6146, rho=7.0):
    ramp = np.linspace(0, 1, n, dtype=np.float64)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return ((hi + ramp * (lo - hi)) ** rho).astype(np.float32)


def flow_sigmas(n, shift=3.0):
    t = np.linspace(1.0, 0.0, n + 1, dtype=np.float64)[:-1]
    t = np.clip(shift * t / (1 + (shift - 1) * t), 0.0, 0.999)
    return t.astype(np.float32)


def workload_sigmas(workload):
    _shape, flow, n_sig, _n = WORKLOADS[workload]
    return flow_sigmas(n_sig) if flow else karras_sigmas(n_sig)


def times_from_sigma(s, flow):
    """(VE sigma, abt, flow t) from sigma the way KSamplerX0Inpaint forms them (nodes.py:242-252)."""
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


def mask_kind_of(shape, kind=None):
    return kind or ("temporal" if len(shape) == 5 else "box")


def make_mask(shape, kind=None):
    """latent_mask (1 = known).  box: 50 % box over the last axis; temporal: the leading latent frames
    known (video latents, C5); blob: a centred disc of inpainting covering ~38 % of every plane."""
    kind = mask_kind_of(shape, kind)
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


def mask_description(shape, kind=None):
    kind = mask_kind_of(shape, kind)
    if kind == "temporal":
        return f"temporal mask (latent frames >= {temporal_known_frames(shape[2]) if len(shape) == 5 else 0} inpainted)"
    return {"box": "50% box mask", "blob": "centred disc mask"}[kind]


def make_inputs(shape, flow, sigma0, seed, device, xp, mask_kind=None):
    """(x, y, noise, mask) of a job: y, noise ~ N(0, 1) from numpy's generator, x = the noised latent at sigma0."""
    g = np.random.default_rng(seed)
    y = g.standard_normal(shape, dtype=np.float32)
    noise = g.standard_normal(shape, dtype=np.float32)
    x = (sigma0 * noise + (1 - sigma0) * y) if flow else (y + noise * sigma0)
    mask = make_mask(shape, mask_kind)
    return tuple(xp(a.astype(np.float32)) for a in (x, y, noise, mask))


def attach_mask_format(mask, fmt):
    """The job set-up step that hands the kernels a compact copy of a binary mask ("f32": nothing attached -- the
    reference's interface; the engine packs such a mask by itself on the job's second sigma call)."""
    if fmt == "bits":
        import lanpaint_amd
        return lanpaint_amd.pack_mask(mask)
    if fmt == "u8":
        mask._lp_u8 = mask.to(torch.uint8).contiguous()
    return mask


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


class Job:
    """A workload's tensors on one device: inputs, the sigma / time tensors of every call, the Euler weights."""

    def __init__(self, workload, device, seed=0, mask_kind=None, mask_format="f32", rows=None):
        shape, self.flow, self.n_sig, self.n_think = WORKLOADS[workload]
        self.workload, self.shape, self.device, self.mask_kind = workload, tuple(shape), torch.device(device), mask_kind
        self.sig_np = workload_sigmas(workload)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)   # noqa: E731
        self.x0, self.y, self.noise, self.mask = make_inputs(self.shape, self.flow, float(self.sig_np[0]), seed, self.device, to, mask_kind)
        self.mask_format = mask_format
        self.mask = attach_mask_format(self.mask, mask_format) if self.device.type == "cuda" else self.mask
        b = self.shape[0]
        self.sig_list = [torch.full((b,), float(s), dtype=torch.float32, device=self.device) for s in self.sig_np]
        self.times_list = [times_from_sigma(s, self.flow) for s in self.sig_list]
        self.ratios = euler_ratios(self.sig_list, len(self.shape))
        self.n_el = int(np.prod(self.shape))

    def renoise(self):
        """x0 from the (possibly replaced) y and this job's noise."""
        s0 = float(self.sig_np[0])
        self.x0 = (s0 * self.noise + (1 - s0) * self.y) if self.flow else (self.y + self.noise * s0)

    def engine(self, model=None, **kw):
        """The engine for this job.  With no keyword: exactly what `LanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize,
        IS_FLUX, IS_FLOW)` builds (reference lanpaint.py:8) -- graph="auto", rng="torch"."""
        from lanpaint_amd import LanPaint
        h = HYPER
        model = model if model is not None else StubBackbone(self.flow)
        return LanPaint(model, self.n_think, h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], False, self.flow, **kw)

    def run(self, engine, n_sigmas=None):
        n = self.n_sig if n_sigmas is None else n_sigmas
        return schedule_pass(engine, self.x0, self.y, self.noise, self.mask, self.sig_list[:n], self.times_list[:n],
                             self.ratios[:max(0, n - 1)], self.n_think)

    def timed(self, engine, steps, warm=5):
        """(think-iterations/s, ms per step) of `steps` passes after `warm` untimed ones, device-synchronised either side."""
        import time
        for _ in range(warm):
            self.run(engine)
        torch.cuda.synchronize(self.device)
        it0, t0 = engine.iterations_run, time.perf_counter()
        for _ in range(steps):
            self.run(engine)
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        return (engine.iterations_run - it0) / dt, 1e3 * dt / steps
