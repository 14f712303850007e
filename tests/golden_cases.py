"""Case table for the golden fixtures (tests/golden/*.npz).

Each case is ONE engine call at one sigma (or a short sigma schedule) with all
inputs derived from a numpy seed, so the inputs can be rebuilt anywhere; the
.npz stores the inputs too, plus the reference's recorded xi stream and outputs.
`make_golden.py` (needs /root/reference) writes the fixtures; tests read them."""
from __future__ import annotations

import numpy as np

from oracle.lanpaint_oracle import times_from_sigma

HYPER_DEFAULT = dict(NSteps=5, Friction=15.0, Lambda=5.0, Beta=1.0, StepSize=0.2, MinStepFrac=0.0)


def box_mask(shape, frac=0.5):
    m = np.zeros(shape, dtype=np.float32)
    w = shape[-1]
    m[..., : int(w * frac)] = 1.0
    return m


def _inputs(seed, shape, sigma0):
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    sig = np.asarray(sigma0, dtype=np.float32).reshape((-1,) + (1,) * (len(shape) - 1))
    x = (y + noise * sig).astype(np.float32)
    return x, y, noise


def build_case(name):
    c = dict(CASES[name])
    shape = tuple(c["shape"])
    flow = bool(c.get("flow", False))
    sigma = np.asarray(c["sigma"], dtype=np.float32)
    x, y, noise = _inputs(c.get("seed", 0), shape, sigma)
    if flow:   # rectified-flow x_t = t*noise + (1-t)*y
        t = sigma.reshape((-1,) + (1,) * (len(shape) - 1))
        x = (t * noise + (1 - t) * y).astype(np.float32)
    if c.get("mask") == "soft":
        mask = np.random.default_rng(1234).random(shape, dtype=np.float32)
    elif c.get("mask") == "ones":
        mask = np.ones(shape, dtype=np.float32)
    elif c.get("mask") == "zeros":
        mask = np.zeros(shape, dtype=np.float32)
    elif c.get("mask") == "checker":
        idx = np.indices(shape).sum(axis=0)
        mask = (idx % 2).astype(np.float32)
    elif c.get("mask") == "temporal":          # video latent [B, C, F, H, W]: the leading frames stay known (BASELINE C5)
        mask = np.zeros(shape, dtype=np.float32)
        mask[:, :, : c.get("known_frames", shape[2] // 2)] = 1.0
    elif c.get("mask") == "blob":
        mask = np.ones(shape, dtype=np.float32)
        h, w = shape[-2], shape[-1]
        mask[..., h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 0.0
    else:
        mask = box_mask(shape)
    if c.get("zero_noise"):
        noise = np.zeros_like(noise)
    ve, abt, flow_t = times_from_sigma(sigma.astype(np.float32), flow)
    hyper = dict(HYPER_DEFAULT)
    hyper.update(c.get("hyper", {}))
    return dict(name=name, shape=shape, flow=flow, flux=bool(c.get("flux", False)), sigma=sigma,
                x=x, y=y, noise=noise, mask=mask,
                times=(ve.astype(np.float32), abt.astype(np.float32), flow_t.astype(np.float32)),
                hyper=hyper, model=c.get("model", "linear_tuple"), n_steps=c.get("n_steps", None),
                model_options=c.get("model_options", None), audio=c.get("audio", None), xi_seed=c.get("xi_seed", None),
                digest=bool(c.get("digest", False)))


DIGEST_SAMPLES = 4096


def digest(a, sample_seed):
    """Compact stand-in for a full-size output (a digest fixture stores this instead of the tensor): float64 sums and
    sums of squares per leading-axes slice (batch row; per frame as well for video latents) and DIGEST_SAMPLES elements at
    seeded positions.  An error anywhere in the tensor moves a slice sum; the samples pin individual values."""
    a = np.asarray(a)
    flat = a.reshape(-1)
    idx = np.random.default_rng(sample_seed).choice(flat.size, size=min(DIGEST_SAMPLES, flat.size), replace=False)
    idx.sort()
    a64 = a.astype(np.float64)
    if a.ndim == 5:       # [B, C, F, H, W] -> per (row, frame)
        sums = a64.sum(axis=(1, 3, 4))
        sq = (a64 ** 2).sum(axis=(1, 3, 4))
    else:
        sums = a64.reshape(a.shape[0], -1).sum(axis=1)
        sq = (a64.reshape(a.shape[0], -1) ** 2).sum(axis=1)
    return dict(sums=sums, sumsq=sq, samples=flat[idx].astype(np.float32), sample_idx=idx.astype(np.int64))


def seeded_xi(xi_seed, shape, n):
    """The xi stream of a compact fixture: n standard-normal draws of `shape` from numpy's PCG64 (portable)."""
    g = np.random.default_rng(xi_seed)
    return [g.standard_normal(shape, dtype=np.float32) for _ in range(n)]


CASES = {
    # VE (SD1.5 / SDXL notation)
    "ve_basic":        dict(shape=(1, 4, 8, 8), sigma=[2.0]),
    "ve_msf1":         dict(shape=(1, 4, 8, 8), sigma=[0.5], hyper=dict(MinStepFrac=1.0), seed=1),
    "ve_big_sigma":    dict(shape=(1, 4, 8, 8), sigma=[14.6146], seed=2),
    "ve_small_sigma":  dict(shape=(1, 4, 8, 8), sigma=[0.05], seed=3, hyper=dict(NSteps=3)),
    "ve_tiny_sigma":   dict(shape=(1, 4, 8, 8), sigma=[0.0292], seed=4),
    "ve_n0":           dict(shape=(1, 4, 8, 8), sigma=[1.0], n_steps=0, seed=5),
    "ve_n1":           dict(shape=(1, 4, 8, 8), sigma=[1.0], n_steps=1, seed=6),
    "ve_n2":           dict(shape=(1, 4, 8, 8), sigma=[1.0], n_steps=2, seed=7),
    "ve_single_out":   dict(shape=(1, 4, 8, 8), sigma=[1.5], model="denoiser_single", seed=8),
    "ve_list_one":     dict(shape=(1, 4, 8, 8), sigma=[1.5], model="list_one", seed=9),
    "ve_offset":       dict(shape=(2, 4, 6, 10), sigma=[3.0, 3.0], model="offset_tuple", seed=10),
    "ve_batch_rows":   dict(shape=(3, 4, 8, 8), sigma=[2.0, 0.7, 5.0], model="denoiser_single", seed=11),
    "ve_soft_mask":    dict(shape=(1, 4, 8, 8), sigma=[2.0], mask="soft", seed=12),
    "ve_all_known":    dict(shape=(1, 4, 8, 8), sigma=[2.0], mask="ones", seed=13),
    "ve_all_inpaint":  dict(shape=(1, 4, 8, 8), sigma=[2.0], mask="zeros", seed=14),
    "ve_checker":      dict(shape=(1, 4, 7, 9), sigma=[0.8], mask="checker", seed=15),
    "ve_odd_numel":    dict(shape=(1, 3, 5, 7), sigma=[1.2], seed=16),
    "ve_zero_noise":   dict(shape=(1, 4, 8, 8), sigma=[2.0], zero_noise=True, seed=17),
    "ve_lambda_beta":  dict(shape=(1, 4, 8, 8), sigma=[2.0], hyper=dict(Lambda=8.0, Beta=0.5, StepSize=0.15), seed=18),
    "ve_sdxl_shape":   dict(shape=(1, 4, 32, 32), sigma=[1.0], seed=19),
    # BASELINE shapes at FULL size, straight from the reference (no oracle in between): C1 = SD1.5 1x4x64x64, C2 = SDXL
    # 1x4x128x128 and C4 = Flux 1x16x64x64, one sigma call of 5 think iterations.  The xi stream comes from a numpy seed (the
    # reference's torch.randn_like is fed from it), so the fixture only has to store the two outputs.
    "ve_sd15_full":    dict(shape=(1, 4, 64, 64), sigma=[2.5], seed=35, xi_seed=4244),          # C1 = SD1.5 1x4x64x64
    "ve_sdxl_full":    dict(shape=(1, 4, 128, 128), sigma=[1.0], seed=33, xi_seed=4242),
    "flow_flux_full":  dict(shape=(1, 16, 64, 64), sigma=[0.6], flow=True, seed=34, xi_seed=4243),
    # C3 = SDXL batch of 4 (per-row sigma: the reference's row broadcast, lanpaint.py:23-29, and its flow-form replace step
    # for per-row sigma, :89-92) and C5 = Wan 1x16x21x60x104 video latent with the temporal mask, both at FULL size from the
    # reference; stored as digests (slice sums + 4 096 sampled elements) because the tensors are 1 MiB / 8 MiB each
    "ve_sdxl_b4_full": dict(shape=(4, 4, 128, 128), sigma=[2.0, 0.7, 5.0, 1.2], seed=36, xi_seed=4245, digest=True),
    "flow_wan_full":   dict(shape=(1, 16, 21, 60, 104), sigma=[0.6], flow=True, seed=37, xi_seed=4246, digest=True,
                            mask="temporal", known_frames=8),
    # flow / flux (Flux, Wan, SD3 notation)
    "flow_basic":      dict(shape=(1, 16, 4, 4), sigma=[0.7], flow=True, seed=20),
    "flow_flux_flag":  dict(shape=(1, 16, 4, 4), sigma=[0.5], flux=True, seed=21),
    "flow_low_t":      dict(shape=(1, 16, 4, 4), sigma=[0.15], flow=True, seed=22),
    "flow_high_t":     dict(shape=(1, 16, 4, 4), sigma=[0.97], flow=True, seed=23),
    "flow_batch":      dict(shape=(2, 16, 4, 4), sigma=[0.9, 0.3], flow=True, seed=24),
    "flow_video5d":    dict(shape=(1, 4, 3, 4, 6), sigma=[0.6], flow=True, seed=25),
    "flow_msf1":       dict(shape=(1, 16, 4, 4), sigma=[0.1], flow=True, hyper=dict(MinStepFrac=1.0), seed=26),
    "flow_beta":       dict(shape=(1, 16, 4, 4), sigma=[0.6], flow=True, hyper=dict(Beta=0.5, Lambda=8.0), seed=27),
    # inner early stop (earlystop.py), default metric
    "ve_earlystop":    dict(shape=(1, 4, 8, 8), sigma=[1.0], mask="blob", seed=28, hyper=dict(NSteps=10),
                            model_options={"lanpaint_semantic_stop": {"threshold": 0.3, "patience": 1}}),
    "ve_earlystop_fast": dict(shape=(1, 4, 8, 8), sigma=[1.0], mask="blob", seed=32, hyper=dict(NSteps=10),
                            model_options={"lanpaint_semantic_stop": {"threshold": 5.0, "patience": 2}}),
    "ve_earlystop_run_all": dict(shape=(1, 4, 8, 8), sigma=[1.0], mask="blob", seed=29, hyper=dict(NSteps=6),
                            model_options={"lanpaint_semantic_stop": {"threshold": 1e-9, "patience": 2}}),
    # MiniMax-H3 style flat AV pack: per-element times (general path)
    "av_flat_pack":    dict(shape=(1, 1, 24), sigma=[0.5], flow=True, seed=30, mask="soft",
                            audio=dict(split=15, flow_a=0.2, corr=0.625)),
    "av_flat_binary":  dict(shape=(1, 2, 16), sigma=[0.6], flow=True, seed=31,
                            audio=dict(split=10, flow_a=0.35, corr=0.8)),
}

# short sigma schedules driven by a host-side Euler sampler (k-diffusion sample_euler form)
SCHEDULES = {
    "sched_ve_karras6": dict(shape=(1, 4, 8, 8), flow=False, n_sigmas=6, sigma_max=14.6146, sigma_min=0.0292,
                             hyper=dict(NSteps=3), seed=40),
    "sched_flow5":      dict(shape=(1, 16, 4, 4), flow=True, n_sigmas=5, hyper=dict(NSteps=2), seed=41),
    "sched_ve_batch2":  dict(shape=(2, 4, 8, 8), flow=False, n_sigmas=4, sigma_max=10.0, sigma_min=0.1,
                             hyper=dict(NSteps=2, MinStepFrac=1.0), seed=42),
    # the inner early stop (earlystop.py) over a schedule: a fresh stopper per sigma call, the abt-scaled threshold moving with
    # sigma -- the noisy head of the schedule runs all iterations, the middle stops early, the tail's threshold scales to ~0
    "sched_ve_earlystop8": dict(shape=(1, 4, 12, 12), flow=False, n_sigmas=8, sigma_max=8.0, sigma_min=0.05, mask="blob",
                                hyper=dict(NSteps=8), seed=43,
                                model_options={"lanpaint_semantic_stop": {"threshold": 6.0, "patience": 1}}),
}


# the sampler-facing callable (reference nodes.py:221-315) over a whole schedule with the NODE defaults: MinStepFrac = 1.0,
# LanPaint_EarlyStop = 1 -> n_eff = round(NSteps (1 - abt)) ramping 5 ... 0 (nodes.py:286-299)
NODE_SCHEDULES = {
    "node_ve_karras12": dict(shape=(2, 4, 16, 16), flow=False, n_sigmas=12, sigma_max=14.6146, sigma_min=0.0292, seed=50,
                             xi_seed=4250),
    "node_flow12":      dict(shape=(1, 16, 8, 8), flow=True, n_sigmas=12, seed=51, xi_seed=4251),
    # a 5-D video latent (BASELINE configs[4] in small) through the reference's sampler callable
    "node_flow_video5d": dict(shape=(1, 16, 3, 8, 10), flow=True, n_sigmas=12, seed=52, xi_seed=4252),
}
NODE_DEFAULTS = dict(NSteps=5, Friction=15.0, Lambda=5.0, Beta=1.0, StepSize=0.2, MinStepFrac=1.0, EarlyStop=1)


def build_node_schedule(name):
    c = dict(NODE_SCHEDULES[name])
    shape, flow = tuple(c["shape"]), c["flow"]
    sig = flow_sigmas(c["n_sigmas"]) if flow else karras_sigmas(c["n_sigmas"], c["sigma_min"], c["sigma_max"])
    rng = np.random.default_rng(c["seed"])
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    x = (sig[0] * noise + (1 - sig[0]) * y).astype(np.float32) if flow else (y + noise * sig[0]).astype(np.float32)
    # ComfyUI's denoise_mask (1 = inpaint), NOT binary on purpose: the callable thresholds it at 0.5 (nodes.py:281)
    denoise_mask = (rng.random(shape) > 0.45).astype(np.float32) * np.float32(0.9) + np.float32(0.05)
    return dict(name=name, shape=shape, flow=flow, sigmas=sig, x=x, y=y, noise=noise, denoise_mask=denoise_mask,
                hyper=dict(NODE_DEFAULTS), xi_seed=c["xi_seed"])


# BASELINE.json's configurations as WHOLE schedules from the reference engine (engine driven directly, Euler update between
# sigmas, stub backbone, 50 % box mask, engine defaults): C1 = SD1.5 20 sigmas x 5, C2 = SDXL 30 x 5 (the headline config),
# C4 = Flux 28 x 10.  Stored as digests of every FULL_SCHEDULE_STRIDE-th denoised and of the final x.
FULL_SCHEDULES = {
    "full_c1_sd15":  dict(shape=(1, 4, 64, 64), flow=False, n_sigmas=20, n_think=5, seed=60, xi_seed=4260),
    "full_c2_sdxl":  dict(shape=(1, 4, 128, 128), flow=False, n_sigmas=30, n_think=5, seed=61, xi_seed=4261),
    "full_c4_flux":  dict(shape=(1, 16, 64, 64), flow=True, n_sigmas=28, n_think=10, seed=62, xi_seed=4262),
    # C3 = SDXL batch of 4 rows (BASELINE configs[2], the per-GPU share of batch 32): every row walks its OWN sigma ramp
    # (row r sits at ROW_SCALE[r] x the Karras schedule), so the whole schedule runs through the reference's per-row
    # broadcast (lanpaint.py:23-33) and its flow-form replace step for per-row sigma (:89-92)
    "full_c3_sdxl_b4": dict(shape=(4, 4, 128, 128), flow=False, n_sigmas=30, n_think=5, seed=63, xi_seed=4263,
                            row_scale=(1.0, 0.9, 0.8, 0.7)),
    # C5 = Wan 1x16x21x60x104 video latent (configs[4]); the latent mask is the REFERENCE's reshape_mask(video_inpainting=True)
    # (nodes.py:59-133) of an 81-frame 480x832 pixel mask (video_pixel_mask below), kept in the fixture as bits
    "full_c5_wan":   dict(shape=(1, 16, 21, 60, 104), flow=True, n_sigmas=30, n_think=5, seed=64, xi_seed=4264, video_mask=True),
}
FULL_SCHEDULE_STRIDE = 5


def video_pixel_mask(latent_shape):
    """ComfyUI denoise mask (1 = inpaint) of the C5 job at PIXEL resolution, [F, H, W] = [4 (f - 1) + 1, 8 h, 8 w]: the second
    half of the video is inpainted whole; in the first half a rectangle is inpainted over a run of frames and
    a thin stroke on ONE frame -- spatial nearest-exact picks (480 -> 60, 832 -> 104), frames the 81 -> 21 resample skips, and
    the 5-tap temporal union all matter for the result."""
    f, h, w = 4 * (latent_shape[2] - 1) + 1, 8 * latent_shape[3], 8 * latent_shape[4]
    m = np.zeros((f, h, w), dtype=np.float32)
    m[f // 2:] = 1.0
    m[9:27, 117:363, 203:601] = 1.0           # edges off the 8-pixel grid on purpose
    m[5, 40:44, 100:700] = 1.0                # a 4-pixel stroke on a single frame: survives only if a latent row picks it
    return m


def build_full_schedule(name):
    c = dict(FULL_SCHEDULES[name])
    shape, flow = tuple(c["shape"]), c["flow"]
    sig = (flow_sigmas(c["n_sigmas"]) if flow else karras_sigmas(c["n_sigmas"]))[:-1]        # the bench's schedule: no trailing 0
    rng = np.random.default_rng(c["seed"])
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    x = (sig[0] * noise + (1 - sig[0]) * y).astype(np.float32) if flow else (y + noise * sig[0]).astype(np.float32)
    hyper = dict(HYPER_DEFAULT)
    hyper["NSteps"] = c["n_think"]
    row_scale = np.asarray(c.get("row_scale", (1.0,) * shape[0]), dtype=np.float32)
    if "row_scale" in c:                      # x_t of row r starts at ITS sigma
        x = (y + noise * (sig[0] * row_scale).reshape((-1,) + (1,) * (len(shape) - 1))).astype(np.float32)
    # (video_mask: the latent mask comes from reshape_mask -- the reference's when the fixture is made, ours in the GPU test;
    # `mask` is then filled in by the caller)
    return dict(name=name, shape=shape, flow=flow, sigmas=sig, x=x, y=y, noise=noise,
                mask=None if c.get("video_mask") else box_mask(shape), hyper=hyper, xi_seed=c["xi_seed"],
                n_draws=len(sig) * (2 * c["n_think"] - 1), row_scale=row_scale, video_mask=bool(c.get("video_mask")))


def seeded_xi_stream(xi_seed, shape):
    """Endless generator form of `seeded_xi` (a full schedule draws hundreds of tensors: not worth holding at once)."""
    g = np.random.default_rng(xi_seed)
    while True:
        yield g.standard_normal(shape, dtype=np.float32)


def karras_sigmas(n, sigma_min=0.0292, sigma_max=14.6146, rho=7.0):
    ramp = np.linspace(0, 1, n, dtype=np.float64)
    min_inv, max_inv = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (max_inv + ramp * (min_inv - max_inv)) ** rho
    return np.concatenate([s, [0.0]]).astype(np.float32)


def flow_sigmas(n, shift=3.0):
    t = np.linspace(1.0, 0.0, n + 1, dtype=np.float64)[:-1]
    t = shift * t / (1 + (shift - 1) * t)
    t = np.clip(t, 0.0, 0.999)          # t == 1 makes VE_sigma infinite (nodes.py:245)
    return np.concatenate([t, [0.0]]).astype(np.float32)


def build_schedule(name):
    c = dict(SCHEDULES[name])
    shape = tuple(c["shape"])
    flow = c["flow"]
    sig = flow_sigmas(c["n_sigmas"]) if flow else karras_sigmas(c["n_sigmas"], c["sigma_min"], c["sigma_max"])
    rng = np.random.default_rng(c["seed"])
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    if flow:
        x = (sig[0] * noise + (1 - sig[0]) * y).astype(np.float32)
    else:
        x = (y + noise * sig[0]).astype(np.float32)
    hyper = dict(HYPER_DEFAULT)
    hyper.update(c.get("hyper", {}))
    mask = box_mask(shape)
    if c.get("mask") == "blob":
        mask = np.ones(shape, dtype=np.float32)
        h, w = shape[-2], shape[-1]
        mask[..., h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 0.0
    return dict(name=name, shape=shape, flow=flow, sigmas=sig, x=x, y=y, noise=noise, mask=mask,
                hyper=hyper, model="linear_tuple", model_options=c.get("model_options"))
