"""ComfyUI-facing layer of the MI355X Langevin path: the drop-in boundary.

Mirrors the public surface of the reference's src/LanPaint/nodes.py for THIS path
(file:line below are in /root/reference/src/LanPaint/nodes.py):
    reshape_mask / prepare_mask            :59-133,159-160  -> HIP kernel lp_reshape_mask (torch's nearest-exact index, bit for bit)
    min_step_frac_effective_steps          :134-144
    _sanitize_param                        :146-157
    sampling_function_LanPaint             :161-175   dual-CFG model function -> (x0, x0_BIG)
    CFGGuider_LanPaint                     :178-216
    KSamplerX0Inpaint                      :221-315   the `model(x, sigma, **extra_args)` callable samplers invoke
    KSAMPLER.sample                        :318-379
    override_sample_function               :384-421   monkey-patch set, restored in `finally`, re-entrancy guarded
    LanPaint_KSampler / _KSamplerAdvanced / _SamplerCustom / _SamplerCustomAdvanced   :452-808
Only host glue lives here; every per-element operation is a kernel behind include/lanpaint_hip.h.
ComfyUI itself is not vendored: the module imports without it (tests stub it exactly as the
reference's tests do) and the sampler nodes raise a clear error when it is absent.
Image/mask utility nodes, the AV encode/decode nodes and the video-mask editor are out of scope
(pixel-space / editor-time, not on the sampling loop).
"""
from __future__ import annotations

import math
import os
import weakref
from contextlib import contextmanager

import torch

from . import _cabi, interp_rule
from .blend import MaskBlend, gaussian_kernel_2d, merge_video_with_mask  # noqa: F401
from .image_nodes import LanPaint_ImageDecode, LanPaint_ImageEncode, _snap_mask_nearest_exact  # noqa: F401
from .resample import _hip_device, _resample  # noqa: F401
from .lanpaint import LanPaint, pack_mask, raw_stream, refresh_packed_mask, tensor_version
from .types import FusedCFGHeads

try:                                    # ComfyUI present (or stubbed by tests)
    import comfy                        # type: ignore
    import comfy.samplers               # type: ignore
    from comfy.model_base import ModelType  # type: ignore
    HAVE_COMFY = True
except Exception:                       # standalone use of the engine / mask kernels
    comfy = None
    HAVE_COMFY = False

    class ModelType:                    # placeholders so `model_type` comparisons still work
        FLUX = "FLUX"
        FLOW = "FLOW"
        FLOW_AV = "FLOW_AV"

try:
    import comfyui_version              # type: ignore
    _ver = tuple(int(p) if p.isdigit() else 0 for p in comfyui_version.__version__.split("."))
except Exception:
    _ver = (9999,)
COMFYUI_VERSION_060_OR_NEWER = _ver >= (0, 6, 0)

try:
    from comfy.ldm.minimax.model import time_shift_sigma, time_shift_slope  # type: ignore
except Exception:
    time_shift_sigma = None
    time_shift_slope = None

FLOW_MODEL_TYPES = (getattr(ModelType, "FLOW", "FLOW"), getattr(ModelType, "FLOW_AV", None))
FUSE_CFG = bool(int(__import__("os").environ.get("LANPAINT_AMD_FUSE_CFG", "1")))   # 0: always run cfg_function eagerly


def _require_comfy(what):
    if not HAVE_COMFY:
        raise RuntimeError(f"{what} needs ComfyUI (the `comfy` package) on sys.path")


# =====================================================================================
# mask preparation (nodes.py:59-160)
# =====================================================================================
def reshape_mask(input_mask, output_shape, video_inpainting=False, device=None):
    """nodes.py:59-133.  Nearest-exact resample to the latent grid, 5-wide temporal union for
    video, channel / batch broadcast -- computed by one HIP launch.  The source index follows the rule of the torch kernel
    the REFERENCE would have run for this mask: it resamples on the mask's own device before `.to(device)` (nodes.py:159-160),
    so a host-resident mask (ComfyUI's) gets ATen's CPU rules, a device-resident one the GPU kernels' rule
    (interp_rule.rule_for; all three forms are fp32 expressions reproduced op for op, e.g. the scalar one
    src = min(int(floorf((i + 0.5f) * (float(in) / float(out)))), in-1)).  Bit-equal to F.interpolate on that device.
    Returns a float mask of `output_shape` on the HIP device the work ran on."""
    output_shape = tuple(int(s) for s in output_shape)
    dev = _hip_device(input_mask, device)
    m = input_mask.to(device=dev, dtype=torch.float32)
    nd_out = len(output_shape)
    rule = None                      # of the reference's FIRST interpolate call where it makes two (the second is an identity)

    # ---- bring the mask to [B', C', (F), H, W] exactly as the reference's unsqueeze rules do
    if video_inpainting:                                                    # :64-73
        if m.ndim == 3:
            m = m[None, None]
        elif m.ndim == 4:
            m = m.permute(1, 0, 2, 3)[None]
        elif m.ndim == 2:
            m = m[None, None, None]
    elif m.ndim == 1 and nd_out == 4:                                      # :74-83 audio [F] -> tokens
        rule = interp_rule.rule_for(input_mask, m.reshape(1, 1, -1), (output_shape[-1],))       # a 1-D interpolate call
        m = m.reshape(1, 1, 1, m.shape[0])          # rows of the (ch, T) layout all read the same 1-D mask
    elif m.ndim == 4 and nd_out == 4 and m.shape[1] == 1 and m.shape[3] == 1:   # :84-89 audio [1,1,F,1]
        rule = interp_rule.rule_for(input_mask, m, (output_shape[-1], 1))                       # 2-D call, size (T, 1)
        m = m.permute(0, 1, 3, 2)
    elif m.ndim == 2:                                                       # :90-91
        m = m[None, None]
    elif m.ndim == 3:                                                       # :92-93
        m = m[:, None]
    if nd_out == 5 and m.ndim == 4 and COMFYUI_VERSION_060_OR_NEWER:        # :96-98
        m = m[:, :, None]

    if video_inpainting:                                                    # :100-122
        if m.ndim != 5:
            raise ValueError(f"video mask must resolve to 5 dims, got shape {tuple(m.shape)}")
        rule = interp_rule.rule_for(input_mask, m, output_shape[2:])
        m = m.contiguous()
        c_out = output_shape[1] if m.shape[1] < output_shape[1] else m.shape[1]
        return _resample(m, output_shape[0], c_out, output_shape[2], output_shape[3], output_shape[4], 5, rule)

    if nd_out == 5 and m.ndim == 4:
        # ComfyUI < 0.6.0 (:124-125): the 4-D mask is resampled on (H, W) only and then hits
        # `repeat((1, C, 1, 1, 1))` on a 4-D tensor, which PREPENDS a dim.  Resample with the
        # kernel, reproduce the repeat/slice quirk with views.
        b4, c4, _, _ = m.shape
        rule = interp_rule.rule_for(input_mask, m, output_shape[-2:])
        r = _resample(m.contiguous()[:, :, None], b4, c4, 1, output_shape[-2], output_shape[-1], 1, rule)[:, :, 0]
        if r.shape[1] < output_shape[1]:
            r = r.repeat((1, output_shape[1], 1, 1, 1))[:, :output_shape[1]]
        return _repeat_to_batch_size(r, output_shape[0])
    if m.ndim != nd_out:
        raise ValueError(f"mask of shape {tuple(input_mask.shape)} does not fit latent shape {output_shape}")
    if nd_out == 4:
        if rule is None:
            rule = interp_rule.rule_for(input_mask, m, output_shape[2:])
        src5 = m.contiguous()[:, :, None]
        c_out = output_shape[1] if m.shape[1] < output_shape[1] else m.shape[1]
        return _resample(src5, output_shape[0], c_out, 1, output_shape[2], output_shape[3], 1, rule)[:, :, 0]
    if nd_out == 5:
        rule = interp_rule.rule_for(input_mask, m, output_shape[2:])
        c_out = output_shape[1] if m.shape[1] < output_shape[1] else m.shape[1]
        return _resample(m.contiguous(), output_shape[0], c_out, output_shape[2], output_shape[3], output_shape[4], 1, rule)
    raise ValueError(f"unsupported latent rank {nd_out}")


def _repeat_to_batch_size(t, batch_size):
    """comfy.utils.repeat_to_batch_size semantics (narrow when larger, tile + narrow when smaller)."""
    if t.shape[0] > batch_size:
        return t[:batch_size]
    if t.shape[0] < batch_size:
        reps = math.ceil(batch_size / t.shape[0])
        return t.repeat((reps,) + (1,) * (t.ndim - 1))[:batch_size]
    return t


def prepare_mask(noise_mask, shape, device, video_inpainting=False):
    """nodes.py:159-160."""
    return reshape_mask(noise_mask, shape, video_inpainting, device=device).to(device)


def min_step_frac_effective_steps(n_steps, frac, min_frac):
    """nodes.py:134-144: inner-step count under the MinStepFrac tail ramp (Python round = banker's)."""
    if min_frac <= 0 or frac >= min_frac or n_steps <= 0:
        return n_steps
    return max(0, round(n_steps * frac / min_frac))


def _sanitize_param(value, default, allowed=None):
    """nodes.py:146-157: coerce a stale / hand-edited widget value to its default."""
    if allowed is not None:
        return value if value in allowed else default
    if isinstance(value, bool) or not isinstance(value, (int, float)):
        return default
    return value


def _detect_minimax_h3_audio(model_patcher, model_options, latent_shapes):
    """nodes.py:34-52: (latent_shapes, shift_video, shift_audio) for a MiniMax-H3 AV pack, else None."""
    if latent_shapes is None or len(latent_shapes) < 2:
        return None
    diff_model = getattr(getattr(model_patcher, "model", None), "diffusion_model", None)
    shift_v = getattr(diff_model, "sigma_shift_video", None)
    shift_a = getattr(diff_model, "sigma_shift_audio", None)
    if shift_v is None or shift_a is None:
        return None
    topts = model_options.get("transformer_options", {}) if isinstance(model_options, dict) else {}
    return (latent_shapes, float(topts.get("minimax_h3_sigma_shift_video", shift_v)),
            float(topts.get("minimax_h3_sigma_shift_audio", shift_a)))


# =====================================================================================
# dual-CFG model function + guider patch (nodes.py:161-216)
# =====================================================================================
def sampling_function_LanPaint(model, x, timestep, uncond, cond, cond_scale, cond_scale_BIG, model_options={}, seed=None):
    """One batched cond/uncond backbone pass, two CFG combinations -> (x0, x0_BIG)  (nodes.py:161-175)."""
    _require_comfy("sampling_function_LanPaint")
    skip_uncond = math.isclose(cond_scale, 1.0) and not model_options.get("disable_cfg1_optimization", False)
    uncond_ = None if skip_uncond else uncond
    conds = [cond, uncond_]
    out = comfy.samplers.calc_cond_batch(model, conds, x, timestep, model_options)
    for fn in model_options.get("sampler_pre_cfg_function", []):
        out = fn({"conds": conds, "conds_out": out, "cond_scale": cond_scale, "timestep": timestep, "input": x,
                  "sigma": timestep, "model": model, "model_options": model_options})
    if FUSE_CFG and "sampler_cfg_function" not in model_options and not model_options.get("sampler_post_cfg_function"):
        # stock cfg_function is `uncond + (cond - uncond) * scale`: let the step kernel form both heads
        # from the one batched pass instead of 2 x 3 eager elementwise launches
        return FusedCFGHeads(out[0], out[1], cond_scale, cond_scale_BIG)
    cfg = comfy.samplers.cfg_function
    return (cfg(model, out[0], out[1], cond_scale, x, timestep, model_options=model_options, cond=cond, uncond=uncond_),
            cfg(model, out[0], out[1], cond_scale_BIG, x, timestep, model_options=model_options, cond=cond, uncond=uncond_))


class CFGGuider_LanPaint:
    """Methods grafted onto comfy.samplers.CFGGuider while the override is active (nodes.py:178-216)."""

    def outer_sample(self, noise, latent_image, sampler, sigmas, denoise_mask=None, callback=None, disable_pbar=False,
                     seed=None, **kwargs):
        self.inner_model, self.conds, self.loaded_models = comfy.sampler_helpers.prepare_sampling(
            self.model_patcher, noise.shape, self.conds, self.model_options)
        device = self.model_patcher.load_device
        wan22 = getattr(comfy.model_base, "WAN22", None)
        if wan22 is not None and isinstance(self.inner_model, wan22):
            self.inner_model.extra_conds = super(wan22, self.inner_model).extra_conds
        self.minimax_h3_audio = _detect_minimax_h3_audio(self.model_patcher, self.model_options,
                                                         kwargs.get("latent_shapes", None))
        if denoise_mask is not None and tuple(denoise_mask.shape) != tuple(noise.shape):
            denoise_mask = prepare_mask(denoise_mask, noise.shape, device,
                                        self.model_options.get("video_inpainting", False))
        noise, latent_image, sigmas = noise.to(device), latent_image.to(device), sigmas.to(device)
        comfy.samplers.cast_to_load_options(self.model_options, device=device, dtype=self.model_patcher.model_dtype())
        try:
            self.model_patcher.pre_run()
            output = self.inner_sample(noise, latent_image, device, sampler, sigmas, denoise_mask, callback,
                                       disable_pbar, seed, **kwargs)
        finally:
            self.model_patcher.cleanup()
        comfy.sampler_helpers.cleanup_models(self.conds, self.loaded_models)
        del self.inner_model
        del self.loaded_models
        return output

    def predict_noise(self, x, timestep, model_options={}, seed=None):
        return sampling_function_LanPaint(self.inner_model, x, timestep, self.conds.get("negative", None),
                                          self.conds.get("positive", None), self.cfg, self.cfg_BIG,
                                          model_options=model_options, seed=seed)


# =====================================================================================
# the sampler-facing callable (nodes.py:221-315)
# =====================================================================================
class KSamplerX0Inpaint:
    """`model(x, sigma, denoise_mask=, model_options=, seed=)` as k-diffusion sampler functions call it.
    Converts sigma to (VE_sigma, abt, flow_t), binarises / inverts the mask, picks the effective inner
    step count and hands the rest to the HIP engine (`self.PaintMethod`)."""

    def __init__(self, model, sigmas):
        self.inner_model = model
        self.sigmas = sigmas
        self.audio_indicator = None
        self.audio_shifts = None
        self._mask_cache = None          # (weakref(denoise_mask), version, latent_mask): binarised once per run
        self._mailbox = None             # pinned host float32[4]: lp_sigma_times writes {step index, mean(1-abt), seq}
        self._seq = 0
        self._node_desc = None           # LpNodeCallDesc of the one-call steady state (lp_node_call)
        self._node_static = None         # the per-run constants last written into it
        self._speculate = os.environ.get("LANPAINT_AMD_SPECULATE", "1") != "0"
        self._last_step = None           # schedule position of the previous call (from the device), for the next guess
        self._spec_misses = 0            # 2 = speculation is off for the run (two wrong guesses among the last four)
        self._spec_hist = []             # outcomes of the last four guesses
        self._n_eff_table = None         # (key, inner-step count per schedule position) from a host mirror of the schedule
        self._sched = None               # (schedule tensor object, dense copy, data_ptr, numel, len - 1): looked at once per run
        self._times = None               # (rows, device, [two sets of (VE sigma, abt, flow t, buffer)]): the three time
                                         # tensors of a call are views of one buffer, made once (a view costs the host
                                         # ~1.5 us) and used by alternate calls

    def _guess_inner_steps(self, nd, rows, flow):
        """The count the NEXT sigma is expected to get, or -1: samplers walk the schedule, so after the call at schedule
        position p the next one is usually at p + 1.  The count for each position comes from a host mirror of the schedule
        (one device->host copy per run) pushed through the same fp32 arithmetic as lp_sigma_times and the same rule.  A
        wrong guess is only slower, never wrong: the device voids the speculated run (lp_node_call); two misses among four
        guesses (a sampler that evaluates the model several times per step) turn guessing off for the run."""
        if self._last_step is None or self._spec_misses >= 2 or not self._speculate:
            return -1
        tab = self._n_eff_table
        key = (rows, flow, nd.n_steps, nd.early_stop, nd.min_step_frac)
        if tab is None or tab[0] != key:
            import numpy as np
            sched = self._sched[1].detach().cpu().numpy().astype(np.float32)            # the one host copy of the run
            one = np.float32(1.0)
            counts = []
            for j, s in enumerate(sched):
                if flow:
                    a = one - s
                    abt = (a * a) / (a * a + s * s) if (a * a + s * s) != 0 else np.float32(np.nan)
                else:
                    abt = one / (one + s * s)
                acc = np.float32(0.0)
                for _ in range(rows):                       # the kernel's sequential fp32 mean over (equal) rows
                    acc = np.float32(acc + np.float32(one - abt))
                frac = float(np.float32(acc / np.float32(rows)))
                counts.append(-1 if frac != frac else
                              _cabi.load().lp_effective_inner_steps(nd.n_steps, float(j), frac, nd.total_steps, nd.early_stop,
                                                                    nd.min_step_frac))
            self._n_eff_table = tab = (key, counts)
        j = self._last_step + 1
        return tab[1][j] if 0 <= j < len(tab[1]) else -1

    def _mailbox_views(self):
        """The pinned host words the device writes the two scalars of the inner-step rule into (fine-grained host
        memory: visible to a polling host thread right after the kernel's system-scope release)."""
        if self._mailbox is None:
            mb = torch.zeros(4, dtype=torch.float32).pin_memory()
            self._mailbox = (mb, mb.numpy(), mb.view(torch.int32).numpy())
        return self._mailbox

    def _wait_mailbox(self, seq, device):
        """Spin on the sequence word (the kernel needs a few us once the GPU reaches it); with a long backlog in
        front of it -- a real backbone -- stop burning the core and block on the stream instead."""
        _mb, f32, i32 = self._mailbox
        for _ in range(20000):
            if i32[2] == seq:
                break
        else:
            torch.cuda.current_stream(device).synchronize()
            if i32[2] != seq:
                raise RuntimeError("lp_sigma_times mailbox was not written (sequence %d, found %d)" % (seq, int(i32[2])))
        return float(f32[0]), float(f32[1])

    def _latent_mask(self, denoise_mask):
        """nodes.py:281-283, computed once per mask tensor OBJECT + version (weak reference: a
        denoise_mask_function may hand back a fresh tensor at a recycled address every step)."""
        c = self._mask_cache
        ver = (tensor_version(denoise_mask), denoise_mask.data_ptr())
        if (c is not None and c[0]() is denoise_mask and c[1][1] == ver[1] and denoise_mask.is_cuda
                and (c[1][0] != ver[0] or ver[0] == -1) and getattr(c[2], "_lp_bits", None) is not None):
            # the same tensor object, possibly rewritten in place: its version counter moved, or it has none (ComfyUI runs its
            # nodes under torch.inference_mode()).  The reference recomputes the mask on every call (nodes.py:277-283); here
            # ONE launch re-derives both forms -- bits and fp32 -- into the buffers of the cached latent mask, so the object,
            # its addresses and every capture made against them stay valid.
            refresh_packed_mask(c[2], denoise_mask)
            self._mask_cache = c = (c[0], ver, c[2])
            return c[2]
        if c is None or c[0]() is not denoise_mask or c[1] != ver:
            if denoise_mask.is_cuda:
                # binary by construction: the think loop streams 1 bit / element for it (one ballot launch)
                latent_mask = pack_mask(denoise_mask, denoise_mask=True)
            else:
                latent_mask = 1 - (denoise_mask > 0.5).float()
            self._mask_cache = c = (weakref.ref(denoise_mask), ver, latent_mask)
        return c[2]

    def __call__(self, x, sigma, denoise_mask, model_options={}, seed=None, **kwargs):
        model_type = self.inner_model.inner_model.model_type
        IS_FLUX = model_type == ModelType.FLUX
        IS_FLOW = model_type in FLOW_MODEL_TYPES
        fused_seq = None
        if (sigma.is_cuda and sigma.dtype == torch.float32 and sigma.ndim == 1 and self.sigmas.is_cuda
                and self.sigmas.dtype == torch.float32 and self.sigmas.device == sigma.device):
            # one launch: the three time tensors AND the two scalars of the inner-step rule, the latter straight into
            # pinned host memory (lp_sigma_times_mailbox) -- no blocking device->host copy
            sc = self._sched
            if sc is None or sc[0] is not self.sigmas:
                dense = self.sigmas.contiguous()
                self._sched = sc = (self.sigmas, dense, dense.data_ptr(), dense.numel(), len(self.sigmas) - 1)
            sig_c = sigma if sigma.is_contiguous() else sigma.contiguous()
            rows = sig_c.shape[0]
            tm = self._times
            if tm is None or tm[0] != rows or tm[1] != sigma.device:
                # Two sets, used by alternate calls: a call's launches read its times in stream order long before the
                # call after the next overwrites them, and a backbone that keeps the previous call's `t` to compare it
                # with the current one still sees two different tensors.
                sets = []
                for _ in range(2):
                    b = torch.empty((3 * rows,), dtype=torch.float32, device=sigma.device)
                    sets.append((b[:rows], b[rows:2 * rows], b[2 * rows:3 * rows], b))
                self._times = tm = (rows, sigma.device, sets)
            VE_Sigma, abt, Flow_t, buf = tm[2][self._seq & 1]
            mb = self._mailbox_views()[0]
            self._seq = fused_seq = (self._seq % 0x7ffffff0) + 1
            pm = getattr(self, "PaintMethod", None)
            if (denoise_mask is not None and self.audio_indicator is None and "denoise_mask_function" not in model_options
                    and getattr(pm, "_last_cap", None) is not None and sigma.device.index == torch.cuda.current_device()
                    and type(self.LanPaint_early_stop) is int and type(pm.n_steps) is int):
                # Steady state of a replayed run: sigma -> times, the replace launch, the wait for the device's answer, the
                # inner-step rule and the launch of the graph for that count are ONE call into the library (lp_node_call)
                nd = self._node_desc
                if nd is None:
                    nd = self._node_desc = _cabi.LpNodeCallDesc()
                    nd.spin_limit = 200000
                    nd.fold_sigma = int(os.environ.get("LANPAINT_AMD_FOLD_SIGMA", "1") != "0")
                # (what does not change from call to call within a run is written once)
                static = (rows, sc[2], sc[3], bool(IS_FLUX or IS_FLOW), pm.n_steps, self.LanPaint_early_stop, sc[4],
                          getattr(self, "LanPaint_min_step_frac", 1.0), mb)
                if self._node_static != static:
                    self._node_static = static
                    nd.rows, nd.schedule, nd.schedule_len, nd.is_flow = rows, sc[2], sc[3], int(static[3])
                    nd.scalars_out, nd.seq_out = mb.data_ptr(), mb.data_ptr() + 8
                    nd.n_steps, nd.early_stop, nd.total_steps = int(pm.n_steps), int(self.LanPaint_early_stop), sc[4]
                    nd.min_step_frac = float(static[7])
                nd.sigma, nd.seq, nd.times_out = sig_c.data_ptr(), fused_seq, buf.data_ptr()
                nd.guess = self._guess_inner_steps(nd, rows, static[3])
                res = pm.node_call(x, self.latent_image, self.noise, sigma, self._latent_mask(denoise_mask),
                                   (VE_Sigma, abt, Flow_t), model_options, seed, nd)
                if res is not None:
                    out = res[0]
                    self._last_step = int(nd.step_f)
                    if nd.speculated:                 # two misses among the last four guesses turn guessing off for the run
                        self._spec_hist = (self._spec_hist + [bool(nd.hit)])[-4:]
                        if self._spec_hist.count(False) >= 2:
                            self._spec_misses = 2
                    step_i = model_options.get("i", kwargs.get("i", 0))          # preview hook, nodes.py:304-313
                    if step_i % 2 == 0:
                        cb = model_options.get("callback", None)
                        if cb is not None:
                            cb({"i": step_i, "denoised": out, "x": x})
                    return out
            args = (sig_c.data_ptr(), rows, sc[2], sc[3], int(bool(IS_FLUX or IS_FLOW)), buf.data_ptr(),
                    mb.data_ptr(), mb.data_ptr() + 8, fused_seq, raw_stream(sigma.device))
            if sigma.device.index == torch.cuda.current_device():     # (the context manager costs the host ~2 us)
                _cabi.check(_cabi.load().lp_sigma_times_mailbox(*args), "lp_sigma_times_mailbox")
            else:
                with torch.cuda.device(sigma.device):
                    _cabi.check(_cabi.load().lp_sigma_times_mailbox(*args), "lp_sigma_times_mailbox")
        elif IS_FLUX or IS_FLOW:                                            # nodes.py:242-245
            Flow_t = sigma
            abt = (1 - Flow_t) ** 2 / ((1 - Flow_t) ** 2 + Flow_t ** 2)
            VE_Sigma = Flow_t / (1 - Flow_t)
        else:                                                               # nodes.py:249-252
            VE_Sigma = sigma
            abt = 1 / (1 + VE_Sigma ** 2)
            Flow_t = (1 - abt) ** 0.5 / ((1 - abt) ** 0.5 + abt ** 0.5)

        current_times_audio = audio_correction = None
        if self.audio_indicator is not None and self.audio_shifts is not None and time_shift_sigma is not None:
            shift_v, shift_a = self.audio_shifts                           # nodes.py:258-275
            Flow_a = time_shift_sigma(Flow_t, shift_v, shift_a)
            abt_a = (1 - Flow_a) ** 2 / ((1 - Flow_a) ** 2 + Flow_a ** 2)
            current_times_audio = (Flow_a / (1 - Flow_a), abt_a, Flow_a)
            ft, c = float(Flow_t), 1.0
            if ft > 1e-4 and time_shift_slope is not None:
                c = float(Flow_a) / (ft * float(time_shift_slope(Flow_t, shift_v, shift_a)))
            audio_correction = (1.0 - self.audio_indicator) + c * self.audio_indicator

        if denoise_mask is not None:
            if "denoise_mask_function" in model_options:
                denoise_mask = model_options["denoise_mask_function"](
                    sigma, denoise_mask, extra_options={"model": self.inner_model, "sigmas": self.sigmas})
            latent_mask = self._latent_mask(denoise_mask)
            current_times = (VE_Sigma, abt, Flow_t)
            # nodes.py:286-299.  Same device arithmetic as the reference; its two host syncs (argmin -> int
            # compare, float(mean)) become one poll of the mailbox -- and the part of the sigma call that does not
            # depend on the answer (replace step, coefficient table) is enqueued BEFORE the host waits for it, so
            # the GPU still has work when the answer arrives (engine.begin_call / finish_call).
            token = None
            pm = self.PaintMethod
            if fused_seq is not None:
                if current_times_audio is None and audio_correction is None and self.audio_indicator is None \
                        and hasattr(pm, "begin_call"):
                    token = pm.begin_call(x, self.latent_image, self.noise, sigma, latent_mask, current_times,
                                          model_options, seed)
                step_f, frac = self._wait_mailbox(fused_seq, sigma.device)
            else:
                current_step = torch.argmin(torch.abs(self.sigmas - torch.mean(sigma)))
                step_f, frac = torch.stack([current_step.to(torch.float32), (1.0 - abt).mean().to(torch.float32)]).tolist()
            total_steps = self._sched[4] if (self._sched is not None and self._sched[0] is self.sigmas) else len(self.sigmas) - 1
            n_eff = self.PaintMethod.n_steps
            if total_steps - int(step_f) <= self.LanPaint_early_stop:
                n_eff = 0
            else:
                n_eff = min_step_frac_effective_steps(n_eff, frac, getattr(self, "LanPaint_min_step_frac", 1.0))
            if token is not None:
                out = pm.finish_call(token, n_eff)
            else:
                out = pm(x, self.latent_image, self.noise, sigma, latent_mask, current_times, model_options,
                         seed, n_steps=n_eff, current_times_audio=current_times_audio,
                         audio_indicator=self.audio_indicator, audio_correction=audio_correction)
        else:
            out, _ = self.inner_model(x, sigma, model_options=model_options, seed=seed)

        step_i = model_options.get("i", kwargs.get("i", 0))                  # preview hook, nodes.py:304-313
        if step_i % 2 == 0:
            cb = model_options.get("callback", None)
            if cb is not None:
                cb({"i": step_i, "denoised": out, "x": x})
        return out


# =====================================================================================
# KSAMPLER.sample replacement + the monkey-patch set (nodes.py:318-421)
# =====================================================================================
_KSAMPLER_BASE = comfy.samplers.KSAMPLER if HAVE_COMFY and hasattr(comfy.samplers, "KSAMPLER") else object


class KSAMPLER(_KSAMPLER_BASE):
    def sample(self, model_wrap, sigmas, extra_args, callback, noise, latent_image=None, denoise_mask=None,
               disable_pbar=False):
        """nodes.py:319-379: build the KSamplerX0Inpaint callable around the HIP engine, then run
        ComfyUI's own sampler function over it."""
        extra_args["denoise_mask"] = denoise_mask
        model_k = KSamplerX0Inpaint(model_wrap, sigmas)
        model_k.latent_image = latent_image
        if self.inpaint_options.get("random", False):
            gen = torch.manual_seed(extra_args.get("seed", 41) + 1)
            model_k.noise = torch.randn(noise.shape, generator=gen, device="cpu").to(noise.dtype).to(noise.device)
        else:
            model_k.noise = noise

        base = model_wrap.inner_model
        patcher = model_wrap.model_patcher
        IS_FLUX = base.model_type == ModelType.FLUX
        IS_FLOW = base.model_type in FLOW_MODEL_TYPES
        model_wrap.cfg_BIG = 1.0 if IS_FLUX else patcher.LanPaint_cfg_BIG
        noise = base.model_sampling.noise_scaling(sigmas[0], noise, latent_image, self.max_denoise(model_wrap, sigmas))

        audio_layout = getattr(model_wrap, "minimax_h3_audio", None)
        if audio_layout is not None and time_shift_sigma is not None:         # nodes.py:340-349
            latent_shapes, shift_v, shift_a = audio_layout
            indicator = torch.zeros(noise.shape, dtype=torch.float32, device=noise.device)
            indicator[..., math.prod(latent_shapes[0][1:]):] = 1.0
            model_k.audio_indicator = indicator
            model_k.audio_shifts = (shift_v, shift_a)

        min_step_frac = getattr(patcher, "LanPaint_MinStepFrac", 1.0)
        model_k.PaintMethod = LanPaint(
            model_k.inner_model, patcher.LanPaint_NumSteps, patcher.LanPaint_Friction, patcher.LanPaint_Lambda,
            patcher.LanPaint_Beta, patcher.LanPaint_StepSize, IS_FLUX=IS_FLUX, IS_FLOW=IS_FLOW,
            EarlyStopThreshold=getattr(patcher, "LanPaint_InnerThreshold", 0.0),
            EarlyStopPatience=getattr(patcher, "LanPaint_InnerPatience", 1),
            EarlyStopHook=extra_args.get("model_options", {}).get("lanpaint_semantic_hook", None),
            MinStepFrac=min_step_frac)
        # this engine lives for ONE run; its noise tensor is created once per run (above / by the caller) and never rewritten
        model_k.PaintMethod.assume_static_noise = True
        model_k.LanPaint_early_stop = patcher.LanPaint_EarlyStop
        model_k.LanPaint_min_step_frac = min_step_frac

        total_steps = len(sigmas) - 1
        k_callback = None if callback is None else (lambda a: callback(a["i"], a["denoised"], a["x"], total_steps))
        samples = self.sampler_function(model_k, noise, sigmas, extra_args=extra_args, callback=k_callback,
                                        disable=disable_pbar, **self.extra_options)
        return base.model_sampling.inverse_noise_scaling(sigmas[-1], samples)


_override_active = False


@contextmanager
def override_sample_function():
    """Swap ComfyUI's CFGGuider.outer_sample / predict_noise, KSAMPLER.sample and
    sampler_helpers.prepare_mask for the LanPaint versions; always restored; a nested entry is a
    no-op so the originals are never lost (nodes.py:384-421)."""
    global _override_active
    if _override_active:
        yield
        return
    _require_comfy("override_sample_function")
    _override_active = True
    guider_cls, ksampler_cls, helpers = comfy.samplers.CFGGuider, comfy.samplers.KSAMPLER, comfy.sampler_helpers
    saved = (guider_cls.outer_sample, guider_cls.predict_noise, ksampler_cls.sample, helpers.prepare_mask)

    def _prepare_mask_with_union(noise_mask, shape, device):
        return prepare_mask(noise_mask, shape, device, video_inpainting=(len(shape) == 5))

    try:
        guider_cls.outer_sample = CFGGuider_LanPaint.outer_sample
        guider_cls.predict_noise = CFGGuider_LanPaint.predict_noise
        ksampler_cls.sample = KSAMPLER.sample
        helpers.prepare_mask = _prepare_mask_with_union
        yield
    finally:
        guider_cls.outer_sample, guider_cls.predict_noise, ksampler_cls.sample, helpers.prepare_mask = saved
        _override_active = False


# =====================================================================================
# sampler nodes (nodes.py:452-808): ComfyUI node protocol, signatures and defaults kept
# =====================================================================================
KSAMPLER_NAMES = ["euler", "euler_ancestral", "heun", "heunpp2", "dpm_2", "dpm_2_ancestral", "dpm_fast", "dpmpp_sde",
                  "dpmpp_sde_gpu", "dpmpp_2m", "dpmpp_2m_sde", "dpmpp_2m_sde_gpu", "dpmpp_3m_sde", "dpmpp_3m_sde_gpu",
                  "ddpm", "deis", "res_multistep", "res_multistep_ancestral", "gradient_estimation", "er_sde",
                  "seeds_2", "seeds_3"]
_PROMPT_MODES = ("Image First", "Prompt First")
_INPAINT_MODES = ("🖼️ Image Inpainting", "🎬 Video Inpainting")
_INFO_TIP = "For more info, visit https://github.com/scraed/LanPaint. If you find it useful, please give a star ⭐️!"
_NUMSTEPS = ("INT", {"default": 5, "min": 0, "max": 100,
                     "tooltip": "The number of steps for the Langevin dynamics, representing the turns of thinking per step."})
_PROMPT_MODE = (list(_PROMPT_MODES), {"tooltip": "Image First: emphasis image quality, Prompt First: emphasis prompt following"})
_INPAINT_MODE = (list(_INPAINT_MODES), {"default": _INPAINT_MODES[0],
                                        "tooltip": "Choose Image mode for photos or Video mode for video frames with temporal consistency"})
_LAMBDA = ("FLOAT", {"default": 5.0, "min": 0.1, "max": 50.0, "step": 0.1, "round": 0.1,
                     "tooltip": "The bidirectional guidance scale. Higher values align with known regions more closely, but may result in instability."})
_STEPSIZE = ("FLOAT", {"default": 0.2, "min": 0.0001, "max": 1.0, "step": 0.01, "round": 0.001,
                       "tooltip": "The step size for the Langevin dynamics. Higher values result in faster convergence but may be unstable."})
_RETIRED_ADV = {k: "DEFAULT" for k in ("LanPaint_Beta", "LanPaint_Friction", "LanPaint_EarlyStop",
                                       "LanPaint_InnerThreshold", "LanPaint_InnerPatience", "LanPaint_MinStepFrac")}


def _schedulers():
    if HAVE_COMFY and hasattr(comfy.samplers, "KSampler") and hasattr(comfy.samplers.KSampler, "SCHEDULERS"):
        return comfy.samplers.KSampler.SCHEDULERS
    return ["normal", "karras", "exponential", "sgm_uniform", "simple", "ddim_uniform", "beta"]


def _pin_hyperparams(model, cfg, num_steps, prompt_mode, lamb=5.0, step_size=0.2):
    """The node layer stores the engine's hyper-parameters on the model patcher
    (nodes.py:492-504 and its three siblings); retired knobs are pinned to fixed values."""
    model.LanPaint_StepSize = step_size
    model.LanPaint_Lambda = lamb
    model.LanPaint_Beta = 1.0
    model.LanPaint_NumSteps = num_steps
    model.LanPaint_MinStepFrac = 1.0
    model.LanPaint_Friction = 15.0
    model.LanPaint_EarlyStop = 1
    model.LanPaint_InnerThreshold = 0.0
    model.LanPaint_InnerPatience = 1
    model.LanPaint_cfg_BIG = cfg if prompt_mode == "Image First" else 0 * cfg - 0.5


def _set_video_mode(model, inpainting_mode):
    if not hasattr(model, "model_options") or model.model_options is None:
        model.model_options = {}
    model.model_options["video_inpainting"] = (inpainting_mode == _INPAINT_MODES[1])


class LanPaint_KSampler:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "model": ("MODEL", {"tooltip": "The model used for denoising the input latent."}),
                "seed": ("INT", {"default": 0, "min": 0, "max": 0xffffffffffffffff, "tooltip": "The random seed used for creating the noise."}),
                "steps": ("INT", {"default": 30, "min": 1, "max": 10000, "tooltip": "The number of steps used in the denoising process."}),
                "cfg": ("FLOAT", {"default": 5.0, "min": 0.0, "max": 100.0, "step": 0.1, "round": 0.01,
                                  "tooltip": "The Classifier-Free Guidance scale."}),
                "sampler_name": (KSAMPLER_NAMES, {"tooltip": "Recommended: euler."}),
                "scheduler": (_schedulers(), {"default": "karras", "tooltip": "The scheduler controls how noise is gradually removed to form the image."}),
                "positive": ("CONDITIONING", {"tooltip": "The conditioning describing the attributes you want to include in the image."}),
                "negative": ("CONDITIONING", {"tooltip": "The conditioning describing the attributes you want to exclude from the image."}),
                "latent_image": ("LATENT", {"tooltip": "The latent image to denoise."}),
                "denoise": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 1.0, "step": 0.01, "tooltip": "The amount of denoising applied."}),
                "LanPaint_NumSteps": _NUMSTEPS,
                "LanPaint_PromptMode": _PROMPT_MODE,
                "LanPaint_Info": ("STRING", {"default": "LanPaint KSampler.", "tooltip": _INFO_TIP}),
                "Inpainting_mode": _INPAINT_MODE,
            },
            "hidden": {"LanPaint_MinStepFrac": "DEFAULT"},
        }

    RETURN_TYPES = ("LATENT",)
    OUTPUT_TOOLTIPS = ("The denoised latent.",)
    FUNCTION = "sample"
    CATEGORY = "sampling"
    DESCRIPTION = "Uses the provided model, positive and negative conditioning to denoise the latent image."

    def sample(self, model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0,
               LanPaint_NumSteps=5, LanPaint_PromptMode="Image First", LanPaint_Info="",
               Inpainting_mode=_INPAINT_MODES[0], **kwargs):
        import nodes as comfy_nodes      # ComfyUI's nodes.py
        num_steps = _sanitize_param(LanPaint_NumSteps, 5)
        mode = _sanitize_param(LanPaint_PromptMode, "Image First", allowed=_PROMPT_MODES)
        _pin_hyperparams(model, cfg, num_steps, mode)
        _set_video_mode(model, _sanitize_param(Inpainting_mode, _INPAINT_MODES[0], allowed=_INPAINT_MODES))
        with override_sample_function():
            return comfy_nodes.common_ksampler(model, seed, steps, cfg, sampler_name, scheduler, positive, negative,
                                               latent_image, denoise=denoise)


class LanPaint_KSamplerAdvanced:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "model": ("MODEL",),
                "add_noise": (["enable", "disable"],),
                "noise_seed": ("INT", {"default": 0, "min": 0, "max": 0xffffffffffffffff}),
                "steps": ("INT", {"default": 30, "min": 1, "max": 10000}),
                "cfg": ("FLOAT", {"default": 5.0, "min": 0.0, "max": 100.0, "step": 0.1, "round": 0.01}),
                "sampler_name": (KSAMPLER_NAMES,),
                "scheduler": (_schedulers(),),
                "positive": ("CONDITIONING",),
                "negative": ("CONDITIONING",),
                "latent_image": ("LATENT",),
                "start_at_step": ("INT", {"default": 0, "min": 0, "max": 10000}),
                "end_at_step": ("INT", {"default": 10000, "min": 0, "max": 10000}),
                "return_with_leftover_noise": (["disable", "enable"],),
                "LanPaint_NumSteps": _NUMSTEPS,
                "LanPaint_Lambda": _LAMBDA,
                "LanPaint_StepSize": _STEPSIZE,
                "LanPaint_PromptMode": _PROMPT_MODE,
                "LanPaint_Info": ("STRING", {"default": "LanPaint KSampler Adv.", "tooltip": _INFO_TIP}),
                "Inpainting_mode": _INPAINT_MODE,
            },
            "hidden": dict(_RETIRED_ADV),
        }

    RETURN_TYPES = ("LATENT",)
    FUNCTION = "sample"
    CATEGORY = "sampling"

    def sample(self, model, add_noise, noise_seed, steps, cfg, sampler_name, scheduler, positive, negative,
               latent_image, start_at_step, end_at_step, return_with_leftover_noise, LanPaint_NumSteps=5,
               LanPaint_Lambda=5.0, LanPaint_StepSize=0.2, LanPaint_PromptMode="Image First", LanPaint_Info="",
               Inpainting_mode=_INPAINT_MODES[0], **kwargs):
        import nodes as comfy_nodes
        _pin_hyperparams(model, cfg, _sanitize_param(LanPaint_NumSteps, 5),
                         _sanitize_param(LanPaint_PromptMode, "Image First", allowed=_PROMPT_MODES),
                         lamb=_sanitize_param(LanPaint_Lambda, 5.0), step_size=_sanitize_param(LanPaint_StepSize, 0.2))
        _set_video_mode(model, _sanitize_param(Inpainting_mode, _INPAINT_MODES[0], allowed=_INPAINT_MODES))
        with override_sample_function():
            return comfy_nodes.common_ksampler(
                model, noise_seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0,
                disable_noise=(add_noise == "disable"), start_step=start_at_step, last_step=end_at_step,
                force_full_denoise=(return_with_leftover_noise != "enable"))


class Noise_EmptyNoise:
    def __init__(self):
        self.seed = 0

    def generate_noise(self, input_latent):
        s = input_latent["samples"]
        return torch.zeros(s.shape, dtype=s.dtype, layout=s.layout, device="cpu")


class Noise_RandomNoise:
    def __init__(self, seed):
        self.seed = seed

    def generate_noise(self, input_latent):
        batch_inds = input_latent["batch_index"] if "batch_index" in input_latent else None
        return comfy.sample.prepare_noise(input_latent["samples"], self.seed, batch_inds)


def _finish_custom(latent, samples, x0_output, model_patcher):
    out = latent.copy()
    out["samples"] = samples
    if "x0" in x0_output:
        out_denoised = latent.copy()
        out_denoised["samples"] = model_patcher.model.process_latent_out(x0_output["x0"].cpu())
    else:
        out_denoised = out
    return out, out_denoised


class LanPaint_SamplerCustom:
    @classmethod
    def INPUT_TYPES(s):
        return {"required": {
            "model": ("MODEL",),
            "add_noise": ("BOOLEAN", {"default": True}),
            "noise_seed": ("INT", {"default": 0, "min": 0, "max": 0xffffffffffffffff, "control_after_generate": True}),
            "cfg": ("FLOAT", {"default": 8.0, "min": 0.0, "max": 100.0, "step": 0.1, "round": 0.01}),
            "positive": ("CONDITIONING",),
            "negative": ("CONDITIONING",),
            "sampler": ("SAMPLER",),
            "sigmas": ("SIGMAS",),
            "latent_image": ("LATENT",),
            "LanPaint_NumSteps": _NUMSTEPS,
            "LanPaint_PromptMode": _PROMPT_MODE,
            "LanPaint_Info": ("STRING", {"default": "LanPaint Custom Sampler.", "tooltip": _INFO_TIP}),
        }}

    RETURN_TYPES = ("LATENT", "LATENT")
    RETURN_NAMES = ("output", "denoised_output")
    FUNCTION = "sample"
    CATEGORY = "sampling/custom_sampling"

    def sample(self, model, sampler, sigmas, add_noise, noise_seed, cfg, positive, negative, latent_image,
               LanPaint_NumSteps, LanPaint_PromptMode, LanPaint_Info=""):
        import latent_preview
        _pin_hyperparams(model, cfg, _sanitize_param(LanPaint_NumSteps, 5),
                         _sanitize_param(LanPaint_PromptMode, "Image First", allowed=_PROMPT_MODES))
        with override_sample_function():
            latent = latent_image.copy()
            latent["samples"] = comfy.sample.fix_empty_latent_channels(model, latent["samples"])
            noise = (Noise_RandomNoise(noise_seed) if add_noise else Noise_EmptyNoise()).generate_noise(latent)
            x0_output = {}
            callback = latent_preview.prepare_callback(model, sigmas.shape[-1] - 1, x0_output)
            samples = comfy.sample.sample_custom(model, noise, cfg, sampler, sigmas, positive, negative,
                                                 latent["samples"], noise_mask=latent.get("noise_mask"),
                                                 callback=callback, disable_pbar=not comfy.utils.PROGRESS_BAR_ENABLED,
                                                 seed=noise_seed)
            return _finish_custom(latent, samples, x0_output, model)


class LanPaint_SamplerCustomAdvanced:
    @classmethod
    def INPUT_TYPES(s):
        return {"required": {
            "noise": ("NOISE",),
            "guider": ("GUIDER",),
            "sampler": ("SAMPLER",),
            "sigmas": ("SIGMAS",),
            "latent_image": ("LATENT",),
            "LanPaint_NumSteps": _NUMSTEPS,
            "LanPaint_Lambda": ("FLOAT", {k: v for k, v in _LAMBDA[1].items() if k != "round"}),
            "LanPaint_StepSize": ("FLOAT", {k: v for k, v in _STEPSIZE[1].items() if k != "round"}),
            "LanPaint_PromptMode": _PROMPT_MODE,
            "LanPaint_Info": ("STRING", {"default": "LanPaint Custom Sampler Adv.", "tooltip": _INFO_TIP}),
        }, "hidden": dict(_RETIRED_ADV)}

    RETURN_TYPES = ("LATENT", "LATENT")
    RETURN_NAMES = ("output", "denoised_output")
    FUNCTION = "sample"
    CATEGORY = "sampling/custom_sampling"

    def sample(self, noise, guider, sampler, sigmas, latent_image, LanPaint_NumSteps, LanPaint_Lambda,
               LanPaint_StepSize, LanPaint_PromptMode, LanPaint_Info="", **kwargs):
        import latent_preview
        patcher = guider.model_patcher
        _pin_hyperparams(patcher, guider.cfg, _sanitize_param(LanPaint_NumSteps, 5),
                         _sanitize_param(LanPaint_PromptMode, "Image First", allowed=_PROMPT_MODES),
                         lamb=_sanitize_param(LanPaint_Lambda, 5.0), step_size=_sanitize_param(LanPaint_StepSize, 0.2))
        with override_sample_function():
            latent = latent_image.copy()
            latent["samples"] = comfy.sample.fix_empty_latent_channels(patcher, latent["samples"])
            x0_output = {}
            callback = latent_preview.prepare_callback(patcher, sigmas.shape[-1] - 1, x0_output)
            samples = guider.sample(noise.generate_noise(latent), latent["samples"], sampler, sigmas,
                                    denoise_mask=latent.get("noise_mask"), callback=callback,
                                    disable_pbar=not comfy.utils.PROGRESS_BAR_ENABLED, seed=noise.seed)
            samples = samples.to(comfy.model_management.intermediate_device())
            return _finish_custom(latent, samples, x0_output, patcher)


NODE_CLASS_MAPPINGS = {
    "LanPaint_ImageEncode": LanPaint_ImageEncode,
    "LanPaint_ImageDecode": LanPaint_ImageDecode,
    "LanPaint_KSampler": LanPaint_KSampler,
    "LanPaint_KSamplerAdvanced": LanPaint_KSamplerAdvanced,
    "LanPaint_SamplerCustom": LanPaint_SamplerCustom,
    "LanPaint_SamplerCustomAdvanced": LanPaint_SamplerCustomAdvanced,
    "LanPaint_MaskBlend": MaskBlend,
}
NODE_DISPLAY_NAME_MAPPINGS = {
    "LanPaint_ImageEncode": "LanPaint Image Encode",
    "LanPaint_ImageDecode": "LanPaint Image Decode",
    "LanPaint_MaskBlend": "LanPaint Mask Blend",
    "LanPaint_KSampler": "LanPaint KSampler",
    "LanPaint_KSamplerAdvanced": "LanPaint KSampler (Advanced)",
    "LanPaint_SamplerCustom": "LanPaint Sampler Custom",
    "LanPaint_SamplerCustomAdvanced": "LanPaint Sampler Custom (Advanced)",
}
