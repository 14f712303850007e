"""CPU oracle for LanPaint's Langevin "think" loop.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (numpy fp32 by default, torch-CPU
through the same code via `TorchBackend`) of the algorithm in the reference
    /root/reference/src/LanPaint/lanpaint.py      (engine)
    /root/reference/src/LanPaint/earlystop.py     (optional inner early stop)
    /root/reference/src/LanPaint/nodes.py:59-160,229-315  (mask prep, sigma->times)
It exists so the HIP path can be checked on a box where /root/reference is
absent.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product package `lanpaint_amd` never imports it
and has no CPU fallback.

Parity status: PINNED.  `tests/golden/*.npz` hold trajectories produced by the
unmodified reference engine (script: tests/golden/make_golden.py, run in the
build container where /root/reference exists); `tests/test_oracle_golden.py`
checks this oracle against them and against the reference tests' known-answer
values (test_av_schedule.py:179-324, test_min_step_frac.py:18-36,
test_reshape_mask.py, test_videomask.py:475-713).

Each function cites the reference lines it restates.  Arithmetic is kept in
the reference's operation order so fp32 rounding follows it closely.
"""
from __future__ import annotations

import math
from typing import Any, Callable, NamedTuple, Optional, Sequence

import numpy as np


# ----------------------------------------------------------------------------
# array backends (numpy is the oracle proper; torch-CPU reproduces the
# reference's own ATen cost structure for the cpu_baseline timing)
# ----------------------------------------------------------------------------
class NumpyBackend:
    name = "numpy"

    @staticmethod
    def asarray(a, like=None):
        return np.asarray(a, dtype=np.float32 if like is None else like.dtype)

    exp = staticmethod(np.exp)
    expm1 = staticmethod(np.expm1)
    abs = staticmethod(np.abs)
    sqrt = staticmethod(np.sqrt)
    where = staticmethod(np.where)

    @staticmethod
    def clamp_min(a, lo):
        return np.maximum(a, np.asarray(lo, dtype=a.dtype))

    @staticmethod
    def mean_float(a):
        return float(np.mean(a))

    @staticmethod
    def sum_float(a):
        return float(np.sum(a))

    @staticmethod
    def sum(a):
        return np.sum(a, dtype=np.float32)

    @staticmethod
    def clone(a):
        return np.array(a, copy=True)

    @staticmethod
    def ndim(a):
        return np.ndim(a)

    @staticmethod
    def reshape(a, shape):
        return np.reshape(a, shape)

    @staticmethod
    def copy_into(dst, src):
        dst[...] = src

    @staticmethod
    def zeros_like_bool(a):
        return np.zeros(a.shape, dtype=bool)

    @staticmethod
    def to_f32(a):
        return a.astype(np.float32)

    @staticmethod
    def randn_like(a, rng=None):
        rng = rng or np.random.default_rng(0)
        return rng.standard_normal(a.shape, dtype=np.float32)


class TorchBackend:
    """Same oracle code on torch CPU tensors (used only for cpu_baseline timing
    and for feeding the oracle the exact torch RNG stream)."""
    name = "torch"

    def __init__(self):
        import torch
        self.torch = torch
        self.exp, self.expm1, self.abs = torch.exp, torch.expm1, torch.abs
        self.sqrt, self.where = torch.sqrt, torch.where

    def asarray(self, a, like=None):
        t = self.torch
        if isinstance(a, t.Tensor):
            return a
        return t.as_tensor(a, dtype=t.float32 if like is None else like.dtype)

    def clamp_min(self, a, lo):
        return self.torch.clamp(a, min=lo)

    def mean_float(self, a):
        return float(self.torch.mean(a))

    def sum_float(self, a):
        return float(self.torch.sum(a))

    def sum(self, a):
        return self.torch.sum(a)

    def clone(self, a):
        return a.detach().clone()

    def ndim(self, a):
        return a.ndim

    def reshape(self, a, shape):
        return a.reshape(shape)

    def copy_into(self, dst, src):
        dst.copy_(src)

    def zeros_like_bool(self, a):
        return self.torch.zeros_like(a, dtype=self.torch.bool)

    def to_f32(self, a):
        return a.to(self.torch.float32)

    def randn_like(self, a, rng=None):
        return self.torch.randn_like(a)


class OracleState(NamedTuple):
    """(v, C, x0) carried between think iterations; v is always None in the live
    overdamped scheme.  Restates types.py:6-9."""
    v: Any
    C: Any
    x0: Any


# ----------------------------------------------------------------------------
# a1/a2: sigma -> (VE_sigma, abt, flow_t), effective inner-step count
# ----------------------------------------------------------------------------
def times_from_sigma(sigma, is_flow: bool):
    """nodes.py:242-252.  Returns (VE_Sigma, abt, Flow_t) elementwise on `sigma`."""
    if is_flow:
        flow_t = sigma
        abt = (1 - flow_t) ** 2 / ((1 - flow_t) ** 2 + flow_t ** 2)
        ve = flow_t / (1 - flow_t)
    else:
        ve = sigma
        abt = 1 / (1 + ve ** 2)
        flow_t = (1 - abt) ** 0.5 / ((1 - abt) ** 0.5 + abt ** 0.5)
    return ve, abt, flow_t


def min_step_frac_effective_steps(n_steps: int, frac: float, min_frac: float) -> int:
    """nodes.py:134-144 (Python round == banker's rounding)."""
    if min_frac <= 0 or frac >= min_frac or n_steps <= 0:
        return n_steps
    return max(0, round(n_steps * frac / min_frac))


def effective_inner_steps(n_steps: int, sigmas: Sequence[float], sigma_now: float, abt_mean: float,
                          early_stop: int, min_step_frac: float) -> int:
    """nodes.py:286-299: 0 on the last `early_stop` sigmas, else the MinStepFrac ramp."""
    sig = np.asarray(sigmas, dtype=np.float32)
    current_step = int(np.argmin(np.abs(sig - np.float32(sigma_now))))
    total_steps = len(sig) - 1
    if total_steps - current_step <= early_stop:
        return 0
    return min_step_frac_effective_steps(n_steps, float(1.0 - abt_mean), min_step_frac)


def binarize_and_invert(denoise_mask):
    """nodes.py:281-283: latent_mask = 1 - (denoise_mask > 0.5)."""
    return (1.0 - (np.asarray(denoise_mask) > 0.5).astype(np.float32)).astype(np.float32)


# ----------------------------------------------------------------------------
# a17: mask preparation -- index math bit-for-bit with torch's nearest-exact
# ----------------------------------------------------------------------------
def nearest_exact_src_index(out_size: int, in_size: int, rule: str = "scalar") -> np.ndarray:
    """Source index chosen by F.interpolate(mode="nearest-exact") for each output index
    (nodes.py:78,88,110-114,124-127 call sites).  ATen evaluates it in FLOAT32, in one of three forms depending on the kernel
    the call dispatches to (scale = float(in) / float(out) in all of them):
      "scalar"       src = min(int(floorf((i + 0.5f) * scale)), in - 1)
                     (aten/src/ATen/native/UpSample.h nearest_neighbor_exact_compute_source_index / nearest_exact_idx:
                     torch's GPU kernels; CPU: the 2-D kernel when out_h + out_w <= 128, channels-last with > 3 channels)
      "generic_fma"  s = max(fma(scale, i + 0.5f, -0.5f), 0);  src = min(int(floorf(float(double(s) + 0.5))), in - 1)
                     (aten/src/ATen/native/cpu/UpSampleKernel.cpp HelperInterpNearestExact -- the CPU's TensorIterator kernel:
                     1-D, 3-D, 2-D with out_h + out_w > 128 -- as its AVX2 / AVX512 builds contract the multiply-subtract)
      "generic"      the same with product and subtraction rounded separately (ATEN_CPU_CAPABILITY=default).
    The fp32 rounding is part of the reference's behaviour: the exact-rational index ((2i+1)*in)//(2*out) differs where
    (i+0.5)*in/out is an integer that fp32 lands just below (e.g. in=14, out=201, i=100 -> 6, not 7, by the scalar rule), and
    the three forms differ from each other on such ties when UP-sampling (2 -> 41 at i = 20: scalar 0, generic_fma 1);
    on every down-sampling pair they agree.  Each is restated op for op."""
    scale = np.float32(in_size) / np.float32(out_size)
    at = np.arange(out_size, dtype=np.float32) + np.float32(0.5)
    if rule == "scalar":
        return np.minimum(np.floor(at * scale).astype(np.int64), in_size - 1)
    if rule == "generic_fma":       # one rounding: the product of two fp32 values is exact in double, and so is the - 0.5
        src = (at.astype(np.float64) * np.float64(scale) - 0.5).astype(np.float32)
    elif rule == "generic":
        src = at * scale - np.float32(0.5)
    else:
        raise ValueError(rule)
    src = np.where(src < 0, np.float32(0), src)
    q = (src.astype(np.float64) + 0.5).astype(np.float32)
    return np.minimum(np.floor(q).astype(np.int64), in_size - 1)


def aten_nearest_exact_rule(mask_on: str, spatial_dims: int, out_sizes: Sequence[int], channels: int = 1,
                            channels_last: bool = False, cpu_fma: bool = True) -> str:
    """Which of the three forms one F.interpolate(mode="nearest-exact") call follows -- ATen's dispatch restated
    (upsample_nearest_exact{1,2,3}d_kernel_impl, _use_vectorized_kernel_cond_2d): mask_on "gpu" -> scalar; "cpu":
    channels-last with more than 3 channels -> scalar; 2-D with out_h + out_w <= 128 -> scalar; else the TensorIterator
    kernel, contracted unless `cpu_fma` is False."""
    if mask_on == "gpu":
        return "scalar"
    if channels_last and channels > 3 and spatial_dims in (2, 3):
        return "scalar"
    if spatial_dims == 2 and int(out_sizes[-2]) + int(out_sizes[-1]) <= 128:
        return "scalar"
    return "generic_fma" if cpu_fma else "generic"


def _interp_nearest_exact(a: np.ndarray, sizes: Sequence[int], mask_on: str = "gpu", cpu_fma: bool = True) -> np.ndarray:
    """Nearest-exact resample of the trailing len(sizes) axes of `a` ([N, C, *spatial]) as torch does it on `mask_on`."""
    nd = len(sizes)
    rule = aten_nearest_exact_rule(mask_on, nd, sizes, a.shape[1] if a.ndim >= 2 else 1, False, cpu_fma)
    for k, out_size in enumerate(sizes):
        axis = a.ndim - nd + k
        a = np.take(a, nearest_exact_src_index(int(out_size), a.shape[axis], rule), axis=axis)
    return a


def _repeat_to_batch_size(a: np.ndarray, batch: int) -> np.ndarray:
    """comfy.utils.repeat_to_batch_size as restated by the reference's own test
    stubs (tests/test_reshape_mask.py:9-15)."""
    if a.shape[0] == batch:
        return a
    if a.shape[0] == 1:
        return np.repeat(a, batch, axis=0)
    reps = (batch + a.shape[0] - 1) // a.shape[0]
    return np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:batch]


def reshape_mask(input_mask, output_shape: Sequence[int], video_inpainting: bool = False,
                 comfy_060_or_newer: bool = True, mask_on: str = "gpu", cpu_fma: bool = True) -> np.ndarray:
    """nodes.py:59-133.  numpy float32 in / out; output has `output_shape`.  `mask_on`: the device the reference holds the
    mask on when it resamples ("cpu": ComfyUI's host tensors, nodes.py:159-160 resamples before `.to(device)`; "gpu"): torch's
    kernels there decide the index rule (nearest_exact_src_index)."""
    m = np.asarray(input_mask, dtype=np.float32)
    output_shape = tuple(int(s) for s in output_shape)
    dims = len(output_shape) - 2
    if video_inpainting:                                        # :64-73
        if m.ndim == 3:
            m = m[None, None]
        elif m.ndim == 4:
            m = np.transpose(m, (1, 0, 2, 3))[None]
        elif m.ndim == 2:
            m = m[None, None, None]
    elif m.ndim == 1 and len(output_shape) == 4:                # :74-83 audio [F]
        t = output_shape[-1]
        m = _interp_nearest_exact(m[None, None], (t,), mask_on, cpu_fma)
        m = np.broadcast_to(m[:, :, None, :], (1, 1, output_shape[-2], t))
    elif m.ndim == 4 and len(output_shape) == 4 and m.shape[1] == 1 and m.shape[3] == 1:   # :84-89
        t = output_shape[-1]
        m = _interp_nearest_exact(m, (t, 1), mask_on, cpu_fma)
        m = np.broadcast_to(np.transpose(m, (0, 1, 3, 2)), (1, 1, output_shape[-2], t))
    elif m.ndim == 2:                                           # :90-91
        m = m[None, None]
    elif m.ndim == 3:                                           # :92-93
        m = m[:, None]
    if len(output_shape) == 5 and m.ndim == 4 and comfy_060_or_newer:   # :96-98
        m = m[:, :, None]
    if video_inpainting:                                        # :100-122
        tf = output_shape[2]
        th, tw = output_shape[-2:]
        m = _interp_nearest_exact(m, (tf, th, tw), mask_on, cpu_fma)
        # max_pool3d kernel (5,1,1) stride 1 pad (2,0,0): -inf padding
        f = m.shape[2]
        padded = np.full(m.shape[:2] + (f + 4,) + m.shape[3:], -np.inf, dtype=np.float32)
        padded[:, :, 2:2 + f] = m
        pooled = padded[:, :, 0:f]
        for k in range(1, 5):
            pooled = np.maximum(pooled, padded[:, :, k:k + f])
        m = pooled
        if m.shape[1] < output_shape[1]:
            m = np.tile(m, (1, output_shape[1], 1, 1, 1))[:, :output_shape[1]]
        m = _repeat_to_batch_size(m, output_shape[0])
    else:                                                       # :123-130
        sizes = output_shape[2:] if comfy_060_or_newer else output_shape[-2:]
        m = _interp_nearest_exact(m, sizes, mask_on, cpu_fma)
        if m.shape[1] < output_shape[1]:
            m = np.tile(m, (1, output_shape[1]) + (1,) * dims)[:, :output_shape[1]]
        m = _repeat_to_batch_size(m, output_shape[0])
    return np.ascontiguousarray(m, dtype=np.float32)


# ----------------------------------------------------------------------------
# a16: early-stop metric pieces
# ----------------------------------------------------------------------------
def abt_scale(abt_val: float) -> float:
    """earlystop.py:13-29."""
    c = min(1.0, max(0.0, abt_val))
    return min(1.0, max(0.0, 4.0 * c * (1.0 - c)))


def boundary_weight(latent_mask, inpaint_weight, xp=None):
    """earlystop.py:32-49: inpaint pixels 4-adjacent (H,W) to a known pixel; 4-D only."""
    xp = xp or NumpyBackend()
    if xp.ndim(latent_mask) != 4:
        return None
    known = latent_mask > 0.5
    nb = xp.zeros_like_bool(known)
    nb[:, :, 1:, :] |= known[:, :, :-1, :]
    nb[:, :, :-1, :] |= known[:, :, 1:, :]
    nb[:, :, :, 1:] |= known[:, :, :, :-1]
    nb[:, :, :, :-1] |= known[:, :, :, 1:]
    return xp.to_f32((~known) & nb) * inpaint_weight


def weighted_mse(a, b, w, xp=None) -> float:
    """earlystop.py:52-55 (fp32 sums, +1e-12 in the denominator)."""
    xp = xp or NumpyBackend()
    d2 = (xp.to_f32(a) - xp.to_f32(b)) ** 2
    denom = xp.sum(w) + 1e-12
    return float(xp.sum(d2 * w) / denom)


class OracleEarlyStopper:
    """earlystop.py:58-336 restated for the default metric (no custom distance_fn, which is host Python in the reference and
    stays host Python in the product).  Works on either backend; `trace_sink` (a list) receives the reference's per-iteration
    record (earlystop.py:315-334) with `tags` = (case_id, outer_step, bench_timestep)."""

    def __init__(self, threshold: float, patience: int, latent_mask, abt_mean: float, xp=None, trace_sink=None,
                 tags=(None, None, None)):
        self.xp = xp = xp or NumpyBackend()
        self.enabled = (threshold > 0.0) and (patience > 0)
        self.patience_eff = max(1, patience) + 1
        self.threshold = threshold
        self.threshold_eff = threshold * abt_scale(abt_mean) if self.enabled else threshold
        self.abt_val = abt_mean
        self.inpaint = self.ring = None
        if self.enabled and self.threshold_eff <= 0.0:
            self.enabled = False
        if self.enabled:
            self.inpaint = xp.to_f32(1 - latent_mask)
            if xp.sum_float(self.inpaint) < 1e-6:
                self.enabled = False
            else:
                self.ring = boundary_weight(latent_mask, self.inpaint, xp)
        self.counter = 0
        self.anchor = None
        self.trace = []
        self.trace_sink, self.tags = (trace_sink if isinstance(trace_sink, list) else None), tags

    def _pair(self, a, b):
        d_in = weighted_mse(a, b, self.inpaint, self.xp)
        d_ring = weighted_mse(a, b, self.ring, self.xp) if self.ring is not None else None
        return d_in, d_ring, (d_in if d_ring is None else max(d_in, d_ring))

    def step(self, x_before, x_after, prev_state, state) -> bool:
        """earlystop.py:238-336."""
        if not self.enabled:
            return False
        x0_prev = None if prev_state is None else prev_state.x0
        x0_cur = None if state is None else state.x0
        d_ring = d_drift = None
        if x0_prev is not None and x0_cur is not None:
            d_in, d_ring, dist = self._pair(x0_cur, x0_prev)
        else:
            d_in = dist = weighted_mse(x_after, x_before, self.inpaint, self.xp)
        if x0_cur is not None:
            if dist <= self.threshold_eff:
                if self.anchor is None:
                    self.anchor = self.xp.clone(x0_cur)
                else:
                    d_drift = self._pair(x0_cur, self.anchor)[2]
                    dist = max(dist, d_drift)
            else:
                self.anchor = None
        if dist <= self.threshold_eff:
            self.counter += 1
        else:
            self.counter = 0
            self.anchor = None
        stop = self.counter >= self.patience_eff
        self.trace.append({"dist": dist, "counter": self.counter, "stopped": stop})
        if self.trace_sink is not None:
            self.trace_sink.append({"case_id": self.tags[0], "outer_step": self.tags[1], "bench_timestep": self.tags[2],
                                    "inner_step": len(self.trace), "dist": dist, "dist_inpaint": d_in, "dist_ring": d_ring,
                                    "dist_drift": d_drift, "threshold": float(self.threshold_eff),
                                    "threshold_eff": float(self.threshold_eff), "patience_counter": int(self.counter),
                                    "patience_eff": int(self.patience_eff), "abt": float(self.abt_val), "custom_dist": False,
                                    "stopped": bool(stop)})
        return stop


# ----------------------------------------------------------------------------
# the engine (a3-a15)
# ----------------------------------------------------------------------------
def unpack_model_output(output):
    """lanpaint.py:34-43."""
    if isinstance(output, (tuple, list)):
        if len(output) >= 2:
            return output[0], output[1]
        if len(output) == 1:
            return output[0], output[0]
        raise ValueError("Model output is empty")
    return output, output


class OracleLanPaint:
    """Restatement of the reference engine class (lanpaint.py:7-328).

    model(x, t, model_options=None, seed=None) -> array | (x0, x0_BIG)
    noise_scaling(sigma_broadcast, noise, latent) -> array   (replace-step source,
        lanpaint.py:84-92; defaults to the VE / flow forms the reference's own
        test stubs pin: tests/test_lanpaint_semantic_stop.py:7-8,
        tests/test_av_schedule.py:117-119)
    randn(like) -> array: the xi source (lanpaint.py:252 draws torch.randn_like).
    """

    def __init__(self, model: Callable, n_steps: int, friction: float, lamb: float, beta: float,
                 step_size: float, is_flux: bool = False, is_flow: bool = False,
                 early_stop_threshold: float = 0.0, early_stop_patience: int = 1,
                 min_step_frac: float = 0.0, backend=None, randn: Optional[Callable] = None,
                 noise_scaling: Optional[Callable] = None, noise_scale: float = 1.0):
        self.model = model
        self.n_steps = n_steps
        self.friction = friction
        self.lamb = lamb
        self.beta = beta
        self.step_size = step_size
        self.is_flux = is_flux
        self.is_flow = is_flow
        self.early_stop_threshold = early_stop_threshold
        self.early_stop_patience = early_stop_patience
        self.min_step_frac = min_step_frac
        self.xp = backend or NumpyBackend()
        self._randn = randn or (lambda like: self.xp.randn_like(like))
        self._noise_scaling = noise_scaling
        self.noise_scale = noise_scale
        self.ndim = None
        self.iterations_run = 0          # think iterations executed (for it/s accounting)
        self.model_calls = 0

    # -- helpers -------------------------------------------------------------
    def _bcast(self, a):
        """add_none_dims, lanpaint.py:23-29."""
        a = self.xp.asarray(a)
        nd = self.xp.ndim(a)
        if nd < self.ndim:
            a = self.xp.reshape(a, tuple(a.shape) + (1,) * (self.ndim - nd))
        return a

    def _row_scalar(self, a):
        """remove_none_dims, lanpaint.py:30-33: [B,1,...] -> [B]."""
        return a[(slice(None),) + (0,) * (self.ndim - 1)]

    def _call_model(self, x, t, model_options, seed):
        self.model_calls += 1
        return unpack_model_output(self.model(x, t, model_options=model_options, seed=seed))

    # -- a4: replace step ------------------------------------------------------
    def _known_region(self, sigma, noise, latent):
        """lanpaint.py:84-92 scale_latent_inpaint."""
        s = self._bcast(sigma)
        if self._noise_scaling is not None and int(np.prod(s.shape)) == 1:
            return self._noise_scaling(s, noise, latent)
        if self.is_flux or self.is_flow or int(np.prod(s.shape)) != 1:
            return s * (self.noise_scale * noise) + (1.0 - s) * latent
        return latent + noise * s

    # -- a8: per-sigma coefficients -------------------------------------------
    def step_coefficients(self, current_times, step_size, sigma_x, sigma_y):
        """prepare_step_size, lanpaint.py:295-328 (Gamma_* omitted: computed by the
        reference but never consumed by the live overdamped scheme)."""
        _sigma, abt, _flow_t = current_times
        abt = self._bcast(abt)
        dtx = 2 * step_size * sigma_x
        dty = 2 * step_size * sigma_y
        a_t_x = (1) / (1 - abt) * dtx / 2
        a_t_y = (1 + self.lamb) / (1 - abt) * dty / 2
        a_x = a_t_x / (dtx / 2)
        a_y = a_t_y / (dty / 2)
        d_x = (2 * abt ** 0) ** 0.5
        d_y = (2 * abt ** 0) ** 0.5
        return abt, dtx / 2, dty / 2, a_x, a_y, d_x, d_y

    # -- a7: masked score split -------------------------------------------------
    def score(self, x_t, y, mask, abt, sigma, tflow, model_options, seed, audio_correction=None):
        """score_model, lanpaint.py:159-184."""
        if self.is_flux or self.is_flow:
            x = x_t / (abt ** 0.5 + (1 - abt) ** 0.5)
            x0, x0_big = self._call_model(x, self._row_scalar(tflow), model_options, seed)
        else:
            x = x_t * (1 + sigma ** 2) ** 0.5
            x0, x0_big = self._call_model(x, self._row_scalar(sigma), model_options, seed)
        if audio_correction is not None:
            x0 = x + audio_correction * (x0 - x)
            x0_big = x + audio_correction * (x0_big - x)
        score_x = -(x_t - x0)
        score_y = -(1 + self.lamb) * (x_t - y) + self.lamb * (x_t - x0_big)
        return score_x * (1 - mask) + score_y * mask

    # -- a11: exact OU step + noise -----------------------------------------------
    def ou_step(self, x_t, dt, a, c, d):
        """advance_time_overdamped, lanpaint.py:232-254."""
        xp = self.xp
        a_dt = a * dt
        e = xp.exp(-a_dt)
        small = xp.abs(a) < 1e-8
        k = xp.where(small, dt, (-xp.expm1(-a_dt)) / a)
        k2 = xp.where(small, dt, (-xp.expm1(-2 * a_dt)) / (2 * a))
        mean = e * x_t + k * c
        var = (d ** 2) * k2
        return mean + self._randn(x_t) * xp.sqrt(xp.clamp_min(var, 0.0))

    # -- a9-a12: one think iteration ----------------------------------------------
    def think_iteration(self, x_t, score_fn, mask, step_size, current_times, sigma_x, sigma_y,
                        state: Optional[OracleState]):
        """langevin_dynamics + run_overdamped, lanpaint.py:192-293."""
        abt, dtx, dty, a_x, a_y, d_x, d_y = self.step_coefficients(current_times, step_size, sigma_x, sigma_y)
        if self.xp.mean_float(dtx) <= 0.0:                                  # :205
            return x_t, state
        a = a_x * (1 - mask) + a_y * mask                                   # :212-214
        d = d_x * (1 - mask) + d_y * mask
        dt = dtx * (1 - mask) + dty * mask

        def coef_c(xt):                                                     # :217-220
            x0 = xt + score_fn(xt)
            c = (abt ** 0.5 * x0 - xt) / (1 - abt) + a * xt
            return c, x0

        if state is None:                                                   # :275-277
            c, x0 = coef_c(x_t)
            x_t = self.ou_step(x_t, dt, a, c, d)
        else:                                                               # :278-284
            c = state.C
            x_t = self.ou_step(x_t, dt / 2, a, c, d)
            c_new, x0 = coef_c(x_t)
            x_t = x_t + (c_new - c) * dt
            x_t = self.ou_step(x_t, dt / 2, a, c, d)      # old C on purpose (:283)
            c = c_new
        self.iterations_run += 1
        return x_t, OracleState(None, c, x0)

    # -- a3-a6, a13: the per-sigma call ---------------------------------------------
    def __call__(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options=None,
                 seed=None, n_steps=None, current_times_audio=None, audio_indicator=None,
                 audio_correction=None):
        """LanPaint.__call__ + LanPaint.LanPaint, lanpaint.py:44-157.  Mutates x."""
        xp = self.xp
        self.ndim = xp.ndim(x)
        if xp.mean_float(xp.abs(noise)) < 1e-8:                              # :51-52
            noise = self._randn(noise)
        if n_steps is None:
            n_steps = self.n_steps
        input_x = x
        ve_sigma, abt, flow_t = (xp.asarray(t) for t in current_times)
        sigma = xp.asarray(sigma)
        replace_sigma = sigma
        if audio_indicator is not None and current_times_audio is not None:   # :68-74
            ve_a, abt_a, flow_a = (xp.asarray(t) for t in current_times_audio)
            ai = audio_indicator
            ve_sigma = ve_sigma * (1 - ai) + ve_a * ai
            abt = abt * (1 - ai) + abt_a * ai
            replace_sigma = sigma * (1 - ai) + flow_a * ai
            current_times = (ve_sigma, abt, flow_t)
        else:
            current_times = (ve_sigma, abt, flow_t)

        step_size = self._bcast(self.step_size * xp.clamp_min(1 - abt, self.min_step_frac))   # :81-82
        x = x * (1 - latent_mask) + self._known_region(replace_sigma, noise, latent_image) * latent_mask   # :94
        flow = self.is_flux or self.is_flow
        abt_b, ve_b = self._bcast(abt), self._bcast(ve_sigma)
        if flow:                                                              # :96-99
            x_t = x * (abt_b ** 0.5 + (1 - abt_b) ** 0.5)
        else:
            x_t = x / (1 + ve_b ** 2) ** 0.5

        stopper = None
        thr, pat = self.early_stop_threshold, self.early_stop_patience
        sink, tags = None, (None, None, None)
        if isinstance(model_options, dict):
            if isinstance(model_options.get("lanpaint_semantic_stop"), dict):                 # earlystop.py:74-95
                ss = model_options["lanpaint_semantic_stop"]
                thr = float(ss.get("threshold", thr))
                pat = int(ss.get("patience", pat))
                if pat > 0 and ss.get("min_steps") is not None:
                    try:
                        ms = int(ss.get("min_steps"))
                    except (TypeError, ValueError):
                        ms = 0
                    if ms > 1:
                        pat = max(pat, ms - 1)
            sink = model_options.get("lanpaint_semantic_trace")
            tags = (model_options.get("bench_case_id"), model_options.get("bench_outer_step"), model_options.get("bench_timestep"))
        if thr > 0.0 and pat > 0:
            stopper = OracleEarlyStopper(thr, pat, latent_mask, xp.mean_float(abt), xp=xp, trace_sink=sink, tags=tags)
            if not stopper.enabled:
                stopper = None
        self.last_stopper = stopper

        state = None
        sigma_x = self._bcast(abt ** 0)                                        # :185-190
        sigma_y = self._bcast(self.beta * abt ** 0)
        for _i in range(n_steps):                                             # :113-142
            score_fn = lambda xt: self.score(xt, latent_image, latent_mask, abt_b, ve_b, self._bcast(flow_t),
                                             model_options, seed, audio_correction)
            prev_state, x_before = state, x_t
            x_t, state = self.think_iteration(x_t, score_fn, latent_mask, step_size, current_times,
                                              sigma_x, sigma_y, state)
            if stopper is not None and stopper.step(x_before, x_t, prev_state, state):
                break

        if flow:                                                              # :144-147
            x = x_t / (abt_b ** 0.5 + (1 - abt_b) ** 0.5)
        else:
            x = x_t * (1 + ve_b ** 2) ** 0.5
        out, _ = self._call_model(x, sigma, model_options, seed)               # :151-153
        out = out * (1 - latent_mask) + latent_image * latent_mask            # :154
        xp.copy_into(input_x, x)                                              # :156
        return out


# ----------------------------------------------------------------------------
# closed-form per-region coefficients (what the HIP table path precomputes);
# float64, used by tests as the known-answer source
# ----------------------------------------------------------------------------
def region_coefficients(abt: float, step: float, lamb: float, beta: float):
    """Returns {region: dict(A, dt, e_full, k_full, std_full, e_half, k_half, std_half)}
    for region 0 (inpaint, "x") and 1 (known, "y").  Follows lanpaint.py:212-214,
    241-252, 315-327 in float64."""
    out = {}
    for r in (0, 1):
        a = (1.0 + lamb * r) / (1.0 - abt)
        dt = step * (beta if r else 1.0)
        ent = {"A": a, "dt": dt}
        for tag, tau in (("full", dt), ("half", dt / 2)):
            e = math.exp(-a * tau)
            k = -math.expm1(-a * tau) / a
            k2 = -math.expm1(-2 * a * tau) / (2 * a)
            ent["e_" + tag], ent["k_" + tag], ent["std_" + tag] = e, k, math.sqrt(max(2.0 * k2, 0.0))
        out[r] = ent
    return out


# ----------------------------------------------------------------------------
# post-decode mask blend: nodes.py:592-647, 1049-1088
# ----------------------------------------------------------------------------
def gaussian_kernel_2d(kernel_size: int) -> np.ndarray:
    """nodes.py:1049-1057: sigma = (size-1)/4, normalised to sum 1; identity for size <= 1."""
    if kernel_size <= 1:
        return np.ones((1, 1), dtype=np.float32)
    sigma = (kernel_size - 1) / 4
    x = np.arange(kernel_size, dtype=np.float32) - (kernel_size // 2)
    xg, yg = np.meshgrid(x, x, indexing="ij")
    k = np.exp(-(xg ** 2 + yg ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (k / k.sum(dtype=np.float32)).astype(np.float32)


def smooth_mask(mask: np.ndarray, k: int) -> np.ndarray:
    """max_pool2d(k, stride 1, pad k//2, -inf padding) then conv2d with the Gaussian (zero padding);
    mask [B, H, W] (nodes.py:625-632, 1080-1087)."""
    m = np.asarray(mask, dtype=np.float32)
    b, h, w = m.shape
    r = k // 2
    pad = np.full((b, h + 2 * r, w + 2 * r), -np.inf, dtype=np.float32)
    pad[:, r:r + h, r:r + w] = m
    d = np.full((b, h, w), -np.inf, dtype=np.float32)
    for dy in range(k):
        for dx in range(k):
            d = np.maximum(d, pad[:, dy:dy + h, dx:dx + w])
    g = gaussian_kernel_2d(k)
    zp = np.zeros((b, h + 2 * r, w + 2 * r), dtype=np.float32)
    zp[:, r:r + h, r:r + w] = d
    out = np.zeros((b, h, w), dtype=np.float32)
    for dy in range(g.shape[0]):
        for dx in range(g.shape[1]):
            out += g[dy, dx] * zp[:, dy:dy + h, dx:dx + w]
    return out


def mask_blend(image1: np.ndarray, image2: np.ndarray, mask: np.ndarray, blend_overlap: int) -> np.ndarray:
    """MaskBlend.blend_images, nodes.py:610-638; images [B, H, W, C]."""
    if image1.shape[1] != image2.shape[1] or image1.shape[2] != image2.shape[2]:
        raise ValueError("Image size mismatch")
    m = smooth_mask(mask, blend_overlap)[..., None]
    return (image1 * (1 - m) + image2 * m).astype(np.float32)


def merge_video_with_mask(orig: np.ndarray, inpainted: np.ndarray, mask: np.ndarray, blend_overlap: int,
                          mask_on: str = "gpu", cpu_fma: bool = True) -> np.ndarray:
    """nodes.py:1060-1088.  `mask_on`: the device the reference holds the mask on (decides torch's nearest-exact index rule
    when a lower-resolution mask is resampled: nearest_exact_src_index)."""
    m = np.asarray(mask, dtype=np.float32)
    if m.ndim == 4:
        m = m[:, 0]
    elif m.ndim == 2:
        m = m[None]
    count = min(orig.shape[0], inpainted.shape[0])
    orig, inpainted = orig[:count], inpainted[:count]
    if m.shape[0] == 1:
        m = np.broadcast_to(m[:1], (count,) + m.shape[1:])
    elif m.shape[0] < count:
        raise ValueError("the mask has fewer frames than the image")
    else:
        m = m[:count]
    if tuple(m.shape[1:]) != tuple(orig.shape[1:3]):
        m = _interp_nearest_exact(m[:, None], orig.shape[1:3], mask_on, cpu_fma)[:, 0]      # nodes.py:1078-1081
    sm = smooth_mask(np.ascontiguousarray(m), blend_overlap)[..., None]
    return (orig * (1 - sm) + inpainted * sm).astype(np.float32)


def pack_mask_bits(mask, denoise_mask=False):
    """Bit-packed mask layout of include/lanpaint_hip.h (LP_FL_MASK_BITS; SURVEY 8b `mask_kind`): element i is bit
    (i & 31) of little-endian 32-bit word (i >> 5), padded with zero bits to whole 64-bit words.  Bit = latent_mask
    value (1 = known): `v > 0.5` of a binary latent mask, or `not (v > 0.5)` of ComfyUI's denoise_mask
    (the reference's `1 - (denoise_mask > 0.5)`, nodes.py:281-283).  Returns uint8[((n + 63) // 64) * 8]."""
    v = np.asarray(mask, dtype=np.float32).reshape(-1)
    bit = ~(v > 0.5) if denoise_mask else (v > 0.5)
    out = np.zeros(((v.size + 63) // 64) * 8, dtype=np.uint8)
    packed = np.packbits(bit.astype(np.uint8), bitorder="little")
    out[:packed.size] = packed
    return out


def unpack_mask_bits(bits, n_el):
    """Inverse of pack_mask_bits: float32 latent mask of n_el elements."""
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little")[:n_el].astype(np.float32)
