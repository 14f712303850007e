"""The engine core: construction, the reference's helper methods, and ONE sigma call as launches of the fused HIP kernels.

    lp_step REPLACE|EMIT|COEFFS       replace step, VP rescale, first model input; the same launch builds the per-row coefficient
                                      table on the device (no host sync)                                   (`_prologue`)
    for i in range(n):   model(x_in)  -> (x0, x0_BIG)
        lp_step POST|PRE_HALF|EMIT    post-model half of iteration i fused with the pre-model half of i+1   (`_think_and_final_model`)
    model(x) ; lp_finalize            known-region reprojection + in-place write-back of x                  (`_epilogue`)

Every per-element operation is inside those launches (include/lanpaint_hip.h); this file fills descriptors with raw device
pointers and calls the backbone.  Capture / replay of whole calls: capture.py.  Loops off the fast path: loops.py."""
from __future__ import annotations

import ctypes
import os
import weakref
from collections import OrderedDict
from time import perf_counter

import torch

from . import _cabi
from ._cabi import (LP_FL_ES, LP_FL_ES_GATED, LP_FL_ES_CLOSE, LP_FL_CFG_FUSED, LP_FL_FLOW, LP_FL_MASK_BITS, LP_FL_MASK_U8, LP_FL_XIN_BF16,
                    LP_FL_XIN_F16, LP_FL_PER_ELEMENT, LP_FL_WRITE_X0S, LP_FL_X0_BF16, LP_FL_X0_F16, LP_FL_X0S_GIVEN, LP_PH_EMIT,
                    LP_PH_POST_FIRST, LP_PH_POST_STEADY, LP_PH_PRE_HALF, LP_PH_REPLACE, LP_REPLACE_FLOW, LP_REPLACE_KNOWN,
                    LP_REPLACE_VE)
from ._util import _as_f32c, _noise_scaling_kind, aten_randn_policy, raw_stream, tensor_version
from .buffers import _CallState, _Workspace
from .earlystop import HostStopper, StopOptions
from .masks import _compact_mask, pack_indicator, pack_mask, refresh_packed_mask
from .types import FusedCFGHeads


class EngineCore:
    """Base of LanPaint (lanpaint.py)."""

    # ------------------------------------------------------------------ construction
    def __init__(self, Model, NSteps, Friction, Lambda, Beta, StepSize, IS_FLUX=False, IS_FLOW=False,
                 EarlyStopThreshold=0.0, EarlyStopPatience=1, EarlyStopHook=None, MinStepFrac=0.0,
                 *, rng=None, philox_seed=None, graph=None, model_dtype=None, early_stop_group=None):
        """Positional signature == reference lanpaint.py:8.  Keyword-only extras:
        rng: "torch" (default; xi = torch.randn_like in the reference's draw order, so a
             seeded run consumes the device generator exactly like the reference),
             "philox" (xi generated inside the fused kernel, nothing read from HBM),
             or a callable `rng(like) -> Tensor` (tests feed recorded streams).
             Env LANPAINT_AMD_RNG overrides the default.
        philox_seed: Philox key; defaults to the `seed` argument of each call.
        graph: capture each sigma call (replace, N x [backbone, fused step], final backbone call,
             finalise) into ONE hipGraph and replay it (the loop is launch bound at image-latent
             sizes).  True needs a capturable backbone (static shapes, no host sync); rng
             "torch"/"philox" only; ignored (eager launches) when per-element times or method
             overrides are in play.  Default (None; env LANPAINT_AMD_GRAPH=1 / 0 forces it on / off):
             "auto" -- the first call of a job runs eagerly and times the backbone on the host; when
             that is cheap to enqueue (< AUTO_MAX_BACKBONE_HOST_US per call: a launch-bound loop, the
             case a graph helps) the second call with the same latent_image / mask / model_options
             objects is captured, the capture is CHECKED against an eager run of the same call
             (rng="torch" only: bitwise, from the same generator state.  rng="philox" replays draw from a
             device-side launch counter, eager launches from a host-side one, so the two streams differ
             by construction and a philox capture is used unchecked -- pass graph=False to keep a stateful
             backbone eager).  The check costs one warm-up, one replay and one eager run of the call on
             clones of x: about 3 x (n_steps + 1) extra calls into the model on that one sigma call, which
             a call-counting backbone will see.  A backbone that cannot be captured
             (host sync inside it), that draws from torch's generator, or whose replay differs from
             eager keeps the engine eager for good, with a warning.  Expensive backbones are never
             captured: the Langevin launches are noise next to them.
        model_dtype: torch.bfloat16 / torch.float16 -> the latent handed to the backbone inside the
             think loop is emitted in that dtype by the kernel (no separate cast pass); the state,
             the written-back x and the arithmetic stay fp32.
        early_stop_group: True / a ProcessGroup when ONE batch is sharded over ranks and the inner early stop must
             take the single-process decision (its metric is defined over the whole batch, earlystop.py:52-55):
             the partial sums are all-reduced each iteration (host-side stopper)."""
        self.n_steps = NSteps
        self.chara_lamb = Lambda
        self.IS_FLUX = IS_FLUX
        self.IS_FLOW = IS_FLOW
        self.step_size = StepSize
        self.inner_model = Model
        self.friction = Friction
        self.chara_beta = Beta
        self.min_step_frac = MinStepFrac
        self.img_dim_size = None
        self.early_stop_threshold = EarlyStopThreshold
        self.early_stop_patience = EarlyStopPatience
        self.early_stop_hook = EarlyStopHook

        self.rng = rng if rng is not None else os.environ.get("LANPAINT_AMD_RNG", "torch")
        if not callable(self.rng) and self.rng not in ("torch", "torch-eager", "philox"):
            raise ValueError(f"rng must be 'torch', 'torch-eager', 'philox' or a callable, got {self.rng!r}")
        self._torch_consumed = 0                 # generator offset this engine advanced itself (LP_RNG_TORCH)
        self._graph_blocked = False              # the backbone draws from torch's generator inside the loop
        self.philox_seed = philox_seed
        self._philox_offset = 0
        if graph is None:
            graph = {"1": True, "0": False}.get(os.environ.get("LANPAINT_AMD_GRAPH", "auto"), "auto")
        self.graph = "auto" if graph == "auto" else bool(graph)
        self._auto = None                        # auto mode: [signature of the job, eager calls seen, backbone host s per call]
        if model_dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"model_dtype must be None, float32, bfloat16 or float16, got {model_dtype}")
        self.model_dtype = None if model_dtype == torch.float32 else model_dtype
        self._graphs = OrderedDict()             # key -> _CapturedCall, LRU-bounded (MAX_GRAPHS)
        self._static_ws = {}                     # (shape, device, model dtype) -> workspace shared by the captures of that shape
        self._last_cap = None                    # the capture the previous call replayed (identity pre-check)
        self._es_close = False                   # capture in progress: its loop closes itself (no trace requested)
        self._rng_counters = {}                  # device -> u64 counter read by captured Philox launches
        self._capturing = None                   # device u64 Philox counter while capturing
        self._cap_offset = 0
        self._lib = _cabi.load()                 # raises if the HIP extension is not built
        self._ws = None
        self._desc = _cabi.LpStepDesc()
        self._fdesc = _cabi.LpFinalDesc()
        self._hyper = _cabi.LpHyper()
        self._noise_check = None                 # (weakref(noise), version, verdict)
        self.assume_static_noise = False         # see _noise_is_zero
        self.auto_pack_mask = os.environ.get("LANPAINT_AMD_AUTO_PACK", "1") != "0"      # see _auto_pack
        self._mask_seen = None
        self._noise_regenerated = False
        self._iterations_run = 0                 # think iterations executed (it/s accounting)
        self.last_inner_steps = 0
        self.early_stop_group = early_stop_group
        self._ds = None                          # _DeviceStop: buffers of the early stop evaluated on the device
        self._es_opts = None                     # host-side early-stop options of the call in flight
        self._es_pending = None                  # a replayed loop whose iteration count the device has yet to report

    @property
    def iterations_run(self):
        """Think iterations executed so far.  A replayed loop with the inner early stop decides its length on the
        device; reading the count waits for that report."""
        self._es_resolve()
        return self._iterations_run

    @iterations_run.setter
    def iterations_run(self, v):
        self._iterations_run = v

    def _indicator_pack(self, indicator, shape):
        """pack_indicator once per sigma call (eligibility check, graph key and prologue all ask; an inference tensor would be
        re-packed -- a launch and a host read -- each time)."""
        c = getattr(self, "_av_pack", None)
        if c is None or c[0] is not indicator or c[1] != tuple(shape):
            c = self._av_pack = (indicator, tuple(shape), pack_indicator(indicator, shape))
        return c[2]

    def _auto_pack(self, latent_mask, x):
        """A caller that hands the engine a plain fp32 mask (the reference's interface) still gets the hard-mask kernels when
        the mask IS binary: the second consecutive call with the same mask tensor packs it (one launch + ONE host read of
        the "values other than 0 and 1" flag per mask tensor; a mask that turns out soft is remembered as such).  The first
        call never packs -- a caller that builds a fresh mask tensor per call would pay the host read every time."""
        # (not under graph=True: that engine captures on the FIRST call, with the mask as it came; packing on the second would
        # only make a second capture -- a caller who forces graphs packs the mask himself)
        if not self.auto_pack_mask or self.graph is True or not latent_mask.is_cuda or latent_mask.dtype != torch.float32 \
                or not latent_mask.is_contiguous() or latent_mask.shape != x.shape or latent_mask.numel() == 0 \
                or getattr(latent_mask, "_lp_u8", None) is not None:
            return
        ver = (tensor_version(latent_mask), latent_mask.data_ptr())
        if ver[0] == -1:                 # no version counter (inference mode): a later in-place edit to soft values could not be seen
            return
        seen = self._mask_seen
        if seen is None or seen[0]() is not latent_mask or seen[1] != ver:
            self._mask_seen = (weakref.ref(latent_mask), ver, False)
            return
        if seen[2]:                      # known to be soft
            return
        try:
            pack_mask(latent_mask, check=True)
            latent_mask._lp_auto = True
        except ValueError:
            self._mask_seen = (seen[0], seen[1], True)
        except Exception:
            self.auto_pack_mask = False

    def rng_position(self, device):
        """Where the engine's own noise streams stand (checkers reproduce the draws of the next sigma call from this):
        rng="philox": (device-side launch counter the replayed launches of this engine add to their sequence numbers -- one
        host read --, host-side count of eager launches; an eager launch k draws with sequence number 2^48 + k);
        rng="torch": (offset of the device generator, its seed)."""
        if self.rng == "philox":
            st = self._rng_counters.get(device)
            return (int(st[0].item()) if st is not None else 0, int(self._philox_offset))
        gen = self._generator(device)
        return (int(gen.get_offset()), int(gen.initial_seed()))

    # ------------------------------------------------------------------ inner early stop, host side
    def _es_options(self, model_options):
        """The inner early stop's options for this call (earlystop.StopOptions: the reference's `lanpaint_semantic_stop` /
        `lanpaint_semantic_trace` contract), plus where the verdict is formed.  None = off."""
        o = StopOptions.parse(model_options, self.early_stop_threshold, self.early_stop_patience, self.early_stop_hook)
        if o is None:
            return None
        return {"threshold": o.threshold, "patience_eff": o.patience_eff, "distance_fn": o.distance_fn, "trace": o.trace,
                "tags": o.tags, "device": not callable(o.distance_fn) and self.early_stop_group is None, "parsed": o}

    # ------------------------------------------------------------------ reference helpers
    def add_none_dims(self, array):
        """lanpaint.py:23-29."""
        while array.ndim < self.img_dim_size:
            array = array.unsqueeze(array.ndim)
        return array

    def remove_none_dims(self, array):
        """lanpaint.py:30-33."""
        return array[(slice(None),) + (0,) * (self.img_dim_size - 1)]

    def unpack_model_output(self, output):
        """lanpaint.py:34-43 (a FusedCFGHeads is a lazy (x0, x0_BIG) pair)."""
        if isinstance(output, FusedCFGHeads):
            return output.materialize()
        if isinstance(output, (tuple, list)):
            if len(output) >= 2:
                return output[0], output[1]
            if len(output) == 1:
                return output[0], output[0]
            raise ValueError("Model output is empty")
        return output, output

    def sigma_x(self, abt):
        """lanpaint.py:185-187."""
        return abt ** 0

    def sigma_y(self, abt):
        """lanpaint.py:188-190."""
        return self.chara_beta * abt ** 0

    def prepare_step_size(self, current_times, step_size, sigma_x, sigma_y):
        """lanpaint.py:295-328, host tensors; kept for API parity (the kernels take
        the same quantities from the lp_coeffs table)."""
        sigma, abt, _flow_t = current_times
        sigma, abt = self.add_none_dims(sigma), self.add_none_dims(abt)
        dtx, dty = 2 * step_size * sigma_x, 2 * step_size * sigma_y
        gam_x = self.friction ** 2 * self.step_size * sigma_x / 0.1 * sigma ** 0 / 2.0
        gam_y = self.friction ** 2 * self.step_size * sigma_y / 0.1 * sigma ** 0 / 2.0
        a_t_x = 1 / (1 - abt) * dtx / 2
        a_t_y = (1 + self.chara_lamb) / (1 - abt) * dty / 2
        a_x, a_y = a_t_x / (dtx / 2), a_t_y / (dty / 2)
        d = (2 * abt ** 0) ** 0.5
        return sigma, abt, dtx / 2, dty / 2, gam_x / (dtx / 2), gam_y / (dty / 2), a_x, a_y, d, d

    def score_model(self, x_t, y, mask, abt, sigma, tflow, model_options, seed):
        """lanpaint.py:159-184 as host tensor ops: the public/compat entry.  The fused
        loop never calls this; it exists so code written against the reference
        (and overrides of it) keeps working."""
        lamb = self.chara_lamb
        if self.IS_FLUX or self.IS_FLOW:
            x = x_t / (abt ** 0.5 + (1 - abt) ** 0.5)
            t = self.remove_none_dims(tflow)
        else:
            x = x_t * (1 + sigma ** 2) ** 0.5
            t = self.remove_none_dims(sigma)
        x_0, x_0_big = self.unpack_model_output(self.inner_model(x, t, model_options=model_options, seed=seed))
        corr = getattr(self, "audio_correction", None)
        if corr is not None:
            x_0 = x + corr * (x_0 - x)
            x_0_big = x + corr * (x_0_big - x)
        score_x = -(x_t - x_0)
        score_y = -(1 + lamb) * (x_t - y) + lamb * (x_t - x_0_big)
        return score_x * (1 - mask) + score_y * mask

    # ------------------------------------------------------------------ plumbing
    def _overridden(self, name):
        return name in self.__dict__ or getattr(type(self), name) is not self._OWN_METHODS[self._OWN_NAMES.index(name)]

    def _stream(self, device):
        return raw_stream(device)

    def _noise_is_zero(self, noise):
        """lanpaint.py:51: `mean|noise| < 1e-8` costs the reference one host sync per sigma; the verdict is
        cached per tensor OBJECT and version (a weak reference, not the address: the caching allocator
        recycles addresses), so it is paid once per sampling run."""
        c = self._noise_check
        ver = (tensor_version(noise), noise.data_ptr())
        # a tensor without a version counter (inference mode) could have been rewritten in place unnoticed: its verdict is
        # only kept when the caller vouches for the run's noise (`assume_static_noise`: KSAMPLER.sample builds the engine for ONE
        # run, whose noise tensor ComfyUI creates once and never touches); otherwise it is re-read every call like the reference
        stale = ver[0] == -1 and not self.assume_static_noise
        if c is None or c[0]() is not noise or c[1] != ver or stale:
            self._noise_check = c = (weakref.ref(noise), ver, bool(torch.mean(torch.abs(noise)) < 1e-8))
        return c[2]

    def _draw(self, like):
        """One N(0,1) tensor in the reference's draw order (lanpaint.py:252), or None
        when the kernel generates it (Philox)."""
        if self.rng == "philox":
            return None
        xi = self.rng(like) if callable(self.rng) else torch.randn_like(like)
        return _as_f32c(xi)

    # ---- rng="torch": the device generator's randn stream, produced inside the step kernel -------------------
    @staticmethod
    def _generator(device):
        torch.cuda.init()
        return torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]

    _policy_cache = {}

    @classmethod
    def _randn_policy(cls, device, numel):
        """(block * grid, generator-offset increment) of ATen's launch for one randn of `numel` fp32 elements
        (calc_execution_policy in native/cuda/DistributionTemplates.h: 256-thread blocks, grid capped at
        SMs * (maxThreadsPerSM / 256), unroll 4)."""
        key = (device.index, numel)
        hit = cls._policy_cache.get(key)
        if hit is None:
            p = torch.cuda.get_device_properties(device)
            hit = cls._policy_cache[key] = aten_randn_policy(numel, p.multi_processor_count,
                                                             p.max_threads_per_multi_processor)
        return hit

    _torch_stream_ok = {}

    @classmethod
    def _check_torch_stream(cls, device):
        """One-time self-check per device: LP_RNG_TORCH claims to reproduce torch.randn bit for bit, which rests on
        ATen's launch policy, rocRAND's Box-Muller and the fp-contraction mode of both builds.  Compare the kernel's
        generator (lp_torch_normal) with torch.randn on one small tensor (one ATen thread per element) and one past
        the grid cap (several elements per thread); the device generator is put back where it was.  False -> the
        engine falls back to explicit torch.randn_like draws ("torch-eager": same values, separate launches)."""
        ok = cls._torch_stream_ok.get(device.index)
        if ok is None:
            lib = _cabi.load()
            state = torch.cuda.get_rng_state(device)
            gen = cls._generator(device)
            ok = True
            try:
                with torch.cuda.device(device):
                    for n in (4099, 1 << 20):
                        off, seed = gen.get_offset(), gen.initial_seed()
                        ref = torch.randn(n, device=device)
                        mine = torch.empty_like(ref)
                        bg, _inc = cls._randn_policy(device, n)
                        _cabi.check(lib.lp_torch_normal(mine.data_ptr(), n, seed, off, bg,
                                                        torch.cuda.current_stream(device).cuda_stream), "lp_torch_normal")
                        ok = ok and bool(torch.equal(ref, mine))
            finally:
                torch.cuda.set_rng_state(state, device)
            cls._torch_stream_ok[device.index] = ok
            if not ok:
                import warnings
                warnings.warn("lanpaint_amd: the in-kernel reproduction of torch.randn's stream does not match this torch / "
                              "ROCm build; rng='torch' falls back to explicit torch.randn_like draws (rng='torch-eager')")
        return ok

    def _fill_hyper(self, flow):
        h = self._hyper
        h.lambda_, h.beta, h.step_size = float(self.chara_lamb), float(self.chara_beta), float(self.step_size)
        h.min_step_frac, h.is_flow = float(self.min_step_frac), int(bool(flow))
        h.one_plus_lambda = 1.0 + float(self.chara_lamb)     # double sum, then fp32 (ctypes c_float)
        return h

    def _workspace(self, like):
        if self._ws is None or not self._ws.matches(like):
            self._ws = _Workspace(like)
        return self._ws

    def _launch_step(self, stream):
        _cabi.check(self._lib.lp_step(ctypes.byref(self._desc), stream), "lp_step")

    def _set_model_output(self, d, output, base_flags, shape):
        """Backbone output -> descriptor: a FusedCFGHeads keeps the CFG combination inside the kernel."""
        if isinstance(output, FusedCFGHeads) and output._heads is None and not (base_flags & LP_FL_PER_ELEMENT) \
                and output.cond.dtype == output.uncond.dtype:
            d.cfg_scale, d.cfg_scale_big = output.scale, output.scale_big
            return self._set_model_heads(d, output.cond, output.uncond, base_flags | LP_FL_CFG_FUSED, shape)
        heads = self.unpack_model_output(output)
        return self._set_model_heads(d, heads[0], heads[1], base_flags, shape)

    def _set_model_heads(self, d, x0, x0_big, base_flags, shape):
        """Point the descriptor at the backbone outputs (fp32/bf16/fp16, made dense)."""
        if x0.shape != shape:
            x0 = x0.expand(shape)
        if x0_big.shape != shape:
            x0_big = x0_big.expand(shape)
        same = x0_big is x0
        if x0.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x0 = x0.float()
        if not x0.is_contiguous():
            x0 = x0.contiguous()
        if same:
            x0_big = x0
        else:
            if x0_big.dtype != x0.dtype:
                x0_big = x0_big.to(x0.dtype)
            if not x0_big.is_contiguous():
                x0_big = x0_big.contiguous()
        fl = base_flags
        if x0.dtype == torch.bfloat16:
            fl |= LP_FL_X0_BF16
        elif x0.dtype == torch.float16:
            fl |= LP_FL_X0_F16
        d.flags = fl
        d.x0, d.x0_big = x0.data_ptr(), x0_big.data_ptr()
        return x0, x0_big          # keep alive until the launch is enqueued

    # ------------------------------------------------------------------ entry points
    def __call__(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, n_steps=None,
                 current_times_audio=None, audio_indicator=None, audio_correction=None):
        """lanpaint.py:44-55."""
        if not x.is_cuda:
            raise RuntimeError("lanpaint_amd.LanPaint runs on a HIP device only (got a %s tensor); "
                               "there is no CPU fallback" % x.device.type)
        if self.rng == "torch" and not self._check_torch_stream(x.device):
            self.rng = "torch-eager"
        # a bit-packed copy made from THIS tensor (pack_mask(latent_mask)) follows it: re-packed in place when the tensor was
        # rewritten since, or -- no version counter (inference mode) -- on every call; masks derived from another tensor
        # (KSamplerX0Inpaint's, from ComfyUI's denoise_mask) are kept current by whoever derived them
        rec = getattr(latent_mask, "_lp_bits_of", None)
        if rec is not None and not rec[2] and rec[0]() is latent_mask:
            if getattr(latent_mask, "_lp_auto", False) and tensor_version(latent_mask) != rec[1]:
                # a mask the ENGINE packed (nobody vouched for it being binary) was rewritten: forget the copy and look again
                for a in ("_lp_bits", "_lp_bits_of", "_lp_auto"):
                    delattr(latent_mask, a)
                self._auto_pack(latent_mask, x)
            else:
                refresh_packed_mask(latent_mask)
        elif rec is None:
            self._auto_pack(latent_mask, x)
        self._es_opts = self._es_options(model_options)
        if self._es_pending is not None and (self._es_opts is None or self._es_opts["trace"] is not None
                                             or self._es_pending[0] is not self._ds):
            self._es_resolve()       # (a loop that only has to be counted is collected when somebody asks)
        self.img_dim_size = len(x.shape)
        self.latent_image = latent_image
        self.noise = noise
        self.audio_indicator = audio_indicator
        self.current_times_audio = current_times_audio
        self.audio_correction = audio_correction
        self._av_pack = None                     # (indicator, shape, pack_indicator's answer) of THIS call, see _indicator_pack
        self._noise_regenerated = self._noise_is_zero(noise)
        if self._noise_regenerated:              # lanpaint.py:51-52: the first draw of the call
            self.noise = self.rng(noise) if callable(self.rng) else torch.randn_like(noise)
        if n_steps is None:
            n_steps = self.n_steps
        cap = self._last_cap
        if cap is not None and self._same_call(cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed):
            self._iterations_run += cap.ran           # same tensors / shapes / options as the previous call:
            self.last_inner_steps = cap.ran          # skip the key construction, go straight to the replay
            return self._replay_fast(cap, x, sigma, current_times)
        graphed = self._graph_eligible(x, model_options, sigma, current_times)
        if graphed and self.graph == "auto":
            graphed = self._auto_ready(x, latent_mask, model_options)
        run = self._call_graphed if graphed else self.LanPaint
        if x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return run(x, sigma, latent_mask, current_times, n_steps, model_options, seed, self.IS_FLUX,
                           self.IS_FLOW)
        return run(x, sigma, latent_mask, current_times, n_steps, model_options, seed, self.IS_FLUX, self.IS_FLOW)

    def LanPaint(self, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """lanpaint.py:56-157: prologue (coefficients + replace step), think loop + final backbone call,
        epilogue (reprojection + in-place write-back)."""
        if x.numel() == 0:               # empty batch: only the model-call structure of the reference remains
            for _ in range(n_steps if float(self.step_size) > 0.0 else 0):
                self.inner_model(x, sigma, model_options=model_options, seed=seed)
            out, _ = self.unpack_model_output(self.inner_model(x, sigma, model_options=model_options, seed=seed))
            return out
        st = self._prologue(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
        final = self._think_and_final_model(st, model_options, seed)
        return self._epilogue(st, final)

    # ---- prologue: per-call descriptor, coefficient table, replace step ------------------------------------
    def _prologue(self, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW, ws=None, ds=None,
                  out=None, defer_launch=False):
        """`out`: the tensor the call returns, when the caller has it already.  `defer_launch`: build the descriptor of the
        replace launch (st.k0_desc) but do not enqueue it -- the capture enqueues it as the first node of the graph."""
        lib, d = self._lib, self._desc
        st = _CallState()
        st.input_x = x
        st.flow = flow = bool(IS_FLUX or IS_FLOW)
        st.xc = xc = _as_f32c(x)
        st.shape, st.n_el, st.rows = shape, n_el, rows = xc.shape, xc.numel(), xc.shape[0]
        st.ws = ws = ws if ws is not None else self._workspace(xc)
        if ws.static_io and (xc.data_ptr() & 15):
            # a captured lp_finalize writes x through the I/O table with the vector width fixed at capture time
            st.xc = xc = xc.clone()
        st.stream = stream = self._stream(xc.device)
        st.sigma = sigma
        st.y = y = _as_f32c(self.latent_image if self.latent_image.shape == shape else self.latent_image.expand(shape))
        nz = _as_f32c(self.noise if self.noise.shape == shape else self.noise.expand(shape))
        m = latent_mask if latent_mask.shape == shape else latent_mask.expand(shape)
        st.m = m = _as_f32c(m)
        # a caller that KNOWS the mask is binary (KSamplerX0Inpaint builds it as 1 - (dm > 0.5); `pack_mask`)
        # may attach a compact copy: the kernels then read 1 bit / 1 byte instead of 4 bytes per element
        st.m_c, st.m_flag = m_c, m_flag = _compact_mask(latent_mask, shape, xc.device)

        VE_Sigma, abt, Flow_t = current_times
        replace_sigma = sigma
        per_el = False
        av = None                 # AV pack on the two-row table (LP_FL_AV): (bits, audio share, the four [2 rows] time arrays)
        if self.audio_indicator is not None and self.current_times_audio is not None:     # lanpaint.py:68-74
            VE_a, abt_a, Flow_a = self.current_times_audio
            ai = self.audio_indicator
            row_sized = all(t.numel() in (1, rows) for t in (VE_Sigma, abt, sigma, Flow_t, VE_a, abt_a, Flow_a))
            host_side = self._overridden("langevin_dynamics") or self._overridden("score_model") or \
                self._overridden("prepare_step_size") or (self._es_opts is not None and not self._es_opts["device"])
            packed = self._indicator_pack(ai, shape) if (row_sized and not host_side and
                                                         os.environ.get("LANPAINT_AMD_AV_TABLE", "1") != "0") else None
            if packed is not None and self._es_opts is not None and not packed[2]:
                packed = None        # rows with different audio shares: the device-side stopper's one `av_frac` would not give the
                                     # reference's threshold (mean of the blended abt) -> reference-shaped path, host stopper
            if packed is not None:
                # a 0/1 indicator and per-row times: the blend picks, per element, one of two per-row time sets exactly
                # (x * 1 + y * 0 = x), so the kernels take the sets from a two-row table and the indicator as bits -- no
                # full-size time tensors, no per-element transcendentals (the reference-shaped path below remains for anything else)
                if ws.av_times is None or ws.av_times.shape[1] != 2 * rows:
                    ws.av_times = torch.empty((4, 2 * rows), dtype=torch.float32, device=xc.device)
                    ws.coef_av = torch.empty((2 * rows, _cabi.LP_COEF_STRIDE), dtype=torch.float32, device=xc.device)
                tv = ws.av_times.view(4, rows, 2)
                t_mod = Flow_t if flow else VE_Sigma
                for k, (v, a) in enumerate(((VE_Sigma, VE_a), (abt, abt_a), (sigma, Flow_a), (t_mod, t_mod))):
                    tv[k, :, 0] = v.reshape(-1)
                    tv[k, :, 1] = a.reshape(-1)
                av = (packed[0], packed[1])
            else:
                VE_Sigma = VE_Sigma * (1 - ai) + VE_a * ai
                abt = abt * (1 - ai) + abt_a * ai
                replace_sigma = sigma * (1 - ai) + Flow_a * ai
                current_times = (VE_Sigma, abt, Flow_t)
        if av is None and (abt.numel() not in (1, rows) or VE_Sigma.numel() not in (1, rows) or replace_sigma.numel() not in (1, rows)):
            per_el = True
        st.abt, st.current_times = abt, current_times
        t_src = Flow_t if flow else current_times[0]
        # (a [B] tensor goes through add_none_dims / remove_none_dims unchanged: six view ops the host can skip)
        t_model = t_src if t_src.ndim == 1 else self.remove_none_dims(self.add_none_dims(t_src))

        # ---- per-call descriptor --------------------------------------------------
        st.base_flags = base_flags = (LP_FL_FLOW if flow else 0) | m_flag
        hyp = self._fill_hyper(flow)
        d.n_el, d.el_per_row, d.rows = n_el, n_el // rows, rows
        d.lambda_, d.one_plus_lambda, d.beta = hyp.lambda_, hyp.one_plus_lambda, hyp.beta
        d.step_size, d.min_step_frac = hyp.step_size, hyp.min_step_frac
        d.y, d.x_t, d.C = y.data_ptr(), ws.x_t.data_ptr(), ws.C.data_ptr()
        d.mask = m_c.data_ptr() if m_c is not None else m.data_ptr()
        d.x0s = None
        d.abt_el = d.ve_el = d.rsig_el = d.corr_el = None
        d.av_bits, d.av_frac = None, 0.0
        keep = st.keep = [nz]          # tensors that must outlive the enqueued launches of this call
        corr = self.audio_correction
        if per_el:
            st.base_flags = base_flags = base_flags | LP_FL_PER_ELEMENT
            abt_el = _as_f32c(self.add_none_dims(abt).expand(shape))
            ve_el = _as_f32c(self.add_none_dims(VE_Sigma).expand(shape))
            rs_el = _as_f32c(self.add_none_dims(replace_sigma).expand(shape))
            keep += [abt_el, ve_el, rs_el]
            d.abt_el, d.ve_el, d.rsig_el = abt_el.data_ptr(), ve_el.data_ptr(), rs_el.data_ptr()
            d.coef = None
        elif av is not None:
            # two time sets per row: the table (2 rows per batch row) comes from lp_coeffs on the interleaved inputs; the replace
            # launch does not rebuild it (no LP_PH_COEFFS) and every launch of the call carries the indicator bits
            st.base_flags = base_flags = base_flags | _cabi.LP_FL_AV
            tv = ws.av_times
            _cabi.check(lib.lp_coeffs(ctypes.byref(hyp), tv[0].data_ptr(), 1, tv[1].data_ptr(), 1, tv[2].data_ptr(), 1, None, 0,
                                      tv[3].data_ptr(), 1, 2 * rows, ws.coef_av.data_ptr(), stream), "lp_coeffs")
            d.coef, d.coef_out = ws.coef_av.data_ptr(), None
            d.av_bits, d.av_frac = av[0].data_ptr(), float(av[1])
            keep.append(av[0])
        else:
            flat = lambda t: _as_f32c(t if t.ndim == 1 else t.reshape(-1))       # noqa: E731
            ve_r, abt_r, rs_r, tm_r = flat(VE_Sigma), flat(abt), flat(replace_sigma), flat(t_model)
            keep += [ve_r, abt_r, rs_r, tm_r]
            # the coefficient table is written by the replace launch itself (LP_PH_COEFFS: lp_coeffs folded in)
            d.t_ve, d.t_abt, d.t_rsig, d.t_model = ve_r.data_ptr(), abt_r.data_ptr(), rs_r.data_ptr(), tm_r.data_ptr()
            d.t_ve_stride, d.t_abt_stride = ve_r.numel() > 1, abt_r.numel() > 1
            d.t_rsig_stride, d.t_model_stride = rs_r.numel() > 1, tm_r.numel() > 1
            d.coef = d.coef_out = ws.coef.data_ptr()
        if corr is not None:
            corr_el = _as_f32c(corr if corr.shape == shape else corr.expand(shape))
            if ws.static_io:          # a replayed loop bakes the address: the call's correction goes through a workspace buffer
                if getattr(ws, "corr", None) is None:
                    ws.corr = torch.empty_like(ws.x_t)
                ws.corr.copy_(corr_el)
                corr_el = ws.corr
            keep.append(corr_el)
            d.corr_el = corr_el.data_ptr()
        if ws.static_io and not per_el:
            # replayed loop: the backbone reads its time / sigma from the table the prologue just refreshed
            n_t, n_s = (rows if t_model.numel() > 1 else 1), (rows if sigma.numel() > 1 else 1)
            table = ws.coef if av is None else ws.coef_av.view(rows, 2 * _cabi.LP_COEF_STRIDE)     # (AV: the video row of each pair)
            st.t_model, st.sigma_model = table[:n_t, _cabi.LP_C_TMODEL], table[:n_s, _cabi.LP_C_RSIGMA]
        else:
            st.t_model, st.sigma_model = t_model, sigma

        # ---- replace-step source (lanpaint.py:84-94) --------------------------------
        ms = self.inner_model.inner_model.model_sampling
        d.noise_scale = 1.0
        d.known = None
        d.noise = nz.data_ptr()
        if replace_sigma.numel() == 1 and av is None:
            kind, ns = _noise_scaling_kind(ms)
            if kind == "callback":
                known = _as_f32c(ms.noise_scaling(self.add_none_dims(replace_sigma), nz, y))
                keep.append(known)
                d.replace_kind, d.known = LP_REPLACE_KNOWN, known.data_ptr()
            elif kind == "ve":
                d.replace_kind = LP_REPLACE_VE
            else:
                d.replace_kind, d.noise_scale = LP_REPLACE_FLOW, ns
        else:        # per-row sigma: the reference emulates the FLOW form elementwise, noise_scale included (lanpaint.py:89-92)
            d.replace_kind, d.noise_scale = LP_REPLACE_FLOW, float(getattr(ms, "noise_scale", 1.0))

        st.compat = self._overridden("langevin_dynamics") or self._overridden("score_model") or \
            self._overridden("prepare_step_size")
        if n_steps > 0 and float(self.step_size) <= 0.0 and not st.compat:
            n_steps = 0          # dtx <= 0: every iteration returns immediately (lanpaint.py:205)
        st.n_steps = n_steps

        # model-space buffers handed to the backbone.  Eager: fresh per call so the tensor the final model call
        # saw stays valid after we return.  Replay: owned by the captured call's workspace.  x_final (fp32) is
        # the x that is written back in place; with a half-precision model_dtype the in-loop emits go to a
        # separate buffer in that dtype.
        if ws.static_io:
            st.x_final, st.x_in = ws.x_final, ws.x_in
        else:
            st.x_final = torch.empty_like(xc)
            st.x_in = st.x_final if self.model_dtype is None else torch.empty_like(xc, dtype=self.model_dtype)
        st.xin_flag = 0 if self.model_dtype is None else (LP_FL_XIN_BF16 if self.model_dtype == torch.bfloat16 else LP_FL_XIN_F16)

        d.x = xc.data_ptr()
        d.xi_post = d.xi_pre = None
        d.rng_offset_ptr = None
        d.rng_seed = int(self.philox_seed if self.philox_seed is not None else (seed or 0)) & 0xFFFFFFFFFFFFFFFF
        d.rng_state_out = None
        if ws.static_io and self.rng == "torch" and not per_el:
            # a replayed loop takes its offsets relative to the generator state this launch publishes
            gen = self._generator(xc.device)
            d.rng_state_out = self._rng_state(xc.device).data_ptr()
            d.rng_state_val[0], d.rng_state_val[1] = gen.get_offset(), gen.initial_seed()
        d.flags = base_flags | self._emit(st, n_steps == 0)
        d.phases = LP_PH_REPLACE | LP_PH_EMIT | (0 if (per_el or av is not None) else _cabi.LP_PH_COEFFS)
        st.out = None
        if ws.static_io and not per_el:
            # a replayed call: its lp_finalize may be a node of the graph; this launch tells it where x and out live
            st.out = out if out is not None else torch.empty_like(xc)
            d.io_table_out, d.io_valid = self._rng_state(xc.device).data_ptr() + 16, 1     # (word 2: "this call is valid")
            d.io_table_val[0], d.io_table_val[1] = xc.data_ptr(), st.out.data_ptr()
        # inner early stop evaluated on the device (default metric, row-table call): this launch resets the state
        st.es = None
        es = self._es_opts
        d.es, d.es_reset = None, 0
        if es is not None and es["device"] and not per_el and not st.compat and (corr is None or av is not None) and n_steps > 0:
            ds = ds if ds is not None else self._device_stop(xc, n_steps)
            ring = ds.ring_for(latent_mask if latent_mask.shape == shape else m, m)
            # a bit-packed mask is binary: the ring then travels as bits too (the phase-specialised kernels take no other form)
            ring_bits = ds.ring_bits() if (ring is not None and m_flag == LP_FL_MASK_BITS) else None
            st.es = dict(es, ds=ds, seq=ds.next_seq(), ring=ring, ring_flag=_cabi.LP_FL_ES_RING_BITS if ring_bits is not None else 0)
            d.es, d.es_reset, d.es_seq_base = ds.state.data_ptr(), 1, st.es["seq"]
            d.es_threshold, d.es_patience_eff, d.es_n_steps = es["threshold"], es["patience_eff"], n_steps
            d.es_host, d.es_partials = ds.mailbox.data_ptr(), ds.partials.data_ptr()
            for k in range(3):
                d.es_x0s[k] = ds.x0s[k].data_ptr()
            d.es_ring = ring_bits.data_ptr() if ring_bits is not None else (ring.data_ptr() if ring is not None else None)
            d.es_xte = ds.x_te.data_ptr()
        if not defer_launch:
            self._launch_step(stream)
        st.replace_kind_static = d.replace_kind != LP_REPLACE_KNOWN and not per_el and av is None
        st.k0_desc = _cabi.LpStepDesc.from_buffer_copy(d) if ws.static_io else None
        d.io_table_out, d.io_valid = None, 0          # the think-loop launches share this descriptor
        d.es_reset = 0
        return st

    def _emit(self, st, final):
        """Point the EMIT phase at the fp32 written-back x (final) or at the backbone-input buffer (in loop)."""
        d = self._desc
        if final:
            d.x_in = st.x_final.data_ptr()
            return 0
        d.x_in = st.x_in.data_ptr()
        return st.xin_flag

    # ---- think loop + final backbone call (the part a hipGraph captures) -------------------------------------
    def _think_and_final_model(self, st, model_options, seed):
        d, ws, shape, base_flags, n_steps, stream = self._desc, st.ws, st.shape, st.base_flags, st.n_steps, st.stream
        # a replayed capture bakes the descriptor of every launch: re-point the per-call fields it reads
        d.n_el, d.el_per_row, d.rows = st.n_el, st.n_el // st.rows, st.rows
        stopper = None
        if self._capturing is None and st.es is None:
            stopper = HostStopper.from_options(
                StopOptions.parse(model_options, self.early_stop_threshold, self.early_stop_patience, self.early_stop_hook),
                st.m, st.abt)
            if stopper is not None and self.early_stop_group is not None:
                stopper.sums.reduce_group = self.early_stop_group        # one batch sharded over ranks
        ran = 0
        if st.compat:
            ran = self._loop_compat(ws, shape, st.m, st.y, st.abt, st.current_times, n_steps, model_options, seed, stopper)
            d.phases = LP_PH_EMIT
            d.flags = base_flags | self._emit(st, True)
            self._launch_step(stream)
        elif st.es is not None:
            ran = self._loop_es(st, n_steps, model_options, seed)
        elif stopper is not None:
            ran = self._loop_unfused(st, n_steps, model_options, seed, stopper)
        else:
            auto = self._auto if (self.graph == "auto" and self._capturing is None) else None
            bb_s = 0.0
            for i in range(n_steps):
                last = i == n_steps - 1
                if auto is not None:         # host cost of enqueueing one backbone call (graph="auto" decides on it)
                    t_bb = perf_counter()
                    output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
                    bb_s += perf_counter() - t_bb
                else:
                    output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
                alive = self._set_model_output(d, output, base_flags | self._emit(st, last), shape)
                d.phases = (LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY) | (0 if last else LP_PH_PRE_HALF) | LP_PH_EMIT
                self._set_xi(d, ws.x_t, want_pre=not last)
                self._launch_step(stream)
                del alive
            ran = n_steps
            if auto is not None and n_steps > 0:
                auto[1] += 1
                auto[2] = min(auto[2], bb_s / n_steps)
        self._iterations_run += ran
        self.last_inner_steps = ran
        x_model = st.x_final if self.model_dtype is None else st.x_final.to(self.model_dtype)
        return self.inner_model(x_model, st.sigma_model, model_options=model_options, seed=seed)     # lanpaint.py:151-153

    # ---- epilogue: known-region reprojection + in-place write-back (lanpaint.py:144-157) ----------------------
    def _fill_final_desc(self, f, st, final, out):
        """lp_finalize descriptor for this call.  Returns False when a backbone output had to be converted /
        made dense (then the descriptor points at a temporary and must not be reused for later replays)."""
        shape = st.shape
        converted = []

        def dense(t):
            t0 = t
            if t.shape != shape:
                t = t.expand(shape)
            if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                t = t.float()
            t = t if t.is_contiguous() else t.contiguous()
            if t is not t0:
                converted.append(t)
            return t

        uncond = None
        if isinstance(final, FusedCFGHeads) and final._heads is None and final.cond.dtype == final.uncond.dtype:
            out_model, uncond = dense(final.cond), dense(final.uncond)      # head 0 formed inside lp_finalize
            f.cfg_scale = final.scale
        else:
            out_model = dense(self.unpack_model_output(final)[0])
        f.n_el = st.n_el
        f.flags = (LP_FL_X0_BF16 if out_model.dtype == torch.bfloat16 else
                   LP_FL_X0_F16 if out_model.dtype == torch.float16 else 0) | (LP_FL_CFG_FUSED if uncond is not None else 0) \
            | st.m_flag
        f.uncond = uncond.data_ptr() if uncond is not None else None
        f.model_out, f.y = out_model.data_ptr(), st.y.data_ptr()
        f.mask = st.m_c.data_ptr() if st.m_c is not None else st.m.data_ptr()
        f.x_src, f.x_dst, f.out = st.x_final.data_ptr(), st.xc.data_ptr(), out.data_ptr()
        f.rng_bump_ptr, f.rng_bump, f.io_table = None, 0, None
        self._final_alive = (out_model, uncond)
        return not converted

    def _epilogue(self, st, final, rng_bump=None, in_graph=False):
        f, xc = self._fdesc, st.xc
        out = st.out if st.out is not None else torch.empty_like(xc)
        if not in_graph:               # (in_graph: the replayed graph ended with its own lp_finalize, fed by the I/O table)
            self._fill_final_desc(f, st, final, out)
            if rng_bump is not None:   # replayed Philox launches read a device-side sequence counter: advance it
                f.rng_bump_ptr, f.rng_bump = rng_bump[0].data_ptr(), int(rng_bump[1])
            _cabi.check(self._lib.lp_finalize(ctypes.byref(f), st.stream), "lp_finalize")
        if xc is not st.input_x:
            st.input_x.copy_(xc)
        return out if out.dtype == st.input_x.dtype else out.to(st.input_x.dtype)

    # ------------------------------------------------------------------ xi plumbing
    def _set_xi(self, d, like, want_pre, want_post=True):
        """Draw in the reference's order: the POST half-step of iteration i, then the
        PRE half-step of iteration i+1 (lanpaint.py:277,280,283)."""
        if self.rng == "philox":
            d.xi_post = d.xi_pre = None
            d.rng_kind = _cabi.LP_RNG_PHILOX
            if self._capturing is not None:      # replayed launches: base comes from the device counter
                d.rng_offset, d.rng_offset_ptr = self._cap_offset, self._capturing.data_ptr()
                self._cap_offset += 1
            else:
                d.rng_offset, d.rng_offset_ptr = (1 << 48) + self._philox_offset, None   # disjoint from replayed ones
                self._philox_offset += 1
            self._xi_alive = None
            return
        if self.rng == "torch":
            # the values torch.randn_like(x_t) would return, generated inside the kernel: same generator state in,
            # same values, same state out (the generator's offset is advanced by what the draws consume)
            d.xi_post = d.xi_pre = None
            d.rng_kind = _cabi.LP_RNG_TORCH
            d.rng_bg, d.rng_inc = self._randn_policy(like.device, like.numel())
            used = (int(bool(want_post)) + int(bool(want_pre))) * d.rng_inc
            if self._capturing is not None:      # replayed launches: offsets relative to the state the replace publishes
                d.rng_offset, d.rng_offset_ptr = self._cap_offset, self._capturing.data_ptr()
                self._cap_offset += used
            else:
                gen = self._generator(like.device)
                off = gen.get_offset()
                d.rng_seed, d.rng_offset, d.rng_offset_ptr = gen.initial_seed(), off, None
                gen.set_offset(off + used)
                self._torch_consumed += used
            self._xi_alive = None
            return
        xa = self._draw(like) if want_post else None
        xb = self._draw(like) if want_pre else None
        d.xi_post = xa.data_ptr() if xa is not None else None
        d.xi_pre = xb.data_ptr() if xb is not None else None
        self._xi_alive = (xa, xb)
