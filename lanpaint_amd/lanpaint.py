"""LanPaint's Langevin "think" loop on MI355X: host side.

Drop-in for the reference engine class (/root/reference/src/LanPaint/lanpaint.py:7-328):
same constructor, same `__call__(x, latent_image, noise, sigma, latent_mask,
current_times, model_options, seed, n_steps=None, ...)`, same in-place mutation of
`x` (lanpaint.py:156), same tuple/list/single model-output handling (lanpaint.py:34-43),
same public helper methods.  The arithmetic does NOT run here: every per-element
operation of the loop is one launch of the fused HIP kernel behind the C ABI
(include/lanpaint_hip.h), reached through ctypes with raw device pointers.

Loop shape (the backbone call is the only cut):
    lp_step REPLACE|EMIT|COEFFS       replace step, VP rescale, first model input; the same launch builds the
                                      per-row coefficient table on the device (no host sync)
    for i in range(n):   model(x_in)  -> (x0, x0_BIG)
        lp_step POST|PRE_HALF|EMIT    post-model half of iteration i fused with the
                                      pre-model half of iteration i+1
    model(x) ; lp_finalize            known-region reprojection + write-back of x
graph=True: everything after the first launch is one hipGraph per sigma call (the captured lp_finalize finds x / out
through a device table the first launch publishes).  Inner early stop (default metric): LP_FL_ES on the POST
launches -- the stop rule runs on the device; eager loops poll one verdict per iteration from a pinned mailbox,
replayed loops are gated on the device-side flag.

There is no CPU / eager fallback: a missing extension or a non-HIP tensor raises.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from collections import OrderedDict
from functools import partial
from time import perf_counter

import torch

from . import _cabi
from ._cabi import (LP_FL_ES, LP_FL_ES_GATED, LP_FL_ES_CLOSE, LP_FL_CFG_FUSED, LP_FL_FLOW, LP_FL_MASK_BITS, LP_FL_MASK_U8, LP_FL_XIN_BF16, LP_FL_XIN_F16, LP_FL_PER_ELEMENT, LP_FL_WRITE_X0S, LP_FL_X0_BF16, LP_FL_X0_F16, LP_FL_X0S_GIVEN,
                    LP_PH_EMIT, LP_PH_POST_FIRST, LP_PH_POST_STEADY, LP_PH_PRE_HALF, LP_PH_REPLACE, LP_REPLACE_FLOW,
                    LP_REPLACE_KNOWN, LP_REPLACE_VE)
from .earlystop import HostStopper, StopOptions
from .types import FusedCFGHeads, LangevinState

def raw_stream(device) -> int:
    """hipStream_t of torch's current stream on `device` (torch.cuda.current_stream() builds a Stream object: 2.2 us
    against 0.1 us for the raw accessor, measured on the MI355X box -- scripts/host_cost_probe.py)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:                          # a torch build without the private accessor
        return torch.cuda.current_stream(device).cuda_stream


def tensor_version(t: torch.Tensor) -> int:
    """`t._version` for the per-tensor caches (noise verdict, packed mask, ring), or -1 for an INFERENCE tensor:
    ComfyUI runs its nodes under torch.inference_mode(), whose tensors do not track a version counter (reading it
    raises).  Such a tensor is identified by object identity (weak reference) and address alone -- the caches hold
    per-job constants (the run's noise, the job's mask) that nobody rewrites in place between sigma calls."""
    try:
        return t._version
    except RuntimeError:
        return -1


def _state_x0(state):
    """x0 of a think-loop state as an overridden langevin_dynamics may return it: a LangevinState, a legacy tuple, or None."""
    if isinstance(state, LangevinState):
        return state.x0
    if isinstance(state, tuple) and len(state) >= 3:
        return state[2]
    return None


def _as_f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


def _compact_mask(latent_mask, shape, device):
    """(tensor, LP_FL_MASK_* flag) of the compact copy attached to a binary mask, or (None, 0).
    `_lp_bits`: uint8 storage of the bit-packed form (`pack_mask`); `_lp_u8`: one byte per element."""
    bits = getattr(latent_mask, "_lp_bits", None)
    if bits is not None and bits.dtype == torch.uint8 and bits.is_contiguous() and bits.device == device \
            and tuple(latent_mask.shape) == tuple(shape) and bits.numel() == _cabi.mask_bits_bytes(latent_mask.numel()):
        return bits, LP_FL_MASK_BITS
    u8 = getattr(latent_mask, "_lp_u8", None)
    if u8 is not None and u8.dtype == torch.uint8 and u8.shape == shape and u8.is_contiguous() and u8.device == device:
        # an attached byte mask is the caller's word that the mask is binary: pack it once (the hot kernels take
        # fp32 or bits; LP_FL_MASK_U8 only runs through the run-time-everything kernel) and keep the bits on the tensor
        if latent_mask.is_cuda and latent_mask.dtype == torch.float32 and latent_mask.is_contiguous():
            pack_mask(latent_mask, check=False)
            return latent_mask._lp_bits, LP_FL_MASK_BITS
        return u8, LP_FL_MASK_U8
    return None, 0


def pack_mask(latent_mask: torch.Tensor, *, denoise_mask: bool = False, check: bool = True) -> torch.Tensor:
    """Attach the bit-packed form of a BINARY mask (LP_FL_MASK_BITS, 1 bit per latent element) so that every
    launch of the think loop reads 0.125 B instead of 4 B per element for it.  Returns the fp32 latent mask
    (1 = known) carrying `_lp_bits`; with `denoise_mask=True` the input is ComfyUI's denoise_mask and
    nodes.py:281-283 (`1 - (dm > 0.5)`) is folded into the same launch.  `check` (one host read) rejects soft
    masks, for which the packed form would not be equivalent.
    The packed copy follows the tensor it was made from: the engine compares the tensor's version counter on every call
    and re-packs IN PLACE (same bits buffer: captured graphs stay valid) when the mask was rewritten; a tensor without a
    version counter (torch.inference_mode) is re-packed on every sigma call -- one small launch, what the reference does
    on every call anyway (nodes.py:277-283)."""
    if not latent_mask.is_cuda:
        raise ValueError("pack_mask needs a mask on a HIP device")
    src = _as_f32c(latent_mask)
    n = src.numel()
    bits = torch.empty(_cabi.mask_bits_bytes(n), dtype=torch.uint8, device=src.device)
    flag = torch.zeros(1, dtype=torch.int32, device=src.device) if (check and not denoise_mask) else None
    with torch.cuda.device(src.device):
        _cabi.check(_cabi.load().lp_pack_mask(src.data_ptr(), n, _cabi.LP_FL_MASK_DENOISE if denoise_mask else 0,
                                              bits.data_ptr(), flag.data_ptr() if flag is not None else None,
                                              torch.cuda.current_stream(src.device).cuda_stream), "lp_pack_mask")
    if flag is not None and int(flag.item()):
        raise ValueError("pack_mask: the mask has values other than 0 and 1; soft masks cannot be bit-packed")
    out = (1 - (src > 0.5).to(torch.float32)) if denoise_mask else latent_mask
    if out.dtype != torch.float32 or not out.is_contiguous():
        out = src
    out._lp_bits = bits
    # what the bits were made from: (weak reference to the source tensor, its version then, denoise form?) -- see refresh_packed_mask
    out._lp_bits_of = (weakref.ref(latent_mask), tensor_version(latent_mask), bool(denoise_mask))
    return out


def refresh_packed_mask(packed: torch.Tensor, source: torch.Tensor = None) -> bool:
    """Bring the bit-packed copy attached to `packed` (pack_mask's return value) up to date with the tensor it was made from
    (`source`, default: the recorded one), IN PLACE -- same bits buffer, same fp32 latent mask tensor, so captured graphs and
    the engine's identity checks keep matching.  A source with a version counter is re-packed only when the counter moved;
    an inference tensor (no counter) every time.  One launch (lp_pack_mask_latent).  Returns True when it re-packed.
    A packed mask must STAY binary: the denoise form thresholds at 0.5 by definition (nodes.py:281-283); for a mask packed from
    its own fp32 tensor a rewrite to soft values is reported by the NEXT call (ValueError), see below."""
    rec = getattr(packed, "_lp_bits_of", None)
    bits = getattr(packed, "_lp_bits", None)
    if rec is None or bits is None:
        return False
    src = source if source is not None else rec[0]()
    if src is None or not src.is_cuda or src.numel() != packed.numel():
        return False
    ver = tensor_version(src)
    if ver != -1 and ver == rec[1] and (source is None or source is rec[0]()):
        return False
    s32 = _as_f32c(src)
    denoise = rec[2]
    # the fp32 latent mask is rewritten too unless it IS the source (pack_mask(latent_mask): the caller's own tensor)
    same = (not denoise) and s32.data_ptr() == packed.data_ptr()
    lib = _cabi.load()
    with torch.cuda.device(src.device):
        stream = raw_stream(src.device)
        if same:
            # The caller vouched for a BINARY mask when packing it; a rewrite to soft values would be binarised at 0.5 without
            # a word.  The re-pack raises the kernel's "values other than 0 and 1" flag straight into pinned host memory (no
            # device -> host copy, no sync); what the PREVIOUS re-pack left there is looked at now.
            soft = getattr(packed, "_lp_soft_flag", None)
            if soft is None:
                soft = packed._lp_soft_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
            elif int(soft[0]) != 0:
                soft[0] = 0
                raise ValueError("pack_mask: the packed mask was rewritten in place to values other than 0 and 1; soft masks cannot "
                                 "be bit-packed (hand the engine the plain fp32 mask instead)")
            _cabi.check(lib.lp_pack_mask(s32.data_ptr(), s32.numel(), 0, bits.data_ptr(), soft.data_ptr(), stream), "lp_pack_mask")
        else:
            _cabi.check(lib.lp_pack_mask_latent(s32.data_ptr(), s32.numel(), _cabi.LP_FL_MASK_DENOISE if denoise else 0,
                                                bits.data_ptr(), packed.data_ptr(), stream), "lp_pack_mask_latent")
    packed._lp_bits_of = (weakref.ref(src), tensor_version(src) if same else ver, denoise)
    return True


def pack_indicator(indicator: torch.Tensor, shape) -> tuple:
    """(bits, audio share, rows share equally?) of an AV pack's stream indicator (lanpaint.py:68-73: 1 = audio element), or None
    when it is not a 0/1 tensor broadcastable to the latent -- then the reference-shaped per-element path runs.  Cached on the
    tensor (weak identity + version), one host read when first packed.  An inference tensor has no version counter: a binary
    one is re-packed IN PLACE on every call with the "values other than 0 and 1" flag checked (one small host read per sigma
    call: a rewrite to soft values must not be binarised silently), a soft one is looked at again on every call.
    The third item: the device-side stopper takes its `abt` mean from the two time rows of every batch row and ONE audio share
    (lp_step_desc.av_frac); that equals the reference's mean over the blended abt tensor (earlystop.py:104-110) only when every
    batch row holds the same share of audio elements -- True for pack layouts (the indicator is a broadcast [1, ...] tensor)."""
    if not indicator.is_cuda:
        return None
    ver = tensor_version(indicator)
    rec = getattr(indicator, "_lp_av", None)
    if rec is not None and (rec[2] != tuple(shape) or (rec[0] is not None and rec[0].device != indicator.device)):
        rec = None
    if rec is not None and rec[3] == ver and ver != -1:
        return (rec[0], rec[1], rec[4]) if rec[0] is not None else None
    try:
        full = _as_f32c(indicator if tuple(indicator.shape) == tuple(shape) else indicator.expand(shape))
    except RuntimeError:
        return None
    n = full.numel()
    bits = rec[0] if (rec is not None and rec[0] is not None) else \
        torch.empty(_cabi.mask_bits_bytes(n), dtype=torch.uint8, device=full.device)
    flag = torch.zeros(1, dtype=torch.int32, device=full.device)
    with torch.cuda.device(full.device):
        _cabi.check(_cabi.load().lp_pack_mask(full.data_ptr(), n, 0, bits.data_ptr(), flag.data_ptr(), raw_stream(full.device)),
                    "lp_pack_mask")
    if rec is not None and rec[0] is not None and ver == -1:
        # no version counter, packed before: bits re-derived in place above (captured launches bake their address); the audio
        # share is a per-job constant of the pack layout and keeps its first value; only the flag is read
        if int(flag.item()):
            try:
                indicator._lp_av = (None, 0.0, tuple(shape), ver, False)
            except Exception:
                pass
            return None
        return bits, rec[1], rec[4]
    rows = int(shape[0]) if len(shape) else 1
    per_row = full.reshape(rows, -1).sum(dim=1, dtype=torch.float64).cpu()        # (the one host read; also waits for the flag)
    frac = float(per_row.sum()) / n
    soft = bool(int(flag.item()))
    rows_equal = bool((per_row == per_row[0]).all())
    try:
        indicator._lp_av = (None, 0.0, tuple(shape), ver, False) if soft else (bits, frac, tuple(shape), ver, rows_equal)
    except Exception:
        pass
    return None if soft else (bits, frac, rows_equal)


def aten_randn_policy(numel: int, multi_processor_count: int, max_threads_per_multi_processor: int):
    """calc_execution_policy of ATen's random kernels for one fp32 randn of `numel` elements: 256-thread blocks, the
    grid capped at SMs * (maxThreadsPerSM / 256), four values per thread and loop trip.  Returns (block * grid,
    philox-offset increment of the call)."""
    grid = min(multi_processor_count * (max_threads_per_multi_processor // 256), (numel + 255) // 256)
    bg = 256 * grid
    return bg, ((numel - 1) // (bg * 4) + 1) * 4


def _noise_scaling_kind(model_sampling):
    """Which closed form the replace step may fuse (lanpaint.py:84-94), and the noise scale that form uses.
    'callback' keeps the reference behaviour for any model_sampling: call its noise_scaling.
    A model_sampling that DECLARES its form (`lanpaint_noise_scaling_kind`) is taken at its word, `noise_scale`
    attribute included; ComfyUI's stock CONST.noise_scaling is `sigma * noise + (1 - sigma) * latent` with no
    noise_scale term, so a subclass that inherits it but carries a `noise_scale` attribute still gets scale 1."""
    kind = getattr(model_sampling, "lanpaint_noise_scaling_kind", None)
    if kind in ("ve", "flow"):
        return kind, float(getattr(model_sampling, "noise_scale", 1.0))
    try:                                   # ComfyUI present: recognise its stock EPS / CONST forms
        import comfy.model_sampling as cms  # type: ignore
        fn = getattr(type(model_sampling), "noise_scaling", None)
        if fn is getattr(getattr(cms, "CONST", None), "noise_scaling", object()):
            return "flow", 1.0
        if fn is getattr(getattr(cms, "EPS", None), "noise_scaling", object()):
            return "ve", 1.0
    except Exception:
        pass
    return "callback", 1.0


class _Workspace:
    """Device buffers reused across sigma calls of one engine (torch-owned).  static_io: also owns the
    backbone-input / final-x buffers (a captured call bakes their addresses)."""

    def __init__(self, like: torch.Tensor, static_io: bool = False, model_dtype=None):
        self.shape, self.device = tuple(like.shape), like.device
        self.x_t = torch.empty_like(like)
        self.C = torch.empty_like(like)
        self.coef = torch.empty((like.shape[0], _cabi.LP_COEF_STRIDE), dtype=torch.float32, device=like.device)
        self.coef_av = None      # lazily: [2 * rows][LP_COEF_STRIDE], two time sets per row (AV packs, LP_FL_AV)
        self.av_times = None     # lazily: [4][2 * rows] interleaved (VE, abt, replace sigma, model time) inputs of that table
        self.x0s = []            # lazily: rotating buffers for LangevinState.x0 (early stop only)
        self.static_io = static_io
        if static_io:
            self.x_final = torch.empty_like(like)
            self.x_in = self.x_final if model_dtype is None else torch.empty_like(like, dtype=model_dtype)

    def matches(self, like):
        return self.shape == tuple(like.shape) and self.device == like.device


class _DeviceStop:
    """Buffers of the inner early stop evaluated on the device (LP_FL_ES): the lp_es_state, three rotating x0s
    buffers, the accumulator sets the blocks add their sums into and the pinned-host mailbox the trace records go to."""

    def __init__(self, like: torch.Tensor, n_steps: int):
        dev = like.device
        self.shape, self.device, self.n_cap = tuple(like.shape), dev, max(8, int(n_steps))
        self.x0s = [torch.empty_like(like) for _ in range(3)]
        self.x_te = torch.empty_like(like)      # gated loops: the state after the tentative half-step (lp_step_desc.es_xte)
        init = _cabi.LpEsState()
        init.cur_slot = init.anchor_slot = -1
        for k in range(3):
            init.x0s_buf[k] = self.x0s[k].data_ptr()
        raw = torch.frombuffer(bytearray(bytes(init) * 2), dtype=torch.uint8)     # two slots (folded gated loops ping-pong)
        self.state = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
        self.state.copy_(raw)
        # the accumulator sets the blocks of an early-stop launch add their sums into (LP_ES_ACC_DOUBLES)
        self.partials = torch.zeros(_cabi.LP_ES_ACC_DOUBLES, dtype=torch.float64, device=dev)
        self.mailbox = torch.zeros(_cabi.LP_ES_TRACE0 + 8 * self.n_cap, dtype=torch.float64).pin_memory()
        self.f64 = self.mailbox.numpy()
        self.i64 = self.mailbox.view(torch.int64).numpy()
        self.seq_base = 0
        self.seen_total = 0         # of the device's running iteration count, what the engine has accounted already
        self.ring = None            # (weakref(mask), version, ring tensor | None, bit-packed ring | None)

    def matches(self, like, n_steps):
        return self.shape == tuple(like.shape) and self.device == like.device and n_steps <= self.n_cap

    def next_seq(self):
        self.seq_base += 2 * _cabi.LP_ES_SEQ_DONE
        return self.seq_base

    def ring_for(self, key, mask):
        """Mask-edge ring weight (earlystop.py:32-49; 4-D latents only) of the dense fp32 `mask`, computed once per
        mask tensor object `key` and version."""
        c = self.ring
        ver = (tensor_version(key), key.data_ptr())
        # (no version counter -- inference mode --: recomputed on every call; always into the SAME buffers when the shape allows,
        # because captured early-stop launches bake the ring's address)
        if c is None or c[0]() is not key or c[1] != ver or ver[0] == -1:
            ring = None
            if mask.dim() == 4:
                old = c[2] if c is not None else None
                ring = old if (old is not None and old.shape == mask.shape and old.device == mask.device) else torch.empty_like(mask)
                b, ch, h, w = mask.shape
                with torch.cuda.device(mask.device):
                    _cabi.check(_cabi.load().lp_boundary_ring(mask.data_ptr(), ring.data_ptr(), b * ch, h, w,
                                                              torch.cuda.current_stream(mask.device).cuda_stream),
                                "lp_boundary_ring")
            bits = None
            if ring is not None:       # the bit-packed form the hard-mask kernels read (LP_FL_ES_RING_BITS): ring pixels are inpaint
                # pixels, so with a binary mask the weight (1 - m) on them is exactly 1 and the ring IS a bit per element
                old_bits = c[3] if c is not None else None
                n_bytes = _cabi.mask_bits_bytes(ring.numel())
                bits = old_bits if (old_bits is not None and old_bits.numel() == n_bytes and old_bits.device == ring.device) \
                    else torch.empty(n_bytes, dtype=torch.uint8, device=ring.device)
                with torch.cuda.device(mask.device):
                    _cabi.check(_cabi.load().lp_pack_mask(ring.data_ptr(), ring.numel(), 0, bits.data_ptr(), None,
                                                          torch.cuda.current_stream(mask.device).cuda_stream), "lp_pack_mask")
            self.ring = c = (weakref.ref(key), ver, ring, bits)
        return c[2]

    def ring_bits(self):
        """Bit-packed form of the ring `ring_for` returned last (None for latents without a ring)."""
        return self.ring[3] if self.ring is not None else None

    def wait(self, seq, device):
        """Block until the mailbox sequence word reaches `seq` (spin briefly, then sleep on the stream)."""
        i64 = self.i64
        for _ in range(20000):
            if i64[0] >= seq:
                return
        torch.cuda.current_stream(device).synchronize()
        if i64[0] < seq:
            raise RuntimeError("early-stop mailbox was not written (expected sequence %d, found %d)" % (seq, int(i64[0])))


class _CallState:
    """Everything one sigma call carries from its prologue to its loop and epilogue."""
    __slots__ = ("input_x", "xc", "shape", "n_el", "rows", "flow", "ws", "stream", "sigma", "y", "m", "m_c", "m_flag", "abt",
                 "current_times", "base_flags", "keep", "t_model", "sigma_model", "compat", "n_steps", "x_final", "x_in",
                 "xin_flag", "k0_desc", "replace_kind_static", "out", "es")


class _CapturedCall:
    """The think loop + final backbone call of one sigma call captured as a hipGraph, with the workspace
    whose addresses it bakes in and the device-side Philox counter its launches read."""

    def __init__(self, counter):
        self.graph = torch.cuda.CUDAGraph()
        self.counter = counter
        self.ws = None
        self.final = None        # the backbone's final output object (static tensors)
        self.ran = 0
        self.launches = 0
        self.keep = None
        self.fast = False        # steady-state replay may reuse the snapshotted descriptors
        self.rows, self.flow, self.hyper, self.k0_desc, self.f_desc = 0, False, None, None, None
        self.call = None         # lp_call_desc: the whole enqueue sequence of a replay in one C call
        self.raw_exec = None     # hipGraphExec_t, when launching it without torch's replay() is equivalent
        self.ident = None        # what the caller passed last time (identity pre-check of the next call)
        self.final_in_graph = False   # lp_finalize is a node of the graph (reads x / out through the I/O table)
        self.es = None                # early stop evaluated on the device inside the graph (LP_FL_ES_GATED): options + buffers
        self.n_steps = 0
        self.key = None               # its key in the engine's graph table (siblings differ in the step count only)
        self.tail = None              # lp_call_desc that launches the graph alone (the replace went ahead, begin_call)
        self.model_options = None     # the dict the captured backbone calls were made with (kept alive: its id is in the key)
        self.alive = True             # still in the engine's graph table
        self.siblings = {}            # n_steps -> the capture of the same call shape for that count (finish_call)
        self.times_seen = ()          # the (VE sigma, abt, flow t) tuples that passed the identity pre-check
        self.node_table = None        # (exec array by inner-step count, captures, -, graphs seen, options): lp_node_call's table
        self.binding = None           # lp_graph_binding: the replace launch is node 0 of the graph (ONE hipGraphLaunch per call)
        self.tail_handles = None      # (hipGraph_t, hipGraphExec_t) of the same graph without node 0 (begin_call / finish_call)

    def __del__(self):
        h, self.tail_handles = self.tail_handles, None
        if h is not None:
            try:
                _cabi.load().lp_graph_release(h[0], h[1])
            except Exception:
                pass


class LanPaint:
    MAX_GRAPHS = 16          # captured sigma calls kept per engine (one per distinct n_steps / tensor set)
    AUTO_MAX_BACKBONE_HOST_US = 100.0   # graph="auto": only loops whose backbone call costs the host less than this are captured

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

    def _device_stop(self, like, n_steps):
        if self._ds is None or not self._ds.matches(like, n_steps):
            self._ds = _DeviceStop(like, n_steps)
        return self._ds

    def _es_trace(self, es, ds, i):
        """Append the reference's trace record of iteration i (earlystop.py:315-334) from the mailbox."""
        trace = es["trace"]
        if trace is None or ds.f64[3] == 0.0:
            return
        rec = ds.f64[_cabi.LP_ES_TRACE0 + 8 * i: _cabi.LP_ES_TRACE0 + 8 * i + 8]
        thr_eff, opt = float(ds.f64[4]), (lambda v: None if v != v else float(v))
        trace.append({"case_id": es["tags"][0], "outer_step": es["tags"][1], "bench_timestep": es["tags"][2],
                      "inner_step": i + 1, "dist": float(rec[0]), "dist_inpaint": opt(rec[1]), "dist_ring": opt(rec[2]),
                      "dist_drift": opt(rec[3]), "threshold": thr_eff, "threshold_eff": thr_eff,
                      "patience_counter": int(rec[4]), "patience_eff": int(es["patience_eff"]), "abt": float(ds.f64[5]),
                      "custom_dist": False, "stopped": bool(rec[5])})

    def _es_resolve(self):
        """A replayed early-stop loop reports how far it ran: wait for its "done" word, account the iterations, hand
        over the trace records, and put torch's generator where the reference leaves it after that many iterations."""
        p = self._es_pending
        if p is None:
            return
        self._es_pending = None
        ds, seq, n_steps, es, dev, inc = p
        ds.wait(seq + _cabi.LP_ES_SEQ_DONE, dev)
        n_ran = int(ds.f64[1])
        total = int(ds.f64[6])             # the device counts across calls: every replay since the last collection
        self._iterations_run += total - ds.seen_total
        ds.seen_total = total
        self.last_inner_steps = n_ran
        for i in range(n_ran):
            self._es_trace(es, ds, i)
        if inc and n_ran < n_steps:      # the launches past the stop drew nothing the reference would have drawn
            gen = self._generator(dev)
            back = 2 * (n_steps - n_ran) * inc
            gen.set_offset(gen.get_offset() - back)
            self._torch_consumed -= back

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
        return name in self.__dict__ or getattr(type(self), name) is not _OWN_METHODS[_OWN_NAMES.index(name)]

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

    # ------------------------------------------------------------------ split-phase call (KSamplerX0Inpaint)
    def begin_call(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed):
        """First half of `__call__` for a caller that does not know `n_steps` yet: enqueue what does not depend on it
        -- the replace step, the VP rescale, the coefficient table, the I/O table -- and return a token for
        `finish_call(token, n_steps)`; None when the call is not a steady-state graph replay (then use `__call__`).
        KSamplerX0Inpaint needs the device to tell it sigma's position in the schedule (nodes.py:286-299) before it
        can fix the inner-step count; with the head of the call already queued the GPU goes on working while the host
        picks the graph."""
        cap = self._last_cap
        if cap is None or cap.tail is None or self.model_dtype is not None or not x.is_cuda:
            return None
        self.img_dim_size = len(x.shape)
        self.latent_image, self.noise = latent_image, noise
        self.audio_indicator = self.current_times_audio = self.audio_correction = None
        self._noise_regenerated = self._noise_is_zero(noise)
        if not self._same_call(cap, x, sigma, latent_mask, current_times, cap.ident[4], model_options, seed):
            return None
        lib, stream = self._lib, self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        out = torch.empty_like(x)
        k0 = cap.k0_desc
        k0.x, k0.noise = x.data_ptr(), noise.data_ptr()
        k0.t_ve, k0.t_abt, k0.t_rsig = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr()
        k0.t_model = (current_times[2] if cap.flow else ve).data_ptr()
        k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        off = 0
        if self.rng == "torch":        # publish the generator state; finish_call advances it by what its graph draws
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
        _cabi.check(lib.lp_step(ctypes.byref(k0), stream), "lp_step")
        return (cap, x, out, sigma, latent_mask, current_times, model_options, seed, off, stream)

    def finish_call(self, token, n_steps):
        """Second half: replay the think loop + final backbone call + finalise captured for `n_steps`."""
        cap0, x, out, sigma, latent_mask, current_times, model_options, seed, off, stream = token
        if n_steps is None:
            n_steps = self.n_steps
        cap = cap0
        if cap0.ident[4] != n_steps:
            cap = cap0.siblings.get(n_steps)
            if cap is None or not cap.alive:
                cap = self._graphs.get(cap0.key[:2] + (int(n_steps),) + cap0.key[3:])
                if cap is None or cap.tail is None or cap.model_options is not model_options or cap.ws is not cap0.ws:
                    # no capture for this count yet: the ordinary path captures it (and re-enqueues the replace step,
                    # which reads the same untouched x and publishes the same generator state)
                    return self(x, self.latent_image, self.noise, sigma, latent_mask, current_times, model_options, seed,
                                n_steps=n_steps)
                cap0.siblings[n_steps] = cap
                self._graphs.move_to_end(cap.key)
        if self.rng == "torch" and cap.launches:
            self._generator(x.device).set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        _cabi.check(self._lib.lp_replay_call(ctypes.byref(cap.tail), stream), "lp_replay_call")
        return out

    def node_call(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, nd):
        """The steady state of KSamplerX0Inpaint.__call__ in ONE trip through the FFI (lp_node_call): sigma -> times and
        the two scalars of the inner-step rule, the replace launch of this call, the wait for the scalars, the rule
        (nodes.py:286-299), and the launch of the graph captured for the resulting count.  `nd` is the caller's
        LpNodeCallDesc with the sigma / schedule / mailbox / rule fields filled in.  Returns (out, n_eff), or None when the
        call is not a steady-state replay (nothing enqueued: the caller takes its ordinary path)."""
        cap0 = self._last_cap
        if cap0 is None or cap0.tail is None or self.model_dtype is not None or not x.is_cuda:
            return None
        self.img_dim_size = len(x.shape)
        self.latent_image, self.noise = latent_image, noise
        self.audio_indicator = self.current_times_audio = self.audio_correction = None
        self._noise_regenerated = self._noise_is_zero(noise)
        if not self._same_call(cap0, x, sigma, latent_mask, current_times, cap0.ident[4], model_options, seed):
            return None
        table = cap0.node_table
        if table is None or table[3] != len(self._graphs) or table[4] is not model_options:
            table = self._node_table(cap0, nd.n_steps, model_options)
        stream = self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        out = torch.empty_like(x)
        k0 = cap0.k0_desc
        k0.x, k0.noise = x.data_ptr(), noise.data_ptr()
        k0.t_ve, k0.t_abt, k0.t_rsig = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr()
        k0.t_model = (current_times[2] if cap0.flow else ve).data_ptr()
        k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        off = 0
        if self.rng == "torch":
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
        if getattr(nd, "_lp_table", None) is not table:     # (the table changes when a new count has been captured)
            nd._lp_table = table
            nd.replace, nd.exec_by_count, nd.n_counts = table[2], table[0], len(table[1])
            nd.valid_word = self._rng_state(x.device).data_ptr() + 32
        rc = self._lib.lp_node_call(ctypes.byref(nd), stream)
        if rc != _cabi.LP_OK:
            # A failed call may have left a speculated, self-voided run in the queue.  The library tries to restore the word
            # the captured lp_finalize checks; do not rely on it: restore it from here as well (an ordinary torch write in
            # stream order) and forget every capture of this engine, so nothing replays against half-published state.
            try:
                self._rng_state(x.device)[4] = 1
            except Exception:
                pass
            self._forget_captures()
            _cabi.check(rc, "lp_node_call")
        n_eff = nd.n_eff
        if not nd.launched:
            # no capture for this count yet: the ordinary path captures it (and re-enqueues the replace step, which reads
            # the same untouched x and publishes the same generator state)
            cap0.node_table = None
            return self(x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, n_steps=n_eff), n_eff
        cap = table[1][n_eff]
        if self.rng == "torch" and cap.launches:
            self._generator(x.device).set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        return out, n_eff

    def _forget_captures(self):
        """Drop every captured sigma call (after a failed native call: the next call takes the full path again)."""
        self._last_cap = None
        for cap in self._graphs.values():
            cap.alive = False
            cap.node_table = None
            cap.siblings = {}
        self._graphs.clear()

    def _node_table(self, cap0, n_max, model_options):
        """hipGraphExec_t of the tail graph captured for every inner-step count 0 .. n_max of this call shape (NULL where
        none exists yet), as the array lp_node_call indexes; holds the captures alive."""
        caps = []
        for n in range(int(n_max) + 1):
            cap = cap0 if cap0.ident[4] == n else cap0.siblings.get(n)
            if cap is None or not cap.alive:
                cap = self._graphs.get(cap0.key[:2] + (n,) + cap0.key[3:])
                if cap is not None and (cap.tail is None or cap.model_options is not model_options or cap.ws is not cap0.ws):
                    cap = None
                if cap is not None:
                    cap0.siblings[n] = cap
            caps.append(cap if (cap is not None and cap.tail is not None) else None)
        arr = (ctypes.c_void_p * len(caps))(*[(c.tail.graph_exec if c is not None else None) for c in caps])
        cap0.node_table = (arr, caps, ctypes.pointer(cap0.k0_desc), len(self._graphs), model_options)
        return cap0.node_table

    # ------------------------------------------------------------------ hipGraph replay of one sigma call
    def _same_call(self, cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed):
        """Identity pre-check of the steady state (a sampler calls the engine once per sigma with the same
        latent_image / mask / options objects): everything `_graph_eligible` and the graph key look at, without
        building the key.  Any miss falls through to the full path."""
        i = cap.ident
        y, nz = self.latent_image, self.noise
        ve, abt, ft = current_times
        f32 = torch.float32
        return (i[0] is y and i[1] is latent_mask and i[2] is model_options and i[3] == x.shape and i[4] == n_steps
                and i[5] == seed and i[6] is self.rng and self.graph and not self._noise_regenerated
                and i[7] == y.data_ptr() and i[8] == latent_mask.data_ptr()
                and i[9] is getattr(latent_mask, "_lp_bits", None) and i[10] is getattr(latent_mask, "_lp_u8", None)
                and i[11] == x.device and i[12] == sigma.numel()
                and self._times_ok(cap, current_times)
                and self.audio_indicator is None and self.audio_correction is None
                and not (self.early_stop_threshold > 0.0 and self.early_stop_patience > 0)
                and x.dtype == f32 and sigma.dtype == f32
                and nz.dtype == f32 and nz.shape == x.shape and x.is_contiguous() and sigma.is_contiguous()
                and nz.is_contiguous()
                and x.device.index == torch.cuda.current_device() and i[16] == self._override_state()
                and i[17] == self._hyper_key() and (x.data_ptr() & 15) == 0 and (nz.data_ptr() & 15) == 0
                and not (isinstance(model_options, dict) and "lanpaint_semantic_stop" in model_options))

    @staticmethod
    def _times_ok(cap, current_times):
        """The time tensors fit the capture (size, fp32, dense).  A caller that hands the same tensor objects call after
        call -- KSamplerX0Inpaint alternates between two sets -- is only checked once per set."""
        i, f32 = cap.ident, torch.float32
        ve, abt, ft = current_times
        for seen in cap.times_seen:                     # identity, never tensor ==
            if seen[0] is ve and seen[1] is abt and seen[2] is ft:
                return True
        ok = (i[13] == ve.numel() and i[14] == abt.numel() and i[15] == ft.numel() and ve.dtype == f32 and abt.dtype == f32
              and ft.dtype == f32 and ve.is_contiguous() and abt.is_contiguous() and ft.is_contiguous())
        if ok:
            cap.times_seen = (cap.times_seen + ((ve, abt, ft),))[-2:]
        return ok

    def _hyper_key(self):
        """The public hyper-parameters a captured launch bakes in (the reference reads them on every call)."""
        return (self.chara_lamb, self.chara_beta, self.step_size, self.min_step_frac)

    def _override_state(self):
        """Which of the three overridable methods are not this module's own (on the instance or on its class -- compared with
        the functions as DEFINED here, so patching the base class itself counts too), plus the model-type switches."""
        d, t, o = self.__dict__, type(self), _OWN_METHODS
        return ("langevin_dynamics" in d or t.langevin_dynamics is not o[0], "score_model" in d or t.score_model is not o[1],
                "prepare_step_size" in d or t.prepare_step_size is not o[2], self.IS_FLUX, self.IS_FLOW, self.model_dtype)

    def _graph_eligible(self, x, model_options, sigma, current_times):
        if not self.graph or self._graph_blocked or callable(self.rng) or self._noise_regenerated:
            return False         # (regenerated noise is a fresh tensor per call: nothing stable to bake into a graph)
        rows = x.shape[0] if x.ndim else 1
        if any(t.numel() not in (1, rows) for t in (sigma, *current_times)):
            return False         # per-element times: the general path, eager only
        if self.audio_indicator is not None or self.audio_correction is not None:
            # AV packs replay only on the two-row table (LP_FL_AV: per-row time pairs + a 0/1 indicator), whose per-call inputs --
            # the interleaved times, the correction tensor -- live in workspace buffers the prologue refreshes; the reference-
            # shaped per-element form builds fresh full-size tensors per call.  (A gated early stop rides along since round 5: a
            # stopped launch re-emits every element with its own stream's scale; rows with different audio shares keep the host
            # stopper -- the device-side threshold takes ONE share, see __call__.)
            if (self.audio_indicator is None or self.current_times_audio is None
                    or any(t.numel() not in (1, rows) for t in self.current_times_audio)
                    or os.environ.get("LANPAINT_AMD_AV_TABLE", "1") == "0"):
                return False
            packed = self._indicator_pack(self.audio_indicator, x.shape)
            if packed is None or (self._es_opts is not None and not packed[2]):
                return False
        if self._es_opts is not None and (not self._es_opts["device"] or self.rng not in ("torch", "philox")):
            return False         # a custom distance_fn / a sharded batch keeps the stopper on the host; a gated loop
                                 # redoes its tentative half-step from a counter-based in-kernel generator only
        if self._overridden("langevin_dynamics") or self._overridden("score_model") or \
                self._overridden("prepare_step_size"):
            return False
        return x.dtype == torch.float32 and x.numel() > 0

    def _auto_ready(self, x, latent_mask, model_options):
        """graph="auto": has this job been seen running eagerly, with a backbone cheap enough on the host for the loop to
        be launch bound?  The first call with a new (latent_image, mask, model_options, shape) starts the record (and
        runs eagerly: its plain loop times the backbone calls); later calls of the same job ask it."""
        a = self._auto
        if a is not None and a[0][0]() is self.latent_image and a[0][1]() is latent_mask and a[0][2] == id(model_options) \
                and a[0][3] == x.shape and a[0][4] == x.device:
            if self._es_opts is not None:
                return False             # (the inner early stop is captured on request only: graph=True)
            return a[1] >= 1 and 1e6 * a[2] < self.AUTO_MAX_BACKBONE_HOST_US
        # [signature, eager calls seen, cheapest per-call mean of the backbone's host time so far (s)]
        # (weak references / an id: the record must not keep a finished job's tensors and options alive; a recycled id
        # only means one more eager call before the capture, whose key checks the dict by identity anyway)
        self._auto = [(weakref.ref(self.latent_image), weakref.ref(latent_mask), id(model_options), x.shape, x.device), 0,
                      float("inf")]
        return False

    def _auto_capture(self, key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """graph="auto": capture, but never let a backbone that cannot be captured take the call down, and take nothing
        on trust -- the first replay is compared with eager launches of the same call before the capture is used."""
        import warnings
        dev = x.device
        saved = (self._iterations_run, torch.cuda.get_rng_state(dev), self._es_opts, self._torch_consumed, self._philox_offset)

        def give_up(why):
            self._capturing = None
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
            self._iterations_run, _, self._es_opts, self._torch_consumed, self._philox_offset = saved
            torch.cuda.set_rng_state(saved[1], dev)
            self._graph_blocked = True
            cap = self._graphs.pop(key, None)
            if cap is not None:
                cap.alive = False
            self._last_cap = None
            warnings.warn("lanpaint_amd: graph='auto' stays with eager launches for this engine: " + why)
            return None

        try:
            cap = self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
        except Exception as e:                       # e.g. a host sync inside the backbone while the stream is capturing
            return give_up("the backbone could not be captured (%s: %s)" % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
        if cap is None or self.rng != "torch":
            return cap
        # one replay and one eager run of this very call on copies of x, from the same generator state: they must agree
        # bit for bit (same kernels, same noise stream), or something in the backbone does not survive capture
        try:
            xa, xb = x.clone(), x.clone()
            out_a = self._run_capture(cap, xa, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            torch.cuda.set_rng_state(saved[1], dev)
            self._last_cap = None
            out_b = self.LanPaint(xb, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            same = bool(torch.equal(out_a, out_b)) and bool(torch.equal(xa, xb))
        except Exception as e:
            return give_up("checking the capture against eager launches failed (%s)" % type(e).__name__)
        if not same:
            return give_up("a replayed sigma call does not reproduce the eager one (the backbone keeps state the graph does not see)")
        self._iterations_run, _, _, self._torch_consumed, self._philox_offset = saved
        torch.cuda.set_rng_state(saved[1], dev)
        return cap

    def _call_graphed(self, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """One sigma call with its think loop replayed as a hipGraph.  Only the part BETWEEN the replace step
        and the finalise is captured (N x [backbone, fused step] + the final backbone call): the prologue
        (lp_coeffs, replace) and the epilogue (lp_finalize) are ordinary launches that read / write the
        caller's tensors directly, so nothing of x / noise / out has to be staged through static buffers."""
        m_c, _ = _compact_mask(latent_mask, x.shape, x.device)
        # every pointer the captured launches bake in is part of the key (y and the mask; the tensors stay the caller's)
        key = (tuple(x.shape), x.device.index, int(n_steps), bool(IS_FLUX), bool(IS_FLOW), self.latent_image.data_ptr(),
               latent_mask.data_ptr(), m_c.data_ptr() if m_c is not None else 0, int(sigma.numel()),
               tuple(int(t.numel()) for t in current_times), id(model_options), seed, self.rng, self._hyper_key(),
               self.model_dtype, None if self._es_opts is None else (self._es_opts["threshold"], self._es_opts["patience_eff"],
                                                                      self._es_opts["trace"] is not None),
               # (AV: the captured launches bake the address of the indicator's bit-packed copy -- that address, not the tensor's
               # id(), which another tensor of the same shape can recycle)
               None if self.audio_indicator is None else (self._indicator_pack(self.audio_indicator, x.shape)[0].data_ptr(),
                                                          self.audio_correction is not None))
        cap = self._graphs.get(key)
        if cap is not None and cap.model_options is not model_options:
            del self._graphs[key]        # another dict at a recycled id(): the captured backbone calls used the old one
            cap.alive = False
            cap = None
        if cap is None:
            capture = self._auto_capture if self.graph == "auto" else self._capture
            cap = capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            if cap is None:          # not capturable after all (see _capture): the eager path
                return self.LanPaint(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            while len(self._graphs) > self.MAX_GRAPHS:       # bound the static memory held by stale captures
                self._graphs.popitem(last=False)[1].alive = False
        else:
            self._graphs.move_to_end(key)
        return self._run_capture(cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)

    def _run_capture(self, cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """Replay `cap` for this call's tensors."""
        srcs = (x, sigma, current_times[0], current_times[1], current_times[2], self.noise)
        fast = cap.fast and all(t.dtype == torch.float32 and t.is_contiguous() for t in srcs) and self.noise.shape == x.shape \
            and (x.data_ptr() & 15) == 0 and (self.noise.data_ptr() & 15) == 0
        if not fast and cap.binding is not None:
            # the capture holds its own replace launch (node 0), which only takes dense fp32 16-byte-aligned caller
            # tensors: this call's do not qualify, so it runs as eager launches
            self._last_cap = None
            return self.LanPaint(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        if fast:
            if not isinstance(model_options, dict) or "lanpaint_semantic_stop" not in model_options:
                cap.ident = (self.latent_image, latent_mask, model_options, x.shape, n_steps, seed, self.rng,
                             self.latent_image.data_ptr(), latent_mask.data_ptr(), getattr(latent_mask, "_lp_bits", None),
                             getattr(latent_mask, "_lp_u8", None), x.device, sigma.numel(), current_times[0].numel(),
                             current_times[1].numel(), current_times[2].numel(), self._override_state(), self._hyper_key())
                self._last_cap = cap
            return self._replay_fast(cap, x, sigma, current_times)
        self._last_cap = None
        st = self._prologue(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW, ws=cap.ws,
                            ds=cap.es["ds"] if cap.es is not None else None)
        cap.graph.replay()
        if self.rng == "torch":        # the replayed launches consumed this much of the generator's stream
            gen = self._generator(x.device)
            gen.set_offset(gen.get_offset() + cap.launches)
            self._torch_consumed += cap.launches
        out = self._epilogue(st, cap.final, rng_bump=(cap.counter, cap.launches) if self.rng == "philox" else None,
                             in_graph=cap.final_in_graph)
        if cap.es is not None and st.es is not None:
            self._es_after_replay(cap, st.es["seq"], x.device)
        return out

    def _replay_fast(self, cap, x, sigma, current_times):
        """Steady-state replay: the two launches around the graph (replace + coefficient table, lp_finalize) reuse
        the descriptors snapshotted at capture; only the caller's pointers (x, noise, sigma, times, out) change.
        With the raw hipGraphExec_t the whole sequence is ONE trip through the FFI (lp_replay_call)."""
        lib, stream = self._lib, self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        t_src = current_times[2] if cap.flow else ve
        out = torch.empty_like(x)
        k0, f = cap.k0_desc, cap.f_desc
        k0.x, k0.noise = x.data_ptr(), self.noise.data_ptr()
        # the replace launch also rebuilds the coefficient table from this call's sigma / times (LP_PH_COEFFS)
        k0.t_ve, k0.t_abt, k0.t_rsig, k0.t_model = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr(), t_src.data_ptr()
        if cap.final_in_graph:         # the captured lp_finalize reads these two through the table the replace publishes
            k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        else:
            f.x_dst, f.out = k0.x, out.data_ptr()
        if self.rng == "torch":        # generator state in (published by the replace launch), state out
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
            gen.set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        seq = 0
        if cap.es is not None:         # the replace launch resets the device-side stopper for this call
            seq = k0.es_seq_base = cap.es["ds"].next_seq()
        if cap.raw_exec is not None:
            _cabi.check(lib.lp_replay_call(ctypes.byref(cap.call), stream), "lp_replay_call")
        else:
            _cabi.check(lib.lp_step(ctypes.byref(k0), stream), "lp_step")
            cap.graph.replay()
            if not cap.final_in_graph:
                _cabi.check(lib.lp_finalize(ctypes.byref(f), stream), "lp_finalize")
        if cap.es is not None:
            self._es_after_replay(cap, seq, x.device)
        return out

    def _es_after_replay(self, cap, seq, dev):
        """The replayed loop decides its own length: remember what to collect.  With the trace requested, or with
        torch's generator to be left exactly where the reference leaves it, collect it now -- ONE host read per
        sigma call; otherwise at the next call / when iterations_run is read."""
        es = dict(cap.es, trace=self._es_opts["trace"] if self._es_opts is not None else None,
                  tags=self._es_opts["tags"] if self._es_opts is not None else (None, None, None))
        inc = self._randn_policy(dev, cap.ws.x_t.numel())[1] if self.rng == "torch" else 0
        self._es_pending = (cap.es["ds"], seq, cap.n_steps, es, dev, inc)
        if inc or es["trace"] is not None:
            self._es_resolve()

    def _rng_state(self, dev):
        """Device u64[4] read by captured launches.  rng="philox": [0] = launch-sequence base, ONE per device and
        bumped by every replay so the streams of different captures never overlap.  rng="torch": (generator
        offset, seed) published by the replace launch of each call.  [2], [3]: the I/O table of the call in flight
        (address of the sampler latent x, address of `out`), published by the replace launch for the captured
        lp_finalize.  [4]: the word that voids a captured lp_finalize when 0 (io_table word 2)."""
        state = self._rng_counters.get(dev)
        if state is None:
            state = self._rng_counters[dev] = torch.zeros(8, dtype=torch.int64, device=dev)
            state[4] = 1          # [4]: "this sigma call is valid" -- 0 voids a captured lp_finalize (a speculated call, lp_node_call)
        return state

    _capture_sentinels = {}

    @classmethod
    def _warm_capture_state(cls, dev):
        """Once per device, before this process's first capture through the engine: a trivial capture OUTSIDE inference mode whose
        graph object is then KEPT for the life of the process.  torch allocates the device generator's graph-capture state (its
        seed / offset tensors) when the FIRST graph registers with the generator, updates it in place at every `capture_begin`,
        and frees it again when the LAST registered graph dies.  Allocated under torch.inference_mode() -- how ComfyUI runs its
        nodes -- the state is inference tensors, and the first capture attempted outside inference mode while any graph is still
        alive dies inside capture_begin on that in-place update, leaving the generator in its capturing state (every later
        torch.randn of the process then raises "Offset increment outside graph capture").  Found by the property test of the
        capture state machine (tests/test_gpu_state_machine.py) -- twice: a warm-up capture that was freed again only moved the
        hazard to the next moment no graph was alive.  With one sentinel graph registered from normal mode and never freed, the
        state stays allocated as normal tensors, which either mode may update."""
        if dev.index in cls._capture_sentinels:
            return
        cls._capture_sentinels[dev.index] = None
        try:
            with torch.inference_mode(False), torch.no_grad():
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    torch.zeros(1, device=dev)
                torch.cuda.current_stream(dev).wait_stream(s)
                cls._capture_sentinels[dev.index] = g
        except Exception:          # (a torch build that refuses: the engine's own captures will say why)
            pass

    def _capture(self, key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                 replace_in_graph=None):
        """Capture one sigma call.  `replace_in_graph` (default: on, LANPAINT_AMD_REPLACE_IN_GRAPH=0 turns it off): the
        replace launch is captured too, as the FIRST node, and every replay refreshes that node's arguments (the caller's
        x / noise / sigma / times, this call's out, generator state) with hipGraphExecKernelNodeSetParams -- the whole
        sigma call is then ONE hipGraphLaunch with nothing eager in front of it (lp_replay_call, replace_binding).  Needs the
        raw graph handles and a call the steady-state path takes (dense fp32 tensors, a fusable replace form); anything
        else is captured the round-2 way, with the replace launch outside the graph."""
        dev = x.device
        self._warm_capture_state(dev)
        if replace_in_graph is None:
            replace_in_graph = os.environ.get("LANPAINT_AMD_REPLACE_IN_GRAPH", "1") != "0"
        raw_ok = self.rng in ("philox", "torch") and os.environ.get("LANPAINT_AMD_RAW_GRAPH", "1") != "0"
        replace_in_graph = bool(replace_in_graph and raw_ok)
        counter = self._rng_state(dev)
        cap = _CapturedCall(counter)
        if replace_in_graph:
            try:
                cap.graph = torch.cuda.CUDAGraph(keep_graph=True)  # the hipGraph_t stays: node 0 has to be found in it
            except (TypeError, RuntimeError):                      # a torch build without `keep_graph`: the round-2 layout
                replace_in_graph = False
        # captures that differ only in the step count (KSamplerX0Inpaint's n_eff ramp) share ONE workspace: sigma
        # calls are serialised on the stream, and the n_steps-independent replace launch can then be enqueued
        # before the count is known (begin_call / finish_call)
        ws_key = (tuple(x.shape), dev.index, self.model_dtype)
        cap.ws = self._static_ws.get(ws_key)
        if cap.ws is None:
            if len(self._static_ws) >= self.MAX_GRAPHS:
                self._static_ws.pop(next(iter(self._static_ws)))
            cap.ws = self._static_ws[ws_key] = _Workspace(x.detach().to(torch.float32).contiguous(), static_io=True,
                                                          model_dtype=self.model_dtype)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        it0 = self._iterations_run
        rng_state = torch.cuda.get_rng_state(dev)   # warm-up + capture must not consume the user's torch stream
        gen = self._generator(dev)
        off0, own0 = gen.get_offset(), self._torch_consumed
        es_user = self._es_opts
        if es_user is not None:                    # the warm-up below is not the caller's run: keep it out of their trace
            self._es_opts = dict(es_user, trace=None)
            # without a trace to fill nobody needs the verdict of the loop's LAST iteration: its launch closes the loop
            # itself (LP_FL_ES_CLOSE) and the closing decision kernel -- one more graph node -- is not captured
            self._es_close = es_user["trace"] is None
        with torch.cuda.stream(side):              # one complete eager call on the side stream: lazy inits
            xw = x.detach().clone()
            st = self._prologue(xw, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                ws=cap.ws)
            self._epilogue(st, self._think_and_final_model(st, model_options, seed))
            # the state the captured launches start from.  With the replace launch inside the graph it is only
            # described here (descriptor snapshot) and enqueued as the first captured launch below.
            st = self._prologue(xw, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                ws=cap.ws, defer_launch=replace_in_graph)
            if replace_in_graph and not (st.k0_desc is not None and st.replace_kind_static and st.xc is st.input_x
                                         and (st.xc.data_ptr() & 15) == 0 and (self.noise.data_ptr() & 15) == 0):
                replace_in_graph = False           # not a call the steady-state path takes: the replace stays outside
                self._launch_step_desc(st.k0_desc, st.stream)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # did the warm-up (backbone included) draw from torch's generator?  Then only torch's own replay() keeps
        # the captured Philox offsets moving and the graph must not be launched behind its back.
        torch_rng_used = (gen.get_offset() - off0) != (self._torch_consumed - own0)
        if torch_rng_used and self.rng == "torch":
            # the engine's in-kernel draws and the backbone's own draws would have to interleave inside the graph
            # exactly as they do eagerly: not representable with one published offset -> this engine stays eager
            torch.cuda.set_rng_state(rng_state, dev)
            self._iterations_run = it0
            self._graph_blocked = True
            self._es_opts = es_user
            return None
        if torch_rng_used and replace_in_graph:
            # (the graph has to go through torch's replay(), which knows nothing of node arguments: capture again the
            # round-2 way)
            torch.cuda.set_rng_state(rng_state, dev)
            self._iterations_run = it0
            self._es_opts = es_user
            return self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                 replace_in_graph=False)
        self._iterations_run = it0
        self._capturing, self._cap_offset = counter, 0
        # No cyclic garbage collection while the stream is capturing: a collection that happens to run inside the captured region
        # finalises whatever unreachable objects the process holds -- among them other engines' captures, whose destructors destroy
        # hipGraph / hipGraphExec handles and free their memory pools.  Those calls are illegal on a capturing thread; the error
        # surfaces in a C++ destructor and aborts the process ("Fatal Python error: Aborted ... Garbage-collecting", met once in
        # the 400-sequence property test, which leaves hundreds of dead engines behind).  torch collects BEFORE a capture; it
        # cannot stop the collector from firing during one.  (Reference counting still frees what drops to zero: tensors go back
        # to the caching allocator, which knows about captures.)
        import gc
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            # thread_local: a live RCCL communicator's watchdog thread issues HIP calls of its own;
            # in the default "global" mode those would invalidate this thread's capture
            f = _cabi.LpFinalDesc()
            with torch.cuda.graph(cap.graph, stream=side, capture_error_mode="thread_local"):
                if replace_in_graph:
                    self._launch_step_desc(st.k0_desc, self._stream(dev))
                cap.final = self._think_and_final_model(st, model_options, seed)
                dense_ok = self._fill_final_desc(f, st, cap.final, st.out)
                if self.rng == "philox":
                    f.rng_bump_ptr, f.rng_bump = counter.data_ptr(), self._cap_offset
                if dense_ok and st.out is not None:
                    # the finalise is the last node of the graph: it takes the caller's x and this call's `out`
                    # from the table the replace launch of the same call publishes
                    f.io_table = counter.data_ptr() + 16
                    _cabi.check(self._lib.lp_finalize(ctypes.byref(f), self._stream(dev)), "lp_finalize")
                    cap.final_in_graph = True
        finally:
            self._capturing = None
            if gc_was_enabled:
                gc.enable()
        cap.launches = self._cap_offset
        self._es_opts = es_user
        torch.cuda.set_rng_state(rng_state, dev)
        cap.ran = self._iterations_run - it0
        self._iterations_run = it0
        cap.keep = st                              # descriptor-side tensors referenced by the baked launches
        # descriptors of the launches whose arguments change from call to call, for the steady-state replay path
        cap.rows, cap.flow = st.rows, st.flow
        cap.hyper = _cabi.LpHyper.from_buffer_copy(self._hyper)
        cap.k0_desc = st.k0_desc
        cap.f_desc = f
        cap.model_options = model_options
        cap.es, cap.n_steps = st.es, st.n_steps
        cap.fast = bool(dense_ok and st.k0_desc is not None and st.replace_kind_static and st.xc is st.input_x)
        if replace_in_graph:
            ok = cap.fast and cap.final_in_graph
            if ok:
                try:
                    cap.graph.instantiate()
                    raw_graph = int(cap.graph.raw_cuda_graph())
                    cap.raw_exec = int(cap.graph.raw_cuda_graph_exec()) or None
                    b = _cabi.LpGraphBinding()
                    ok = cap.raw_exec is not None and self._lib.lp_graph_bind_replace(
                        raw_graph, ctypes.byref(cap.k0_desc), ctypes.byref(b)) == _cabi.LP_OK
                    if ok:
                        cap.binding = b
                        tg, te = ctypes.c_void_p(), ctypes.c_void_p()
                        if self._lib.lp_graph_clone_tail(raw_graph, ctypes.byref(tg), ctypes.byref(te)) == _cabi.LP_OK:
                            cap.tail_handles = (tg.value, te.value)
                except Exception:
                    ok = False
            if not ok:       # not a steady-state call after all, or this runtime does not give the handles: round-2 layout
                return self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX,
                                     IS_FLOW, replace_in_graph=False)
        if cap.fast:
            if cap.binding is None and raw_ok and not torch_rng_used:
                try:
                    cap.raw_exec = int(cap.graph.raw_cuda_graph_exec()) or None
                except Exception:
                    cap.raw_exec = None
            c = cap.call = _cabi.LpCallDesc()          # hyper = NULL: no separate lp_coeffs launch, the replace does it
            c.replace = ctypes.pointer(cap.k0_desc)
            c.final = None if cap.final_in_graph else ctypes.pointer(f)
            c.rows, c.coef_table, c.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.raw_exec
            if cap.binding is not None:
                c.replace_binding = ctypes.pointer(cap.binding)
                if cap.tail_handles is not None:
                    t = cap.tail = _cabi.LpCallDesc()  # the graph minus node 0: its replace launch went ahead (begin_call)
                    t.rows, t.coef_table, t.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.tail_handles[1]
            elif cap.raw_exec is not None and cap.final_in_graph:
                t = cap.tail = _cabi.LpCallDesc()      # the graph alone: its replace launch went ahead (begin_call)
                t.rows, t.coef_table, t.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.raw_exec
        cap.key = key
        self._graphs[key] = cap
        return cap

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

    # ------------------------------------------------------------------ loops off the fast path
    def _x0s_buffer(self, ws, exclude):
        """A rotating x0s buffer not aliased by any tensor in `exclude`."""
        ptrs = {t.data_ptr() for t in exclude if t is not None}
        for buf in ws.x0s:
            if buf.data_ptr() not in ptrs:
                return buf
        buf = torch.empty_like(ws.x_t)
        ws.x0s.append(buf)
        return buf

    def _loop_es(self, st, n_steps, model_options, seed):
        """Inner early stop with the default metric, evaluated on the device (LP_FL_ES): the POST launch of every
        iteration also reduces the weighted MSEs of earlystop.py:279-306 and applies the threshold / patience /
        drift-anchor rule in its last block.
        Eager: the host reads the verdict from the pinned mailbox once per iteration and leaves the loop like the
        reference does (no backbone call is wasted); the PRE half-step of the next iteration is a launch of its own.
        Captured (hipGraph): nobody watches -- the launches are gated on the device-side flag (LP_FL_ES_GATED), keep
        the fused one-launch-per-iteration shape, and the backbone calls after the stop still run (their results are
        ignored).  Returns the iterations run (0 while capturing: the replay reports it, _es_resolve)."""
        d, ws, shape, base_flags, stream = self._desc, st.ws, st.shape, st.base_flags, st.stream
        es, ds = st.es, st.es["ds"]
        d.es_n_steps = n_steps
        gated = self._capturing is not None
        ran = 0
        for i in range(n_steps):
            last = i == n_steps - 1
            if i > 0 and not gated:                   # first half-step of iteration i, committed (lanpaint.py:280)
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
            output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
            if gated:
                close = LP_FL_ES_CLOSE if (last and self._es_close) else 0
                alive = self._set_model_output(d, output, base_flags | LP_FL_ES | LP_FL_ES_GATED | es["ring_flag"] | close | self._emit(st, last), shape)
                d.phases = (LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY) | (0 if last else LP_PH_PRE_HALF) | LP_PH_EMIT
                self._set_xi(d, ws.x_t, want_pre=not last)
            else:
                alive = self._set_model_output(d, output, base_flags | LP_FL_ES | es["ring_flag"], shape)
                d.phases = LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY
                self._set_xi(d, ws.x_t, want_pre=False)
            d.es_index = i
            self._launch_step(stream)
            del alive
            if gated:
                continue
            ds.wait(es["seq"] + i + 1, st.xc.device)
            ran += 1
            self._es_trace(es, ds, i)
            if ds.f64[2] != 0.0:                      # stopped
                break
            if i == 0 and ds.f64[3] == 0.0 and not last:
                # The stopper can never fire in this call (threshold_eff <= 0 at this abt, or nothing to inpaint:
                # earlystop.py:111-117 -- the reference's from_options returns None and runs its plain loop).  Do the same:
                # no more verdicts to wait for, no early-stop streams, one fused launch per iteration from here on.
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
                for j in range(1, n_steps):
                    last_j = j == n_steps - 1
                    output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
                    alive = self._set_model_output(d, output, base_flags | self._emit(st, last_j), shape)
                    d.phases = LP_PH_POST_STEADY | (0 if last_j else LP_PH_PRE_HALF) | LP_PH_EMIT
                    self._set_xi(d, ws.x_t, want_pre=not last_j)
                    self._launch_step(stream)
                    del alive
                ds.seen_total += 1                # (the device counted iteration 0 only)
                return n_steps
        if not gated:
            ds.seen_total += ran                  # this loop's iterations are accounted by its caller
            d.phases, d.flags = LP_PH_EMIT, base_flags | self._emit(st, True)
            self._launch_step(stream)
        return ran

    def _loop_unfused(self, st, n_steps, model_options, seed, stopper):
        """Early stop enabled: the stopper decides after every iteration, so the POST
        half of iteration i cannot be fused with the PRE half of iteration i+1."""
        d, ws, shape, base_flags, stream = self._desc, st.ws, st.shape, st.base_flags, st.stream
        args = None
        ran = 0
        for i in range(n_steps):
            x_t_before = ws.x_t.clone() if args is None or stopper.has_custom_distance_fn else None
            if i > 0:
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
            output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
            x0s = self._x0s_buffer(ws, [args.x0 if args else None, stopper.anchor])
            alive = self._set_model_output(d, output, base_flags | LP_FL_WRITE_X0S, shape)
            d.x0s = x0s.data_ptr()
            d.phases = LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY
            self._set_xi(d, ws.x_t, want_pre=False)
            self._launch_step(stream)
            del alive
            prev_args, args = args, LangevinState(None, ws.C, x0s)
            ran += 1
            ctx = {"step": i, "steps_done": i + 1, "n_steps": n_steps, "mask": st.m, "latent_image": st.y,
                   "current_times": st.current_times, "seed": seed}
            if stopper.observe(i, x_before=x_t_before, x_after=ws.x_t,
                               x_prev_for_user=x_t_before if stopper.has_custom_distance_fn else None,
                               x0_prev=prev_args.x0 if prev_args is not None else None, x0_cur=args.x0, ctx=ctx):
                break
        d.x0s = None
        d.phases, d.flags = LP_PH_EMIT, base_flags | self._emit(st, True)
        self._launch_step(stream)
        return ran

    def _loop_compat(self, ws, shape, m, y, abt, current_times, n_steps, model_options, seed, stopper):
        """A subclass / instance overrides langevin_dynamics, score_model or
        prepare_step_size: run the reference's per-iteration call structure
        (lanpaint.py:113-142) so the overrides see the calls they expect."""
        abt_b = self.add_none_dims(abt)
        step_size = self.add_none_dims(self.step_size * (1 - abt).clamp(min=self.min_step_frac))
        x_t, args, ran = ws.x_t, None, 0
        for i in range(n_steps):
            score_func = partial(self.score_model, y=y, mask=m, abt=abt_b, sigma=self.add_none_dims(current_times[0]),
                                 tflow=self.add_none_dims(current_times[2]), model_options=model_options, seed=seed)
            prev_args = args
            x_prev = x_t.detach().clone() if (stopper is not None and stopper.has_custom_distance_fn) else None
            x_before = x_t.detach().clone() if stopper is not None else None
            x_t, args = self.langevin_dynamics(x_t, score_func, m, step_size, current_times,
                                               sigma_x=self.add_none_dims(self.sigma_x(abt)),
                                               sigma_y=self.add_none_dims(self.sigma_y(abt)), args=args)
            ran += 1
            if stopper is not None:
                ctx = {"step": i, "steps_done": i + 1, "n_steps": n_steps, "mask": m, "latent_image": y,
                       "current_times": current_times, "seed": seed}
                if stopper.observe(i, x_before=x_before, x_after=x_t, x_prev_for_user=x_prev, x0_prev=_state_x0(prev_args),
                                   x0_cur=_state_x0(args), ctx=ctx):
                    break
        if x_t.data_ptr() != ws.x_t.data_ptr():
            ws.x_t.copy_(x_t)
        return ran

    def langevin_dynamics(self, x_t, score, mask, step_size, current_times, sigma_x=1, sigma_y=0, args=None):
        """Public single-iteration entry with the reference signature (lanpaint.py:192-293):
        `score` is any callable x_t -> score tensor.  Returns (x_t_new, LangevinState)."""
        if args is not None and not isinstance(args, LangevinState) and isinstance(args, tuple):
            if len(args) == 2:
                args = LangevinState(args[0], args[1], None)
            elif len(args) >= 3:
                args = LangevinState(args[0], args[1], args[2])
        if not x_t.is_cuda:
            raise RuntimeError("lanpaint_amd.LanPaint.langevin_dynamics runs on a HIP device only; no CPU fallback")
        if self.img_dim_size is None:
            self.img_dim_size = x_t.ndim
        step_sizes = self.prepare_step_size(current_times, step_size, sigma_x, sigma_y)
        _sig, abt_b, dtx = step_sizes[0], step_sizes[1], step_sizes[2]
        if torch.mean(dtx) <= 0.0:                                   # lanpaint.py:205
            return x_t, args
        lib = self._lib
        shape, rows = x_t.shape, x_t.shape[0]
        xt = _as_f32c(x_t).clone()
        mk = _as_f32c(mask if mask.shape == shape else mask.expand(shape))
        stream = self._stream(xt.device)
        flow = bool(self.IS_FLUX or self.IS_FLOW)
        sx = torch.as_tensor(sigma_x, dtype=torch.float32, device=xt.device)
        sy = torch.as_tensor(sigma_y, dtype=torch.float32, device=xt.device)
        step_t = torch.as_tensor(step_size, dtype=torch.float32, device=xt.device)
        if sx.numel() > 1 and bool((sx != sx.reshape(-1)[0]).any()) or sy.numel() > 1 and bool((sy != sy.reshape(-1)[0]).any()):
            raise NotImplementedError("non-uniform sigma_x / sigma_y are not supported by the HIP path")
        sx0, sy0 = float(sx.reshape(-1)[0]), float(sy.reshape(-1)[0])
        VE_Sigma, abt, _ft = current_times
        d = _cabi.LpStepDesc()
        hyp = _cabi.LpHyper()
        hyp.lambda_, hyp.step_size, hyp.min_step_frac = float(self.chara_lamb), float(self.step_size), 0.0
        hyp.beta = (sy0 / sx0) if sx0 != 0.0 else 0.0
        hyp.is_flow, hyp.one_plus_lambda = int(flow), 1.0 + float(self.chara_lamb)
        base = (LP_FL_FLOW if flow else 0) | LP_FL_X0S_GIVEN | LP_FL_WRITE_X0S
        d.n_el, d.el_per_row, d.rows = xt.numel(), xt.numel() // rows, rows
        d.lambda_, d.one_plus_lambda, d.beta = hyp.lambda_, hyp.one_plus_lambda, hyp.beta
        d.noise_scale = 1.0
        d.mask, d.x_t = mk.data_ptr(), xt.data_ptr()
        if abt.numel() not in (1, rows) or step_t.numel() not in (1, rows):
            # per-element times (AV packs): the kernel derives the step from abt itself,
            # StepSize*max(1-abt, MinStepFrac) -- what the engine passes as `step_size` here
            base |= LP_FL_PER_ELEMENT
            abt_el = _as_f32c(self.add_none_dims(abt).expand(shape))
            ve_el = _as_f32c(self.add_none_dims(VE_Sigma).expand(shape))
            d.abt_el, d.ve_el, d.coef = abt_el.data_ptr(), ve_el.data_ptr(), None
            d.step_size, d.min_step_frac = float(self.step_size) * sx0, float(self.min_step_frac)
        else:
            coef = torch.empty((rows, _cabi.LP_COEF_STRIDE), dtype=torch.float32, device=xt.device)
            ve_r, abt_r = _as_f32c(VE_Sigma.reshape(-1)), _as_f32c(abt.reshape(-1))
            step_r = _as_f32c((step_t * sx0).reshape(-1))
            _cabi.check(lib.lp_coeffs(ctypes.byref(hyp), ve_r.data_ptr(), int(ve_r.numel() > 1), abt_r.data_ptr(),
                                      int(abt_r.numel() > 1), None, 0, step_r.data_ptr(), int(step_r.numel() > 1), None, 0,
                                      rows, coef.data_ptr(), stream), "lp_coeffs")
            d.step_size, d.min_step_frac = hyp.step_size, 0.0
            d.coef = coef.data_ptr()
        d.rng_seed = int(self.philox_seed or 0)
        if args is None:
            c_buf = torch.empty_like(xt)
            d.C = c_buf.data_ptr()
        else:
            c_buf = _as_f32c(args.C).clone()
            d.C = c_buf.data_ptr()
            d.phases, d.flags = LP_PH_PRE_HALF, base                # first half-step with the old C
            self._set_xi(d, xt, want_pre=True, want_post=False)
            self._launch_step_desc(d, stream)
        x0s_in = _as_f32c(xt + score(xt))                            # Coef_C: x0 = x_t + score(x_t)
        x0s_out = torch.empty_like(xt)
        d.x0, d.x0_big, d.x0s = x0s_in.data_ptr(), x0s_in.data_ptr(), x0s_out.data_ptr()
        d.phases, d.flags = (LP_PH_POST_FIRST if args is None else LP_PH_POST_STEADY), base
        self._set_xi(d, xt, want_pre=False)
        self._launch_step_desc(d, stream)
        return xt.to(x_t.dtype), LangevinState(None, c_buf, x0s_out)

    def _launch_step_desc(self, d, stream):
        _cabi.check(self._lib.lp_step(ctypes.byref(d), stream), "lp_step")


# the overridable methods as defined in this module (LanPaint._override_state / _overridden compare against these)
_OWN_NAMES = ("langevin_dynamics", "score_model", "prepare_step_size")
_OWN_METHODS = tuple(LanPaint.__dict__[n] for n in _OWN_NAMES)
