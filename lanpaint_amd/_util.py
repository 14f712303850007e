"""Small host-side helpers shared by the engine's modules: stream / tensor-version accessors, dtype normalisation, the capture-time
garbage-collector hold, ATen's randn launch policy, recognition of a model_sampling's noise_scaling form."""
from __future__ import annotations

import threading

import torch

from .types import LangevinState

_GC_LOCK = threading.Lock()
_GC_HOLDERS = [0, False]          # [captures in flight in this process, was the collector enabled when the first one began]


def _gc_hold():
    """No cyclic collection while ANY thread of the process is capturing (see LanPaint._capture): the first capture to begin
    switches the collector off, the last one to end puts it back the way it was -- two engines capturing on two threads cannot
    re-enable it under each other."""
    import gc
    with _GC_LOCK:
        if _GC_HOLDERS[0] == 0:
            _GC_HOLDERS[1] = gc.isenabled()
            gc.disable()
        _GC_HOLDERS[0] += 1


def _gc_release():
    import gc
    with _GC_LOCK:
        _GC_HOLDERS[0] = max(0, _GC_HOLDERS[0] - 1)
        if _GC_HOLDERS[0] == 0 and _GC_HOLDERS[1]:
            gc.enable()


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
