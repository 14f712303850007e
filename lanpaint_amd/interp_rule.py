"""Which nearest-exact source-index rule the REFERENCE's `F.interpolate(mode="nearest-exact")` call follows.

The reference resamples masks with torch on whatever device the mask tensor lives on -- for a ComfyUI workflow the host:
`prepare_mask` calls `reshape_mask(...)` BEFORE `.to(device)` (nodes.py:159-160), and the image / video merge nodes work on
node tensors (nodes.py:1079, 1278-1287).  ATen does not have ONE nearest-exact index formula: its GPU kernels and part of its
CPU kernels evaluate `floorf((i + 0.5f) * scale)`, the CPU's TensorIterator kernel evaluates
`floorf(float(double(max(scale * (i + 0.5f) - 0.5f, 0)) + 0.5))` with the multiply-subtract contracted to an FMA in the AVX2 /
AVX512 builds.  They agree on every down-sampling pair and differ on ties when up-sampling (audio `[F]` -> tokens, a
low-resolution mask merged into a full-resolution video).  BASELINE.json asks for bit-exact mask index math, so the HIP
kernels (lp_reshape_mask, lp_mask_blend) implement all three forms (include/lanpaint_hip.h: LP_NN_ATEN_*) and this module
picks the one the reference's call would have taken -- ATen's own dispatch conditions (aten/src/ATen/native/cpu/
UpSampleKernel.cpp: upsample_nearest_exact{1,2,3}d_kernel_impl, _use_vectorized_kernel_cond_2d), restated and checked against
torch on the CPU over every (in, out) <= 512 in tests/test_oracle_properties.py.  No resampling happens here.
"""
from __future__ import annotations

import torch

from ._cabi import LP_NN_ATEN_CPU_GENERIC, LP_NN_ATEN_CPU_GENERIC_FMA, LP_NN_ATEN_SCALAR

_generic = None


def cpu_generic_rule() -> int:
    """The form this torch build's CPU TensorIterator kernel takes on THIS host: contracted (LP_NN_ATEN_CPU_GENERIC_FMA) or
    not (LP_NN_ATEN_CPU_GENERIC; `ATEN_CPU_CAPABILITY=default`, a CPU without FMA).  Decided once by asking torch for the
    one tie that tells them apart -- 2 -> 41 elements, 1-D: output 20 reads source 1 with the contracted form, 0 without
    (a 41-element host op; nothing of a mask is resampled on the host)."""
    global _generic
    if _generic is None:
        try:
            probe = torch.nn.functional.interpolate(torch.tensor([[[0.0, 1.0]]]), size=(41,), mode="nearest-exact")
            _generic = LP_NN_ATEN_CPU_GENERIC_FMA if float(probe[0, 0, 20]) == 1.0 else LP_NN_ATEN_CPU_GENERIC
        except Exception:
            cap = str(getattr(torch.backends.cpu, "get_cpu_capability", lambda: "AVX2")()).upper()
            _generic = LP_NN_ATEN_CPU_GENERIC if cap in ("DEFAULT", "NO AVX") else LP_NN_ATEN_CPU_GENERIC_FMA
    return _generic


def aten_rule(on_cuda: bool, spatial_dims: int, out_sizes, channels: int = 1, channels_last: bool = False) -> int:
    """ATen's dispatch for one nearest-exact call: GPU kernels -> scalar rule; CPU: channels-last with more than 3 channels ->
    scalar (cpu_upsample_nearest_channels_last); 2-D with out_h + out_w <= 128 -> scalar (_use_vectorized_kernel_cond_2d);
    everything else (1-D, 3-D, larger 2-D outputs) -> the TensorIterator kernel."""
    if on_cuda:
        return LP_NN_ATEN_SCALAR
    if channels_last and channels > 3 and spatial_dims in (2, 3):
        return LP_NN_ATEN_SCALAR
    if spatial_dims == 2 and int(out_sizes[-2]) + int(out_sizes[-1]) <= 128:
        return LP_NN_ATEN_SCALAR
    return cpu_generic_rule()


def rule_for(source: torch.Tensor, view: torch.Tensor, out_sizes) -> int:
    """Rule for the reference's `F.interpolate(view, size=out_sizes, mode="nearest-exact")`, where `view` is the [N, C, *spatial]
    tensor the reference would pass (any device: only its shape / memory format are looked at) and `source` the caller's
    own tensor, whose DEVICE decides which of torch's kernels the reference would have run."""
    nd = len(out_sizes)
    cl = False
    if view.ndim == nd + 2 and nd in (2, 3) and view.shape[1] > 3:
        fmt = torch.channels_last if nd == 2 else torch.channels_last_3d
        cl = view.is_contiguous(memory_format=fmt)
    return aten_rule(source.is_cuda, nd, out_sizes, int(view.shape[1]) if view.ndim >= 2 else 1, cl)
