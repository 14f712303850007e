"""One launch of lp_reshape_mask (nearest-exact resample + optional temporal union) on device tensors: the primitive under
nodes.reshape_mask and the image nodes' mask snap."""
from __future__ import annotations

import torch

from . import _cabi


def _hip_device(t, device=None):
    if device is not None and torch.device(device).type == "cuda":
        return torch.device(device)
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("lanpaint_amd.reshape_mask runs on a HIP device only; no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _resample(src5, out_b, out_c, out_f, out_h, out_w, taps, rule=0):
    """One lp_reshape_mask launch: src5 is [B', C', F, H, W] fp32 on the device; `rule`: the LP_NN_ATEN_* source-index rule
    (interp_rule.rule_for: the one torch's kernel follows on the device the reference would have resampled on)."""
    lib = _cabi.load()
    sb, sc, sf, sh, sw = src5.shape
    dst = torch.empty((out_b, out_c, out_f, out_h, out_w), dtype=torch.float32, device=src5.device)
    with torch.cuda.device(src5.device):
        _cabi.check(lib.lp_reshape_mask(src5.data_ptr(), sb, sc, sf, sh, sw, dst.data_ptr(), out_b, out_c, out_f, out_h,
                                        out_w, taps, int(rule) << _cabi.LP_RESHAPE_RULE_SHIFT,
                                        torch.cuda.current_stream(src5.device).cuda_stream),
                    "lp_reshape_mask")
    return dst
