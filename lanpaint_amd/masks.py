"""Compact forms of the binary tensors the think loop streams: the bit-packed latent mask (LP_FL_MASK_BITS, 1 bit per element
instead of the reference's 4-byte fp32 mask) and the bit-packed stream indicator of AV packs -- made once per job, kept
current when the tensor they were made from is rewritten in place."""
from __future__ import annotations

import weakref

import torch

from . import _cabi
from ._cabi import LP_FL_MASK_BITS, LP_FL_MASK_U8
from ._util import _as_f32c, raw_stream, tensor_version

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
    # what the PREVIOUS re-pack of a mask packed from its own tensor found (the kernel raises its "values other than 0 and 1" flag
    # straight into pinned host memory: a cheap read, no sync) is looked at on EVERY call, before the version shortcut -- a single
    # in-place rewrite to soft values must not be binarised in the bits for the rest of the job without a word.  Once seen, the
    # packed copy is dropped for good: the engine goes on with the plain fp32 mask (the soft-mask arithmetic of the reference).
    soft = getattr(packed, "_lp_soft_flag", None)
    if soft is not None and int(soft[0]) != 0:
        for attr in ("_lp_bits", "_lp_bits_of", "_lp_soft_flag", "_lp_auto"):
            if hasattr(packed, attr):
                delattr(packed, attr)
        raise ValueError("pack_mask: the packed mask was rewritten in place to values other than 0 and 1; soft masks cannot "
                         "be bit-packed.  The packed copy has been dropped: later calls use the plain fp32 mask")
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
            if soft is None:
                soft = packed._lp_soft_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
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
