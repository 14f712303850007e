"""Post-decode mask blend on the GPU (pixel space, once per job).

Mirrors the reference's `MaskBlend` node (src/LanPaint/nodes.py:592-647), `merge_video_with_mask`
(nodes.py:1060-1088) and `gaussian_kernel_2d` (nodes.py:1049-1057): dilate the mask with a
`blend_overlap`-wide max-pool, smooth it with the Gaussian of the same width, then
`before * (1 - m) + after * m`.  One fused HIP launch (lp_mask_blend: LDS tile, separable passes)
instead of max_pool2d + conv2d + 4 elementwise passes.  HIP tensors only, no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _cabi, interp_rule


def gaussian_kernel_2d(kernel_size):
    """nodes.py:1049-1057 (host helper kept for API parity; the kernel builds the separable profile itself)."""
    if kernel_size <= 1:
        return torch.ones(1, 1)
    sigma = (kernel_size - 1) / 4
    x = torch.arange(kernel_size).float() - kernel_size // 2
    xg, yg = torch.meshgrid(x, x, indexing="ij")
    k = torch.exp(-(xg ** 2 + yg ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def _launch(mask, image1, image2, k, want_smooth=False, nn_rule=0):
    if not image1.is_cuda:
        raise RuntimeError("lanpaint_amd.blend runs on a HIP device only; no CPU fallback")
    if not isinstance(k, int) or k < 1 or k > 51 or k % 2 == 0:
        raise ValueError(f"blend_overlap must be an odd integer in [1, 51], got {k!r}")
    lib = _cabi.load()
    dev = image1.device
    i1, i2 = _f32c(image1), _f32c(image2.to(dev))
    m = _f32c(mask.to(dev))
    b, h, w, c = i1.shape
    out = torch.empty_like(i1)
    smooth = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_smooth else None
    d = _cabi.LpBlendDesc()
    d.batch, d.height, d.width, d.channels, d.k = b, h, w, c, k
    d.mask_batch, d.mask_h, d.mask_w = m.shape[0], m.shape[1], m.shape[2]
    d.nn_rule = int(nn_rule)
    d.mask, d.image1, d.image2, d.out = m.data_ptr(), i1.data_ptr(), i2.data_ptr(), out.data_ptr()
    d.smooth_out = smooth.data_ptr() if smooth is not None else None
    with torch.cuda.device(dev):
        _cabi.check(lib.lp_mask_blend(ctypes.byref(d), torch.cuda.current_stream(dev).cuda_stream), "lp_mask_blend")
    return (out, smooth) if want_smooth else out


def mask_blend(image1, image2, mask, blend_overlap):
    """MaskBlend.blend_images (nodes.py:610-638): images [B, H, W, C], mask [B, H, W]."""
    if image1.shape[1] != image2.shape[1] or image1.shape[2] != image2.shape[2]:
        raise ValueError(
            "Image size mismatch: Image1 and Image2 must have the same dimensions.\n"
            "Additionally, ensure both images have width and height that are multiples of 8 (VAE decode always "
            "produces such sizes).\nCurrent sizes - Image1: {}x{}, Image2: {}x{}".format(
                image1.shape[2], image1.shape[1], image2.shape[2], image2.shape[1]))
    m = mask.float()
    if m.shape[0] not in (1, image1.shape[0]) or tuple(m.shape[1:]) != tuple(image1.shape[1:3]):
        raise ValueError(f"mask shape {tuple(mask.shape)} does not match images {tuple(image1.shape)}")
    return _launch(m, image1, image2, blend_overlap)


def merge_video_with_mask(orig, inpainted, mask, blend_overlap):
    """nodes.py:1060-1088: frames at batch; the mask may be [F,H,W], [F,1,H,W] or a single [H,W] image
    mask, at the image resolution or lower (then resampled nearest-exact inside the kernel)."""
    m = mask.float()
    if m.ndim == 4:
        m = m[:, 0]
    elif m.ndim == 2:
        m = m.unsqueeze(0)
    count = min(orig.shape[0], inpainted.shape[0])
    orig, inpainted = orig[:count], inpainted[:count]
    if m.shape[0] == 1:
        m = m[:1]
    elif m.shape[0] < count:
        raise ValueError("the mask has fewer frames than the image")
    else:
        m = m[:count]
    # the reference resamples a lower-resolution mask with F.interpolate on the MASK's device (nodes.py:1078-1081: a 2-D call on
    # [F, 1, h, w]): the kernel follows the index rule of the torch kernel that call would have run
    rule = interp_rule.rule_for(mask, m.unsqueeze(1), tuple(orig.shape[1:3]))
    return _launch(m, orig, inpainted, blend_overlap, nn_rule=rule)


class MaskBlend:
    """ComfyUI node with the reference's protocol (nodes.py:592-608)."""

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {
            "image1": ("IMAGE", {"tooltip": "Image before inpaint"}),
            "image2": ("IMAGE", {"tooltip": "Image after inpaint"}),
            "mask": ("MASK",),
            "blend_overlap": ("INT", {"default": 1, "min": 1, "max": 51, "step": 2,
                                      "tooltip": "The number of pixels to blend between the two images."}),
        }}

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "blend_images"
    CATEGORY = "image/postprocessing"

    def blend_images(self, image1, image2, mask, blend_overlap):
        dev = image1.device if image1.is_cuda else torch.device("cuda", torch.cuda.current_device())
        out = mask_blend(image1.to(dev), image2.to(dev), mask.to(dev), blend_overlap)
        return (out.to(image1.device),)

    def gaussian_kernel(self, kernel_size):
        return gaussian_kernel_2d(kernel_size)
