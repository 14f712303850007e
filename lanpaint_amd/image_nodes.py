"""Image encode / decode nodes: the callers either side of the mask snap (lp_reshape_mask) and of the post-decode merge
(lp_mask_blend); reference nodes.py:1230-1342.  The VAE is the caller's object."""
from __future__ import annotations

import torch

from . import interp_rule
from .blend import merge_video_with_mask
from .resample import _hip_device, _resample


def _snap_mask_nearest_exact(mask_hw, out_h, out_w):
    """[H, W] mask -> [out_h, out_w], F.interpolate(mode="nearest-exact") semantics, on the HIP kernel; the result
    goes back to the mask's own device (node tensors normally live on the host)."""
    if tuple(mask_hw.shape) == (out_h, out_w):
        return mask_hw
    dev = _hip_device(mask_hw)
    src = mask_hw.to(device=dev, dtype=torch.float32).contiguous()
    rule = interp_rule.rule_for(mask_hw, src.reshape(1, 1, *src.shape), (out_h, out_w))      # nodes.py:1278-1287: 2-D call on the mask's device
    return _resample(src.reshape(1, 1, 1, *src.shape), 1, 1, 1, out_h, out_w, 1, rule)[0, 0, 0].to(mask_hw.device)


class LanPaint_ImageEncode:
    """nodes.py:1230-1290: VAE encode + attach the inpainting mask snapped (nearest-exact) to the latent's
    spatial size; 4-D image latents and 5-D video-VAE latents ([1, 1, T, H, W] mask)."""

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "image": ("IMAGE", {"tooltip": "The image to encode (1 = regenerate region comes from the mask)."}),
                "vae": ("VAE", {"tooltip": "The VAE."}),
            },
            "optional": {
                "mask": ("MASK", {"tooltip": "Inpainting mask [H, W] (1 = regenerate, 0 = keep). Snapped to the latent size automatically."}),
            },
        }

    RETURN_TYPES = ("LATENT",)
    RETURN_NAMES = ("latent",)
    FUNCTION = "encode"
    CATEGORY = "image"
    DESCRIPTION = "Encode an image and attach an inpainting mask to the latent (replaces VAEEncode + SetLatentNoiseMask)."

    def encode(self, image, vae, mask=None):
        z = vae.encode(image)
        ndim = len(z.shape)
        if ndim not in (4, 5):
            raise ValueError(f"LanPaint_ImageEncode expects a 4D or 5D latent, got {ndim}D")
        latent = {"samples": z}
        if mask is not None:
            m = mask.float()
            if m.ndim == 4:        # [1, 1, H, W] from SetLatentNoiseMask
                m = m[0, 0]
            elif m.ndim == 3:
                m = m[0]
            h, w = z.shape[-2:]
            m = _snap_mask_nearest_exact(m, h, w)
            if ndim == 4:
                latent["noise_mask"] = m.unsqueeze(0).unsqueeze(0)
            else:                  # one mask slice per latent frame
                latent["noise_mask"] = m.unsqueeze(0).unsqueeze(0).unsqueeze(2).expand(1, 1, z.shape[-3], h, w)
        return (latent,)


class LanPaint_ImageDecode:
    """nodes.py:1292-1342: VAE decode, resize to the original's exact size, merge with the original inside the
    mask with a MaskBlend-style boundary (one lp_mask_blend launch)."""

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "samples": ("LATENT", {"tooltip": "The inpainted latent to decode."}),
                "vae": ("VAE", {"tooltip": "The VAE."}),
            },
            "optional": {
                "image": ("IMAGE", {"tooltip": "The original image. When given, the decoded output is resized to its exact dimensions."}),
                "mask": ("MASK", {"tooltip": "The inpainting mask (1 = take the inpainted pixels, 0 = keep the original)."}),
                "blend_overlap": ("INT", {"default": 9, "min": 1, "max": 51, "step": 2,
                                          "tooltip": "Boundary blend width in pixels between the inpainted and original image (MaskBlend-style)."}),
            },
        }

    RETURN_TYPES = ("IMAGE",)
    RETURN_NAMES = ("image",)
    FUNCTION = "decode"
    CATEGORY = "image"
    DESCRIPTION = ("Decode an inpainted latent, resize to the original image's exact dimensions, and merge with the "
                   "original inside the mask (replaces VAEDecode + MaskBlend).")

    def decode(self, samples, vae, image=None, mask=None, blend_overlap=9):
        img = vae.decode(samples["samples"])
        if len(img.shape) == 5:    # [1, F, H, W, C]: combine batches (video-style VAE)
            img = img.reshape(-1, img.shape[-3], img.shape[-2], img.shape[-1])
        if image is None:
            return (img,)
        target_h, target_w = image.shape[1], image.shape[2]
        if tuple(img.shape[1:3]) != (target_h, target_w):
            img = torch.nn.functional.interpolate(img.movedim(-1, 1), size=(target_h, target_w), mode="bilinear",
                                                  align_corners=False).movedim(1, -1)
        if mask is None:
            return (img,)
        dev = _hip_device(image)
        # (the mask is handed over where it lives: its device decides which of torch's index rules the reference's resample followed)
        merged = merge_video_with_mask(image.to(dev), img.to(dev), mask, blend_overlap)
        return (merged.to(image.device),)
