#!/usr/bin/env python3
"""lp_mask_blend vs the eager op chain the reference runs (max_pool2d + conv2d + 4 elementwise) on the
GPU, 1080p frames.  Algorithmic bytes: mask 4 B + (2 reads + 1 write) * C * 4 B per pixel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanpaint_amd import blend            # noqa: E402


def eager(i1, i2, m, k):
    m = torch.nn.functional.max_pool2d(m, kernel_size=k, stride=1, padding=k // 2)
    ker = blend.gaussian_kernel_2d(k).to(i1.device)[None, None]
    m = torch.nn.functional.conv2d(m[:, None], ker, padding=k // 2)[:, 0]
    return i1 * (1 - m[..., None]) + i2 * m[..., None]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = "cuda"
    for (b, h, w, c), k in [((1, 1080, 1920, 3), 9), ((8, 1080, 1920, 3), 9), ((1, 1080, 1920, 3), 51), ((81, 480, 832, 3), 15)]:
        g = torch.Generator(device=dev).manual_seed(0)
        i1 = torch.rand((b, h, w, c), device=dev, generator=g)
        i2 = torch.rand((b, h, w, c), device=dev, generator=g)
        m = (torch.rand((b, h, w), device=dev, generator=g) > 0.9).float()
        t_hip = timeit(lambda: blend.mask_blend(i1, i2, m, k))
        t_ref = timeit(lambda: eager(i1, i2, m, k))
        err = float((blend.mask_blend(i1, i2, m, k) - eager(i1, i2, m, k)).abs().max())
        byts = b * h * w * (4 + 3 * c * 4)
        print(f"{b}x{h}x{w}x{c} k={k}: lp_mask_blend {t_hip:9.1f} us ({byts / t_hip / 1e3:7.0f} GB/s algorithmic)  "
              f"eager torch chain {t_ref:9.1f} us  speedup {t_ref / t_hip:5.1f}x  max|diff| {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
