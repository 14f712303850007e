"""Random-init SD1.5-shaped stand-in backbone (BASELINE.json configs[0], SURVEY.md 8d stand-in (ii)):
conv-in -> ResBlock -> downsample -> ResBlock + self-attention (MFMA via SDPA) -> upsample -> ResBlock
-> conv-out, sigma-conditioned, bf16 compute.  NOT a deliverable: it only gives the Langevin engine
a real nn.Module with a dual-head (x0, x0_BIG) output to sit in front of."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Res(nn.Module):
    def __init__(self, ch, temb):
        super().__init__()
        self.n1, self.c1 = nn.GroupNorm(8, ch), nn.Conv2d(ch, ch, 3, padding=1)
        self.n2, self.c2 = nn.GroupNorm(8, ch), nn.Conv2d(ch, ch, 3, padding=1)
        self.t = nn.Linear(temb, ch)

    def forward(self, x, t):
        h = self.c1(F.silu(self.n1(x))) + self.t(t)[:, :, None, None]
        return x + self.c2(F.silu(self.n2(h)))


class _Attn(nn.Module):
    def __init__(self, ch, heads=4):
        super().__init__()
        self.n, self.qkv, self.o, self.h = nn.GroupNorm(8, ch), nn.Linear(ch, 3 * ch), nn.Linear(ch, ch), heads

    def forward(self, x):
        b, c, hh, ww = x.shape
        t = self.n(x).flatten(2).transpose(1, 2)
        q, k, v = self.qkv(t).view(b, hh * ww, 3, self.h, c // self.h).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, hh * ww, c)
        return x + self.o(a).transpose(1, 2).reshape(b, c, hh, ww)


class DummyUNet(nn.Module):
    def __init__(self, in_ch=4, ch=128, temb=256, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.temb = temb
        self.t1, self.t2 = nn.Linear(temb, temb), nn.Linear(temb, temb)
        self.cin, self.r1 = nn.Conv2d(in_ch, ch, 3, padding=1), _Res(ch, temb)
        self.down, self.r2, self.attn = nn.Conv2d(ch, ch, 3, stride=2, padding=1), _Res(ch, temb), _Attn(ch)
        self.r3, self.cout = _Res(ch, temb), nn.Conv2d(ch, 2 * in_ch, 3, padding=1)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / math.sqrt(max(p[0].numel(), 1)) if p.ndim > 1 else 0.01))

    def _time(self, sigma):
        half = self.temb // 2
        f = torch.exp(-math.log(10000.0) * torch.arange(half, device=sigma.device, dtype=torch.float32) / half)
        a = torch.log(sigma.float().clamp_min(1e-4))[:, None] * f[None]
        return self.t2(F.silu(self.t1(torch.cat([a.sin(), a.cos()], dim=1).to(self.t1.weight.dtype))))

    def forward(self, x, sigma):
        dt = self.cin.weight.dtype
        t = self._time(sigma.reshape(-1).expand(x.shape[0]) if sigma.numel() == 1 else sigma.reshape(-1))
        c_in = (1.0 / (sigma.float().reshape(-1, 1, 1, 1) ** 2 + 1.0).sqrt())
        h = self.r1(self.cin((x * c_in).to(dt)), t)
        m = self.attn(self.r2(self.down(h), t))
        h = self.r3(h + F.interpolate(m, scale_factor=2.0, mode="nearest"), t)
        o = torch.tanh(self.cout(F.silu(h)).float())
        a, b = o.chunk(2, dim=1)
        x32 = x.float()
        return 0.85 * x32 * c_in + 0.1 * a, 0.75 * x32 * c_in + 0.1 * b        # (x0, x0_BIG)


class DummyUNetBackbone:
    """The `model(x, t, model_options=, seed=)` wrapper the engine calls (reference stub shape)."""

    def __init__(self, device, flow=False, dtype=torch.bfloat16, seed=0):
        self.inner_model = self
        self.net = DummyUNet(seed=seed).to(device=device, dtype=dtype).eval()
        self.model_sampling = type("S", (), {"lanpaint_noise_scaling_kind": "flow" if flow else "ve", "noise_scale": 1.0,
                                             "noise_scaling": staticmethod(
                                                 (lambda s, n, l, max_denoise=False: s * n + (1.0 - s) * l) if flow else
                                                 (lambda s, n, l, max_denoise=False: l + n * s))})()
        self.calls = 0

    @torch.no_grad()
    def __call__(self, x, t, model_options=None, seed=None):
        self.calls += 1
        return self.net(x, t)
