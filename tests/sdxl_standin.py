"""Random-init SDXL-SHAPED stand-in backbone for BASELINE.json configs[1] (SDXL 1x4x128x128 latent, bf16) and the "MFMA
attention path" of configs[3].  NOT a deliverable -- the backbone belongs to the caller -- it is the
smallest nn.Module that exercises what the Langevin path was built to sit behind:

  * bf16 end to end, a latent at the SDXL shape (4 x 128 x 128), three resolution levels 128 -> 64 -> 32 of ResBlocks
    (GroupNorm / SiLU / 3x3 conv: MIOpen implicit-GEMM convolutions on the matrix cores),
  * self-attention over the 32 x 32 = 1024 tokens of the lowest level (SDPA) plus a cross-attention to SDXL-shaped text states
    [77, 2048] (hipBLASLt GEMMs), pooled / ADM vector [2816] into the time embedding,
  * ONE batched cond + uncond pass per call: the latent is doubled along the batch axis, the two halves differ only in their
    conditioning, and the call returns `FusedCFGHeads(cond, uncond, scale, scale_BIG)` -- what
    `lanpaint_amd.nodes.sampling_function_LanPaint` returns for ComfyUI's stock cfg_function -- so the step kernel forms both
    CFG heads itself (LP_FL_CFG_FUSED),
  * ComfyUI's data flow around the network (comfy/model_base.py BaseModel.apply_model): the latent is cast to the network's
    dtype on the way in, the network's output comes back as fp32 and the denoised prediction x0 = c_skip x + c_out eps is
    formed in fp32 from the fp32 latent -- the predictions the Langevin path receives are fp32 tensors.  (`half_out=True`
    instead rounds the predictions to the network's dtype -- LP_FL_X0_BF16 next to an engine with model_dtype = bf16,
    LP_FL_XIN_BF16: every stream half-width.  At sigma ~ 14 a latent value is ~ 60, where bf16 resolves 0.25: such a run is
    precision-limited by its own storage format, whoever computes the Langevin step.)

The same object serves the oracle side of the parity check through `as_oracle_model()`: numpy in, the identical module on the
device, the reference's eager `uncond + (cond - uncond) * scale` twice (nodes.py:161-175 with ComfyUI's cfg_function), numpy out.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Res(nn.Module):
    def __init__(self, cin, cout, temb):
        super().__init__()
        self.n1, self.c1 = nn.GroupNorm(32, cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.n2, self.c2 = nn.GroupNorm(32, cout), nn.Conv2d(cout, cout, 3, padding=1)
        self.t = nn.Linear(temb, cout)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()

    def forward(self, x, t):
        h = self.c1(F.silu(self.n1(x))) + self.t(t)[:, :, None, None]
        return self.skip(x) + self.c2(F.silu(self.n2(h)))


class _Transformer(nn.Module):
    """Self-attention over the level's tokens, cross-attention to the text states, GEGLU-free MLP: one SDXL-style block."""

    def __init__(self, ch, ctx, heads):
        super().__init__()
        self.h = heads
        self.n = nn.GroupNorm(32, ch)
        self.ln1, self.ln2, self.ln3 = nn.LayerNorm(ch), nn.LayerNorm(ch), nn.LayerNorm(ch)
        self.qkv, self.o1 = nn.Linear(ch, 3 * ch), nn.Linear(ch, ch)
        self.q2, self.kv2, self.o2 = nn.Linear(ch, ch), nn.Linear(ctx, 2 * ch), nn.Linear(ch, ch)
        self.f1, self.f2 = nn.Linear(ch, 4 * ch), nn.Linear(4 * ch, ch)

    def forward(self, x, ctx):
        b, c, hh, ww = x.shape
        t = self.n(x).flatten(2).transpose(1, 2)                                     # [B, T, C]
        d = c // self.h
        q, k, v = self.qkv(self.ln1(t)).view(b, -1, 3, self.h, d).permute(2, 0, 3, 1, 4)
        t = t + self.o1(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, -1, c))
        q = self.q2(self.ln2(t)).view(b, -1, self.h, d).transpose(1, 2)
        k, v = self.kv2(ctx).view(b, -1, 2, self.h, d).permute(2, 0, 3, 1, 4)
        t = t + self.o2(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, -1, c))
        t = t + self.f2(F.gelu(self.f1(self.ln3(t))))
        return x + t.transpose(1, 2).reshape(b, c, hh, ww)


class SDXLShapedNet(nn.Module):
    def __init__(self, in_ch=4, ch=(128, 256, 512), temb=512, ctx=2048, adm=2816, heads=8, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.temb = temb
        self.t1, self.t2, self.adm = nn.Linear(temb, temb), nn.Linear(temb, temb), nn.Linear(adm, temb)
        c0, c1, c2 = ch
        self.cin = nn.Conv2d(in_ch, c0, 3, padding=1)
        self.r0, self.d0 = _Res(c0, c0, temb), nn.Conv2d(c0, c0, 3, stride=2, padding=1)
        self.r1, self.d1 = _Res(c0, c1, temb), nn.Conv2d(c1, c1, 3, stride=2, padding=1)
        self.r2, self.tr, self.r3 = _Res(c1, c2, temb), _Transformer(c2, ctx, heads), _Res(c2, c2, temb)
        self.u1, self.r4 = nn.Conv2d(c2, c1, 3, padding=1), _Res(2 * c1, c1, temb)
        self.u0, self.r5 = nn.Conv2d(c1, c0, 3, padding=1), _Res(2 * c0, c0, temb)
        self.nout, self.cout = nn.GroupNorm(32, c0), nn.Conv2d(c0, in_ch, 3, padding=1)
        with torch.no_grad():
            for p in self.parameters():
                if p.ndim > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(p[0].numel())))
                else:
                    p.zero_()
            for m in self.modules():
                if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                    m.weight.fill_(1.0)

    def _time(self, sigma):
        half = self.temb // 2
        f = torch.exp(-math.log(10000.0) * torch.arange(half, device=sigma.device, dtype=torch.float32) / half)
        a = (0.25 * torch.log(sigma.float().clamp_min(1e-4)))[:, None] * f[None]
        return torch.cat([a.sin(), a.cos()], dim=1)

    def forward(self, x, sigma, ctx, adm):
        """x [N, 4, H, W] (any float dtype), sigma [N], ctx [N, 77, 2048], adm [N, 2816] -> the network output [N, 4, H, W]
        in the module's dtype (an eps-like correction; the wrapper turns it into the denoised prediction)."""
        dt = self.cin.weight.dtype
        t = self.t2(F.silu(self.t1(self._time(sigma).to(dt)))) + self.adm(adm.to(dt))
        h0 = self.r0(self.cin(x.to(dt)), t)
        h1 = self.r1(self.d0(h0), t)
        h2 = self.r3(self.tr(self.r2(self.d1(h1), t), ctx.to(dt)), t)
        u1 = self.r4(torch.cat([self.u1(F.interpolate(h2, scale_factor=2.0, mode="nearest")), h1], dim=1), t)
        u0 = self.r5(torch.cat([self.u0(F.interpolate(u1, scale_factor=2.0, mode="nearest")), h0], dim=1), t)
        return self.cout(F.silu(self.nout(u0)))


class _Sampling:
    def __init__(self, flow):
        self.lanpaint_noise_scaling_kind = "flow" if flow else "ve"
        self.noise_scale = 1.0
        self.flow = flow

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return (sigma * noise + (1.0 - sigma) * latent_image) if self.flow else (latent_image + noise * sigma)


class SDXLShapedBackbone:
    """`model(x, t, model_options=, seed=)` as the engine calls it (the shape of nodes.py:161-175 behind a CFGGuider): one
    batched cond + uncond pass, the x0 predictions of both halves, CFG left to the consumer (FusedCFGHeads)."""

    def __init__(self, device, flow=False, dtype=torch.bfloat16, seed=0, cfg_scale=5.0, cfg_scale_big=8.0, fused=True,
                 channels=(128, 256, 512), half_out=False):
        from lanpaint_amd.types import FusedCFGHeads
        self._heads_type = FusedCFGHeads
        self.inner_model = self
        self.model_sampling = _Sampling(flow)
        self.device, self.dtype, self.flow, self.fused, self.half_out = device, dtype, flow, fused, half_out
        self.net = SDXLShapedNet(ch=channels, seed=seed).to(device=device, dtype=dtype).eval()
        g = torch.Generator().manual_seed(1000 + seed)
        self.ctx = torch.randn((1, 77, 2048), generator=g).to(device=device, dtype=dtype)          # SDXL text states
        self.adm = torch.randn((1, 2816), generator=g).to(device=device, dtype=dtype)              # pooled text + size conditioning
        self.scale, self.scale_big = float(cfg_scale), float(cfg_scale_big)
        self.calls = 0
        self.n_params = sum(p.numel() for p in self.net.parameters())

    @torch.no_grad()
    def predict(self, x, t):
        """(cond, uncond) x0 predictions from ONE pass over the doubled batch (cond half: the text states; uncond half: zeros --
        ComfyUI's calc_cond_batch concatenates the two the same way); fp32 unless `half_out`."""
        b = x.shape[0]
        sig = t.reshape(-1).float()
        sig = sig.expand(b) if sig.numel() == 1 else sig
        x32 = x.float()
        if self.flow:
            c_in, c_out = torch.ones_like(sig), -sig                                               # x0 = x - t * v
        else:
            c_in, c_out = 1.0 / (sig ** 2 + 1.0).sqrt(), -sig                                      # EPS: x0 = x - sigma * eps
        v = lambda a: a.reshape(-1, 1, 1, 1)                                                        # noqa: E731
        net_in = (x32 * v(c_in)).to(self.dtype)
        ctx = torch.cat([self.ctx.expand(b, -1, -1), torch.zeros_like(self.ctx).expand(b, -1, -1)], dim=0)
        adm = torch.cat([self.adm.expand(b, -1), torch.zeros_like(self.adm).expand(b, -1)], dim=0)
        eps = self.net(torch.cat([net_in, net_in], dim=0), torch.cat([sig, sig]), ctx, adm).float()
        x0 = torch.cat([x32, x32], dim=0) + eps * v(torch.cat([c_out, c_out])) * 0.25
        if self.half_out:
            x0 = x0.to(self.dtype)
        return x0[:b], x0[b:]

    def __call__(self, x, t, model_options=None, seed=None):
        self.calls += 1
        cond, uncond = self.predict(x, t)
        if self.fused:
            return self._heads_type(cond, uncond, self.scale, self.scale_big)
        diff = cond - uncond                                                                          # the reference's eager form
        return uncond + diff * self.scale, uncond + diff * self.scale_big

    def as_oracle_model(self, input_dtype=None):
        """The oracle's view of the same backbone: numpy fp32 in (rounded to `input_dtype` first when the engine emits its
        model-space latent in that dtype: model_dtype); the identical module on the device; the reference's eager CFG
        combination `uncond + (cond - uncond) * scale`, twice (nodes.py:161-175 with ComfyUI's cfg_function); numpy fp32 out."""
        outer = self

        class _OracleModel:
            def __init__(self):
                self.inner_model = self
                self.model_sampling = outer.model_sampling
                self.calls = 0

            def __call__(self, x, t, model_options=None, seed=None):
                self.calls += 1
                xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(outer.device)
                if input_dtype is not None:
                    xt = xt.to(input_dtype)
                tt = torch.from_numpy(np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(-1))).to(outer.device)
                cond, uncond = outer.predict(xt, tt)
                diff = cond.float() - uncond.float()
                h0 = uncond.float() + diff * outer.scale
                h1 = uncond.float() + diff * outer.scale_big
                return h0.cpu().numpy(), h1.cpu().numpy()

        return _OracleModel()
