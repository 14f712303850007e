"""Inner-loop early stop for the Langevin think loop, metric on the GPU.

Behavioural mirror of the reference's `LanPaintEarlyStopper`
(/root/reference/src/LanPaint/earlystop.py:58-336): same options contract
(`model_options["lanpaint_semantic_stop"]`, `lanpaint_semantic_trace`), same
decision rule (max of inpaint-region and mask-boundary-ring weighted MSE of
successive x0, abt-scaled threshold, patience+1 consecutive hits, drift anchor),
same trace dict.  What differs is where the arithmetic runs: the ring stencil and
the weighted-MSE reductions are HIP kernels (lp_boundary_ring / lp_wmse_pair) and
one iteration costs ONE device->host read of 8 doubles instead of the
reference's 2-6 `.item()` syncs (earlystop.py:55).
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Optional

import torch

from . import _cabi
from .types import LangevinState


def _clamp01(v: float) -> float:
    return 0.0 if v <= 0.0 else (1.0 if v >= 1.0 else v)


def _abt_scale(abt_val: float) -> float:
    """earlystop.py:21-29: 0 at abt in {0,1}, 1 at abt = 0.5."""
    a = _clamp01(abt_val)
    return _clamp01(4.0 * a * (1.0 - a))


class _Metric:
    """Device buffers + launches for { wMSE(a,b; 1-mask), wMSE(a,b; ring) }."""
    SCRATCH_BLOCKS = 1024

    def __init__(self, latent_mask: torch.Tensor):
        self.lib = _cabi.load()
        m = latent_mask
        if m.dtype != torch.float32 or not m.is_contiguous():
            m = m.float().contiguous()
        self.mask = m
        self.n_el = m.numel()
        dev = m.device
        self.ring = None
        if m.dim() == 4:                                   # earlystop.py:38-39: ring only for 4-D latents
            self.ring = torch.empty_like(m)
            b, c, h, w = m.shape
            _cabi.check(self.lib.lp_boundary_ring(m.data_ptr(), self.ring.data_ptr(), b * c, h, w, self._stream()),
                        "lp_boundary_ring")
        # False: single-process metric (default).  True / a ProcessGroup: the batch tensor is sharded
        # over those ranks and the metric is taken over the whole batch, as the reference defines it.
        self.reduce_group = False
        self.acc = torch.zeros((2, 4), dtype=torch.float64, device=dev)
        self.scratch = torch.empty((self.SCRATCH_BLOCKS * 4,), dtype=torch.float64, device=dev)

    def _stream(self):
        return torch.cuda.current_stream(self.mask.device).cuda_stream

    @staticmethod
    def _f32(t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        return t

    def inpaint_weight_sum(self) -> float:
        z = torch.zeros_like(self.mask)
        self._launch(z, z, 0)
        return float(self.acc[0, 1].item())

    def _launch(self, a, b, slot):
        a, b = self._f32(a), self._f32(b)
        ring_ptr = self.ring.data_ptr() if self.ring is not None else None
        _cabi.check(self.lib.lp_wmse_pair(a.data_ptr(), b.data_ptr(), self.mask.data_ptr(), ring_ptr, self.n_el,
                                          self.acc[slot].data_ptr(), self.scratch.data_ptr(), self.SCRATCH_BLOCKS,
                                          self._stream()), "lp_wmse_pair")

    def distances(self, pairs):
        """pairs: list of up to 2 (a, b).  Returns [(d_inpaint, d_ring|None), ...] with ONE sync."""
        for slot, (a, b) in enumerate(pairs):
            self._launch(a, b, slot)
        if self.reduce_group is not False:                  # batch sharded over ranks: sum the partial sums
            from .distributed import all_reduce_stop_sums
            all_reduce_stop_sums(self.acc, None if self.reduce_group is True else self.reduce_group)
        vals = self.acc.tolist()                            # the single host sync of this iteration
        out = []
        for slot in range(len(pairs)):
            s1, w1, s2, w2 = vals[slot]
            d_in = s1 / (w1 + 1e-12)
            d_ring = (s2 / (w2 + 1e-12)) if self.ring is not None else None
            out.append((d_in, d_ring))
        return out


class LanPaintEarlyStopper:
    """Per-iteration convergence test (off unless threshold > 0 and patience > 0)."""

    @classmethod
    def from_options(cls, *, model_options: Optional[dict], latent_mask: torch.Tensor, abt: torch.Tensor,
                     default_threshold: float, default_patience: int,
                     default_distance_fn: Optional[Callable[..., Any]]) -> Optional["LanPaintEarlyStopper"]:
        """earlystop.py:63-156."""
        opts = model_options.get("lanpaint_semantic_stop") if isinstance(model_options, dict) else None
        threshold, patience, distance_fn = float(default_threshold), int(default_patience), default_distance_fn
        if isinstance(opts, dict):
            threshold = float(opts.get("threshold", threshold))
            patience = int(opts.get("patience", patience))
            distance_fn = opts.get("distance_fn", distance_fn)
            if patience > 0 and opts.get("min_steps") is not None:      # legacy knob -> patience floor
                try:
                    min_steps = int(opts.get("min_steps"))
                except (TypeError, ValueError):
                    min_steps = 0
                if min_steps > 1:
                    patience = max(patience, min_steps - 1)
        if not (threshold > 0.0 and patience > 0):
            return None
        try:
            abt_val = float(torch.mean(abt).item())
        except (TypeError, ValueError):
            abt_val = 0.0
        threshold_eff = threshold * _abt_scale(abt_val)
        if threshold_eff <= 0.0:
            return None
        metric = _Metric(latent_mask)
        if metric.inpaint_weight_sum() < 1e-6:
            return None
        trace = model_options.get("lanpaint_semantic_trace") if isinstance(model_options, dict) else None
        tags = (None, None, None)
        if isinstance(trace, list) and isinstance(model_options, dict):
            tags = (model_options.get("bench_case_id"), model_options.get("bench_outer_step"),
                    model_options.get("bench_timestep"))
        return cls(threshold=threshold, threshold_eff=threshold_eff, patience_eff=max(1, patience) + 1, metric=metric,
                   distance_fn=distance_fn, trace=trace, tags=tags, abt_val=abt_val)

    def __init__(self, *, threshold, threshold_eff, patience_eff, metric, distance_fn=None, trace=None,
                 tags=(None, None, None), abt_val=None):
        self.enabled = True
        self.threshold = float(threshold)
        self.threshold_eff = float(threshold_eff)
        self.patience_eff = int(patience_eff)
        self.metric = metric
        self.trace = trace
        self.bench_case_id, self.bench_outer_step, self.bench_timestep = tags
        self.abt_val = abt_val
        self.patience_counter = 0
        self.x0_anchor = None
        self._dist_wrapper = self._wrap_distance_fn(distance_fn)

    @property
    def has_custom_distance_fn(self) -> bool:
        return self._dist_wrapper is not None

    @property
    def ring_weight(self):
        return self.metric.ring

    @staticmethod
    def _wrap_distance_fn(distance_fn):
        """Normalise a user metric to fn(prev, cur, ctx) -> scalar | None (earlystop.py:187-236):
        3+ positional or *args -> (prev, cur, ctx); a `ctx` / **kwargs parameter ->
        (prev, cur, ctx=ctx); otherwise the legacy 2-arg form is called as (cur, prev)."""
        if not callable(distance_fn):
            return None
        try:
            params = list(inspect.signature(distance_fn).parameters.values())
        except (ValueError, TypeError):
            def fallback(p, c, ctx):
                try:
                    return distance_fn(p, c, ctx)
                except TypeError as e:
                    tb = e.__traceback__
                    if tb is not None and tb.tb_frame.f_code is not fallback.__code__:
                        raise
                    return distance_fn(c, p)
            return fallback
        positional = [p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        if len(positional) >= 3 or any(p.kind == p.VAR_POSITIONAL for p in params):
            return lambda p, c, ctx: distance_fn(p, c, ctx)
        if any(p.name == "ctx" for p in params) or any(p.kind == p.VAR_KEYWORD for p in params):
            return lambda p, c, ctx: distance_fn(p, c, ctx=ctx)
        return lambda p, c, ctx: distance_fn(c, p)

    @staticmethod
    def _state_x0(arg):
        if isinstance(arg, LangevinState):
            return arg.x0
        if isinstance(arg, tuple) and len(arg) >= 3:
            return arg[2]
        return None

    def step(self, *, i: int, n_steps: int, x_t_before, x_t_after, x_t_prev_for_custom, prev_args, args, ctx) -> bool:
        """earlystop.py:238-336.  Returns True when the think loop should break."""
        if not self.enabled:
            return False
        dist = None
        dist_inpaint = dist_ring = dist_drift = x0_cur = None
        if self._dist_wrapper is not None:
            dist = self._dist_wrapper(x_t_prev_for_custom, x_t_after, ctx)
            if dist is not None:
                if isinstance(dist, torch.Tensor):
                    if dist.numel() != 1:
                        raise TypeError("distance_fn must return None or a scalar / 0-d (1-element) tensor")
                    dist = float(dist.item())
                else:
                    dist = float(dist)
        custom = dist is not None
        threshold_used = self.threshold if custom else self.threshold_eff

        if not custom:
            x0_prev, x0_cur = self._state_x0(prev_args), self._state_x0(args)
            if x0_prev is not None and x0_cur is not None:
                pairs = [(x0_cur, x0_prev)]
                if self.x0_anchor is not None:        # drift is only consulted on a hit; fetch it in the same sync
                    pairs.append((x0_cur, self.x0_anchor))
                res = self.metric.distances(pairs)
                dist_inpaint, dist_ring = res[0]
                dist = dist_inpaint if dist_ring is None else max(dist_inpaint, dist_ring)
                drift = res[1] if len(res) > 1 else None
            else:
                (dist_inpaint, _), = self.metric.distances([(x_t_after, x_t_before)])
                dist = dist_inpaint
                drift = None
            if x0_cur is not None:                    # drift guard (default metric only)
                if dist <= threshold_used:
                    if self.x0_anchor is None:
                        self.x0_anchor = x0_cur.detach().clone()
                    else:
                        if drift is None:
                            drift = self.metric.distances([(x0_cur, self.x0_anchor)])[0]
                        dist_drift = drift[0] if drift[1] is None else max(drift[0], drift[1])
                        dist = max(dist, dist_drift)
                else:
                    self.x0_anchor = None

        if dist <= threshold_used:
            self.patience_counter += 1
        else:
            self.patience_counter = 0
            self.x0_anchor = None
        should_stop = self.patience_counter >= self.patience_eff

        if isinstance(self.trace, list):
            self.trace.append({
                "case_id": self.bench_case_id, "outer_step": self.bench_outer_step,
                "bench_timestep": self.bench_timestep, "inner_step": i + 1, "dist": dist,
                "dist_inpaint": None if dist_inpaint is None else float(dist_inpaint),
                "dist_ring": None if dist_ring is None else float(dist_ring),
                "dist_drift": None if dist_drift is None else float(dist_drift),
                "threshold": float(threshold_used), "threshold_eff": float(self.threshold_eff),
                "patience_counter": int(self.patience_counter), "patience_eff": int(self.patience_eff),
                "abt": None if self.abt_val is None else float(self.abt_val),
                "custom_dist": bool(custom), "stopped": bool(should_stop),
            })
        return bool(should_stop)
