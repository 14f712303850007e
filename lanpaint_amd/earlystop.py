"""Inner-loop early stop of the think loop, host side.

Three pieces, none of which does per-element arithmetic on the host:

  StopOptions   the options contract of the reference (/root/reference/src/LanPaint/earlystop.py:74-95, 121-131): the
                `lanpaint_semantic_stop` dict {threshold, patience, distance_fn, min_steps}, the `lanpaint_semantic_trace`
                list and its three bench tags -- parsed once per sigma call, shared with the device-side stopper.
  stop_rule     the decision of ONE iteration as a pure function (six weighted sums, state) -> (state', record).  It is the
                host twin of the device's `es_decide` (csrc/step_kernel.hip: same sums layout, same order of the
                threshold / drift-anchor / patience tests) and is unit-tested against the oracle's stopper and the trace
                records of the reference-generated fixtures (tests/test_earlystop_rule.py).
  HostStopper   drives the rule when the verdict has to be formed on the host: a user `distance_fn` (host Python by
                contract) or one batch sharded over ranks (the sums are all-reduced first).  The sums come from two
                launches of lp_wmse_pair and ONE device->host read per iteration (the reference: 2-6 `.item()` syncs).

The default metric on an unsharded batch never comes here: it is evaluated on the device inside the step launches.
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, NamedTuple, Optional

import torch

from . import _cabi

# Keys of one trace record, in the reference's order (earlystop.py:315-334); the dict a user finds in the trace list.
TRACE_KEYS = ("case_id", "outer_step", "bench_timestep", "inner_step", "dist", "dist_inpaint", "dist_ring", "dist_drift",
              "threshold", "threshold_eff", "patience_counter", "patience_eff", "abt", "custom_dist", "stopped")


def abt_scale(abt_val: float) -> float:
    """4 a (1 - a) on a clamped to [0, 1], itself clamped: 0 at the ends of the schedule, 1 at abt = 0.5."""
    a = min(1.0, max(0.0, abt_val))
    return min(1.0, max(0.0, 4.0 * a * (1.0 - a)))


class StopOptions(NamedTuple):
    threshold: float
    patience_eff: int               # consecutive quiet iterations required: max(1, patience) + 1
    distance_fn: Optional[Callable[..., Any]]
    trace: Optional[list]
    tags: tuple                     # (bench_case_id, bench_outer_step, bench_timestep), only looked up with a trace list

    @classmethod
    def parse(cls, model_options, threshold, patience, distance_fn) -> Optional["StopOptions"]:
        """None = the inner early stop is off for this call (threshold or patience not positive)."""
        mo = model_options if isinstance(model_options, dict) else {}
        user = mo.get("lanpaint_semantic_stop")
        threshold, patience = float(threshold), int(patience)
        if isinstance(user, dict):
            threshold = float(user.get("threshold", threshold))
            patience = int(user.get("patience", patience))
            distance_fn = user.get("distance_fn", distance_fn)
            floor = _int_or_zero(user.get("min_steps")) - 1          # the legacy knob is a floor on patience, nothing more
            if patience > 0 and floor > 0:
                patience = max(patience, floor)
        if threshold <= 0.0 or patience <= 0:
            return None
        trace = mo.get("lanpaint_semantic_trace")
        trace = trace if isinstance(trace, list) else None
        tags = tuple(mo.get(k) for k in ("bench_case_id", "bench_outer_step", "bench_timestep")) if trace is not None else (None,) * 3
        return cls(threshold, max(1, patience) + 1, distance_fn, trace, tags)


def _int_or_zero(v) -> int:
    try:
        return int(v)
    except (TypeError, ValueError):
        return 0


class StopState(NamedTuple):
    counter: int = 0                # consecutive iterations at or under the threshold
    anchored: bool = False          # an x0 anchor is held (set by the first quiet iteration of a streak)


class StopRecord(NamedTuple):
    dist: float
    dist_inpaint: Optional[float]
    dist_ring: Optional[float]
    dist_drift: Optional[float]
    counter: int
    stopped: bool
    take_anchor: bool               # the caller keeps this iteration's x0 as the new anchor
    drop_anchor: bool               # the caller forgets the anchor


def stop_rule(sums, state: StopState, threshold: float, patience_eff: int, *, have_prev: bool, has_ring: bool,
              have_anchor: bool, have_x0: bool = True):
    """One iteration of the default-metric rule.  `sums` = (S w1 dA^2, S w1, S w2 dA^2, S w2, S w1 dB^2, S w2 dB^2) with
    w1 = inpaint weight, w2 = ring weight, dA = this iteration's x0 minus the previous one (or x_t after minus before on
    iteration 0: `have_prev` False, no ring term then), dB = x0 minus the anchor (`have_anchor`): the layout of the device's
    accumulator set (kEsSums, step_kernel.hip).  Distances are sum / (weight + 1e-12).  `have_x0`: this iteration produced an
    x0 the drift guard can hold on to (always, unless an overridden langevin_dynamics returns a state without one)."""
    s_a1, w1, s_a2, w2, s_b1, s_b2 = (float(v) for v in sums)
    d_in = s_a1 / (w1 + 1e-12)
    d_ring = s_a2 / (w2 + 1e-12) if (have_prev and has_ring) else None
    dist = d_in if d_ring is None else max(d_in, d_ring)
    d_drift, take, drop = None, False, False
    anchored = state.anchored
    if have_x0:                                     # the drift guard holds on to an x0 of a quiet iteration
        if dist > threshold:
            drop, anchored = anchored, False
        elif not anchored:
            take, anchored = True, True
        elif have_anchor:
            d_drift = s_b1 / (w1 + 1e-12)
            if has_ring:
                d_drift = max(d_drift, s_b2 / (w2 + 1e-12))
            dist = max(dist, d_drift)
    quiet = dist <= threshold
    counter = state.counter + 1 if quiet else 0
    if not quiet and anchored:                      # (drift pushed a quiet step over the threshold)
        drop, anchored, take = True, False, False
    return StopState(counter, anchored), StopRecord(dist, d_in, d_ring, d_drift, counter, counter >= patience_eff, take, drop)


def bind_distance_fn(fn):
    """A user metric as f(prev, cur, ctx) -> scalar | None.  Accepted shapes: three positionals (or *args) -> called
    (prev, cur, ctx); a `ctx` keyword (or **kwargs) -> (prev, cur, ctx=ctx); the legacy pair -> called (cur, prev).  A callable
    whose signature cannot be read is tried with three arguments first and, if the CALL itself (not the body) rejects them,
    with the legacy pair."""
    if not callable(fn):
        return None
    try:
        params = inspect.signature(fn).parameters.values()
    except (TypeError, ValueError):
        def probe(prev, cur, ctx):
            try:
                return fn(prev, cur, ctx)
            except TypeError as err:
                raised_inside = err.__traceback__ is not None and err.__traceback__.tb_next is not None
                if raised_inside:
                    raise
                return fn(cur, prev)
        return probe
    kinds = [p.kind for p in params]
    n_pos = sum(k in (inspect.Parameter.POSITIONAL_ONLY, inspect.Parameter.POSITIONAL_OR_KEYWORD) for k in kinds)
    if n_pos >= 3 or inspect.Parameter.VAR_POSITIONAL in kinds:
        return lambda prev, cur, ctx: fn(prev, cur, ctx)
    if inspect.Parameter.VAR_KEYWORD in kinds or any(p.name == "ctx" for p in params):
        return lambda prev, cur, ctx: fn(prev, cur, ctx=ctx)
    return lambda prev, cur, ctx: fn(cur, prev)


def scalar_distance(value) -> Optional[float]:
    """The return contract of a distance_fn: None (fall back to the default metric) or one number."""
    if value is None:
        return None
    if isinstance(value, torch.Tensor):
        if value.numel() != 1:
            raise TypeError("distance_fn must return None or a scalar / 0-d (1-element) tensor")
        return float(value.item())
    return float(value)


class WeightedSums:
    """Device buffers + launches for the six sums of `stop_rule` (lp_boundary_ring once, lp_wmse_pair per pair)."""
    SCRATCH_BLOCKS = 1024

    def __init__(self, latent_mask: torch.Tensor):
        self.lib = _cabi.load()
        m = latent_mask if (latent_mask.dtype == torch.float32 and latent_mask.is_contiguous()) else latent_mask.float().contiguous()
        self.mask, self.n_el, self.ring = m, m.numel(), None
        if m.dim() == 4:                                   # the ring weight exists for image latents only
            self.ring = torch.empty_like(m)
            b, c, h, w = m.shape
            _cabi.check(self.lib.lp_boundary_ring(m.data_ptr(), self.ring.data_ptr(), b * c, h, w, self._stream()), "lp_boundary_ring")
        self.reduce_group = False       # True / a ProcessGroup: the batch is sharded over those ranks, the sums are all-reduced
        self.acc = torch.zeros((2, 4), dtype=torch.float64, device=m.device)
        self.scratch = torch.empty((self.SCRATCH_BLOCKS * 4,), dtype=torch.float64, device=m.device)

    def _stream(self):
        return torch.cuda.current_stream(self.mask.device).cuda_stream

    def _launch(self, a, b, slot):
        a = a if (a.dtype == torch.float32 and a.is_contiguous()) else a.float().contiguous()
        b = b if (b.dtype == torch.float32 and b.is_contiguous()) else b.float().contiguous()
        _cabi.check(self.lib.lp_wmse_pair(a.data_ptr(), b.data_ptr(), self.mask.data_ptr(),
                                          self.ring.data_ptr() if self.ring is not None else None, self.n_el,
                                          self.acc[slot].data_ptr(), self.scratch.data_ptr(), self.SCRATCH_BLOCKS, self._stream()),
                    "lp_wmse_pair")

    def inpaint_weight(self) -> float:
        z = torch.zeros_like(self.mask)
        self._launch(z, z, 0)
        return float(self.acc[0, 1].item())

    def six(self, pair_a, pair_b=None):
        """The sums of pair A (and B) in `stop_rule`'s layout: the launches, the optional all-reduce, ONE host read."""
        self._launch(pair_a[0], pair_a[1], 0)
        if pair_b is not None:
            self._launch(pair_b[0], pair_b[1], 1)
        if self.reduce_group is not False:
            from .distributed import all_reduce_stop_sums
            all_reduce_stop_sums(self.acc, None if self.reduce_group is True else self.reduce_group)
        (a1, w1, a2, w2), (b1, _w1, b2, _w2) = self.acc.tolist()
        return (a1, w1, a2, w2, b1 if pair_b is not None else 0.0, b2 if pair_b is not None else 0.0)


class HostStopper:
    """The stop decision formed on the host, one `observe` per think iteration."""

    @classmethod
    def from_options(cls, opts: Optional[StopOptions], latent_mask: torch.Tensor, abt: torch.Tensor) -> Optional["HostStopper"]:
        """None when the stopper cannot fire in this call: off, a threshold scaled to zero at this abt, nothing to inpaint."""
        if opts is None:
            return None
        try:
            abt_val = float(torch.mean(abt).item())
        except (TypeError, ValueError):
            abt_val = 0.0
        threshold_eff = opts.threshold * abt_scale(abt_val)
        if threshold_eff <= 0.0:
            return None
        sums = WeightedSums(latent_mask)
        if sums.inpaint_weight() < 1e-6:
            return None
        return cls(opts, threshold_eff, abt_val, sums)

    def __init__(self, opts: StopOptions, threshold_eff: float, abt_val: float, sums: WeightedSums):
        self.opts, self.threshold_eff, self.abt_val, self.sums = opts, float(threshold_eff), abt_val, sums
        self.state = StopState()
        self.anchor = None              # the x0 the drift guard compares with
        self.user_metric = bind_distance_fn(opts.distance_fn)

    @property
    def has_custom_distance_fn(self) -> bool:
        return self.user_metric is not None

    def observe(self, i, *, x_before, x_after, x_prev_for_user, x0_prev, x0_cur, ctx) -> bool:
        """Iteration i has run: True when the loop should stop.  Appends the trace record when a trace list was given."""
        user = scalar_distance(self.user_metric(x_prev_for_user, x_after, ctx)) if self.user_metric is not None else None
        if user is not None:            # a user distance is compared with the UNSCALED threshold and bypasses the drift guard
            threshold = self.opts.threshold
            quiet = user <= threshold
            self.state = StopState(self.state.counter + 1 if quiet else 0, self.state.anchored and quiet)
            if not quiet:
                self.anchor = None
            rec = StopRecord(user, None, None, None, self.state.counter, self.state.counter >= self.opts.patience_eff, False, False)
        else:
            threshold = self.threshold_eff
            have_prev = x0_prev is not None and x0_cur is not None
            pair_a = (x0_cur, x0_prev) if have_prev else (x_after, x_before)
            pair_b = (x0_cur, self.anchor) if (x0_cur is not None and self.anchor is not None) else None
            six = self.sums.six(pair_a, pair_b)
            self.state, rec = stop_rule(six, self.state, threshold, self.opts.patience_eff, have_prev=have_prev,
                                        has_ring=self.sums.ring is not None, have_anchor=pair_b is not None,
                                        have_x0=x0_cur is not None)
            if rec.take_anchor:
                self.anchor = x0_cur.detach().clone()
            elif rec.drop_anchor:
                self.anchor = None
        if self.opts.trace is not None:
            values = (*self.opts.tags, i + 1, rec.dist, rec.dist_inpaint, rec.dist_ring, rec.dist_drift, float(threshold),
                      self.threshold_eff, rec.counter, self.opts.patience_eff, self.abt_val, user is not None, rec.stopped)
            self.opts.trace.append(dict(zip(TRACE_KEYS, values)))
        return rec.stopped
