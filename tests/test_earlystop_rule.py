"""The host-side stop rule (lanpaint_amd/earlystop.py::stop_rule) as a pure function: replayed over the per-iteration trace
records the unmodified reference wrote into the fixtures, walked next to the oracle's stopper on random distance sequences
(hypothesis), and the options / trace-key / distance_fn contracts.  No GPU: the rule takes sums, not tensors; its device twin
(es_decide, csrc/step_kernel.hip) is compared with the same oracle stopper by the GPU suite."""
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from lanpaint_amd import earlystop as es
from oracle import lanpaint_oracle as orc
from tests import golden_cases as gc
from tests.helpers import load_golden

TRACE_CASES = [n for n in sorted(gc.CASES) if (gc.CASES[n].get("model_options") or {}).get("lanpaint_semantic_stop")]


def _opt(v):
    return None if (isinstance(v, float) and math.isnan(v)) else float(v)


@pytest.mark.parametrize("name", TRACE_CASES)
def test_rule_replays_the_references_trace(name):
    """Feed the rule the distances the reference measured at every inner iteration (fixture: dist_inpaint, dist_ring, dist_drift
    of earlystop.py:315-334): it must walk the reference's patience counter, take / drop the drift anchor where the reference
    did (dist_drift is present exactly when an anchor was consulted) and stop on the same iteration."""
    g = load_golden(name)
    opts = es.StopOptions.parse(gc.CASES[name]["model_options"], 0.0, 1, None)
    thr_eff = float(g["trace_threshold_eff"][0])
    assert opts.patience_eff == int(g["trace_patience_eff"][0])
    assert thr_eff == pytest.approx(opts.threshold * es.abt_scale(float(g["trace_abt"][0])), rel=1e-6)
    state = es.StopState()
    for i in range(len(g["trace_dist"])):
        d_in, d_ring, d_drift = (_opt(float(g[f"trace_{k}"][i])) for k in ("dist_inpaint", "dist_ring", "dist_drift"))
        have_prev = i > 0
        assert (d_ring is not None) == have_prev                                   # 4-D latents: a ring from iteration 1 on
        sums = (d_in, 1.0, d_ring or 0.0, 1.0, d_drift or 0.0, d_drift or 0.0)
        anchored_before = state.anchored
        state, rec = es.stop_rule(sums, state, thr_eff, opts.patience_eff, have_prev=have_prev, has_ring=True,
                                  have_anchor=anchored_before)
        assert rec.counter == int(g["trace_counter"][i]) and rec.stopped == bool(g["trace_stopped"][i]), i
        assert rec.dist == pytest.approx(float(g["trace_dist"][i]), rel=1e-9)
        assert (rec.dist_drift is not None) == (d_drift is not None), i
        assert rec.dist_ring == (None if d_ring is None else pytest.approx(d_ring, rel=1e-9))
    assert rec.stopped == bool(g["trace_stopped"][-1])


class _ScriptedOracleStopper(orc.OracleEarlyStopper):
    """The oracle's stopper (pinned to the reference by the fixtures) with its distances scripted instead of measured."""

    def __init__(self, threshold_eff, patience, script, ring):
        self.xp = orc.NumpyBackend()
        self.enabled, self.patience_eff = True, max(1, patience) + 1
        self.threshold = self.threshold_eff = threshold_eff
        self.abt_val, self.counter, self.anchor, self.trace, self.trace_sink, self.tags = 0.5, 0, None, [], None, (None,) * 3
        self.inpaint, self.ring = np.ones(1, np.float32), (np.ones(1, np.float32) if ring else None)
        self.script = script

    def _pair(self, a, b):
        """iteration self.i: the step pair (x0 against the previous x0) or, when `b` is the held anchor, the drift pair"""
        d_in, d_ring = self.script["drift" if b is self.anchor else "step"][self.i]
        d_ring = d_ring if self.ring is not None else None
        return d_in, d_ring, (d_in if d_ring is None else max(d_in, d_ring))


dist = st.floats(min_value=0.0, max_value=2.0, allow_nan=False, width=32)


@settings(max_examples=300, deadline=None)
@given(first=dist, steps=st.lists(st.tuples(dist, dist, dist, dist), min_size=1, max_size=12),
       thr=st.floats(min_value=0.05, max_value=1.5), patience=st.integers(1, 3), ring=st.booleans())
def test_rule_walks_with_the_oracle_stopper(first, steps, thr, patience, ring):
    """Random per-iteration distances (step pair and drift pair, inpaint and ring weight): the pure rule and the oracle's
    stopper -- restated from earlystop.py:238-336 independently and pinned to the reference's traces -- agree on every
    iteration's counter, distance, drift consultation and verdict, and on when the anchor is held."""
    # the oracle stopper, scripted: iteration 0 compares x_t (fallback metric), later ones the x0 pair, the drift pair on demand
    o = _ScriptedOracleStopper(thr, patience, {"step": [(a, b) for a, b, _c, _d in steps], "drift": [(c, d) for _a, _b, c, d in steps]}, ring)
    state = es.StopState()
    # iteration 0: fallback metric on x_t -- one-element arrays whose weighted MSE is `first`
    x_before, x_after = np.zeros(1, np.float32), np.full(1, np.sqrt(np.float32(first)), np.float32)
    cur = orc.OracleState(None, None, np.zeros(1, np.float32))
    o.i = -1
    stop_o = o.step(x_before, x_after, None, cur)
    d0 = orc.weighted_mse(x_after, x_before, o.inpaint)
    state, rec = es.stop_rule((d0, 1.0, 0.0, 1.0, 0.0, 0.0), state, thr, o.patience_eff, have_prev=False, has_ring=ring, have_anchor=False)
    assert (rec.counter, rec.stopped, state.anchored) == (o.counter, stop_o, o.anchor is not None)
    for i, (a, b, c, d) in enumerate(steps):
        if stop_o:
            break
        prev, cur = cur, orc.OracleState(None, None, np.full(1, float(i + 1), np.float32))
        o.i = i
        anchored = o.anchor is not None
        stop_o = o.step(None, None, prev, cur)
        state, rec = es.stop_rule((a, 1.0 - 1e-12, b, 1.0 - 1e-12, c, d), state, thr, o.patience_eff, have_prev=True, has_ring=ring,
                                  have_anchor=anchored)
        t = o.trace[-1]
        assert rec.counter == t["counter"] and rec.stopped == t["stopped"] == stop_o and rec.dist == pytest.approx(t["dist"], rel=1e-12)
        assert state.anchored == (o.anchor is not None)
        assert rec.take_anchor == ((not anchored) and o.anchor is not None) and rec.drop_anchor == (anchored and o.anchor is None)


def test_options_contract():
    P = es.StopOptions.parse
    assert P(None, 0.0, 1, None) is None and P({}, 0.5, 0, None) is None and P({"lanpaint_semantic_stop": {"threshold": 0}}, 0.5, 1, None) is None
    o = P({"lanpaint_semantic_stop": {"threshold": 0.3, "patience": 2}}, 0.0, 1, None)
    assert (o.threshold, o.patience_eff, o.trace, o.tags) == (0.3, 3, None, (None, None, None))
    # the legacy min_steps knob is a floor on patience (min_steps - 1), ignored when unreadable or when patience is off
    assert P({"lanpaint_semantic_stop": {"threshold": 1.0, "patience": 1, "min_steps": 4}}, 0, 1, None).patience_eff == 4
    assert P({"lanpaint_semantic_stop": {"threshold": 1.0, "patience": 5, "min_steps": 4}}, 0, 1, None).patience_eff == 6
    assert P({"lanpaint_semantic_stop": {"threshold": 1.0, "patience": 1, "min_steps": "x"}}, 0, 1, None).patience_eff == 2
    assert P({"lanpaint_semantic_stop": {"threshold": 1.0, "patience": 0, "min_steps": 9}}, 0, 1, None) is None
    sink = []
    o = P({"lanpaint_semantic_trace": sink, "bench_case_id": "c", "bench_outer_step": 2, "bench_timestep": 0.1}, 0.5, 1, abs)
    assert o.trace is sink and o.tags == ("c", 2, 0.1) and o.distance_fn is abs and o.patience_eff == 2
    assert P({"lanpaint_semantic_trace": "not a list", "bench_case_id": "c"}, 0.5, 1, None).tags == (None, None, None)
    assert es.abt_scale(0.5) == 1.0 and es.abt_scale(0.0) == es.abt_scale(1.0) == es.abt_scale(-3.0) == es.abt_scale(7.0) == 0.0
    assert es.abt_scale(0.2) == pytest.approx(0.64)
    assert list(es.TRACE_KEYS) == ["case_id", "outer_step", "bench_timestep", "inner_step", "dist", "dist_inpaint", "dist_ring", "dist_drift",
                                   "threshold", "threshold_eff", "patience_counter", "patience_eff", "abt", "custom_dist", "stopped"]
    assert sorted(es.TRACE_KEYS) == list(load_golden("ve_earlystop")["trace_keys"])       # the reference's own record


def test_distance_fn_shapes():
    seen = []
    three = es.bind_distance_fn(lambda prev, cur, ctx: seen.append(("3", prev, cur, ctx)) or 1.0)
    kw = es.bind_distance_fn(lambda a, b, *, ctx=None: seen.append(("kw", a, b, ctx)) or 2.0)
    var_kw = es.bind_distance_fn(lambda a, b, **k: seen.append(("kwargs", a, b, k["ctx"])) or 2.5)
    var_pos = es.bind_distance_fn(lambda *a: seen.append(("args",) + a) or 2.75)
    legacy = es.bind_distance_fn(lambda cur, prev: seen.append(("2", cur, prev)) or 3.0)
    assert [f("P", "C", "X") for f in (three, kw, var_kw, var_pos, legacy)] == [1.0, 2.0, 2.5, 2.75, 3.0]
    assert seen == [("3", "P", "C", "X"), ("kw", "P", "C", "X"), ("kwargs", "P", "C", "X"), ("args", "P", "C", "X"), ("2", "C", "P")]
    assert es.bind_distance_fn(None) is None and es.bind_distance_fn(3) is None

    class OpaquePair:                 # no readable signature: three arguments first, the legacy pair when the CALL rejects them
        __signature__ = "unreadable"
        got = None

        def __call__(self, cur, prev):
            self.got = (cur, prev)
            return 4.0

    class OpaqueBroken(OpaquePair):   # ... but a TypeError raised INSIDE the metric is the user's bug and surfaces
        def __call__(self, prev, cur, ctx):
            raise TypeError("inside")
    two = OpaquePair()
    assert es.bind_distance_fn(two)("P", "C", "X") == 4.0 and two.got == ("C", "P")
    with pytest.raises(TypeError, match="inside"):
        es.bind_distance_fn(OpaqueBroken())("P", "C", "X")

    import torch
    assert es.scalar_distance(None) is None and es.scalar_distance(2) == 2.0 and es.scalar_distance(torch.tensor([0.25])) == 0.25
    with pytest.raises(TypeError, match="scalar"):
        es.scalar_distance(torch.zeros(2))
