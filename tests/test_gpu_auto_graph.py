"""graph="auto" -- what an engine built with NO optional keyword gets: the first call of a job runs
eagerly and times the backbone on the host; a cheap (launch-bound) backbone gets the job's later calls captured, the
capture is checked against an eager run of the same call before it is used, and a backbone that cannot be captured, keeps
hidden state, or is expensive leaves the engine on eager launches -- never a wrong result, never an exception."""
import time
import warnings

import numpy as np
import pytest

from tests import golden_cases as gc
from tests.stubs import LinearTupleModel

pytestmark = pytest.mark.gpu
SHAPE, N_SIG, N_THINK = (1, 4, 32, 32), 7, 3


@pytest.fixture(autouse=True)
def _no_forced_mode(monkeypatch):
    """These tests are about the DEFAULT mode: a suite run with LANPAINT_AMD_GRAPH=1 / 0 must not override it here."""
    monkeypatch.delenv("LANPAINT_AMD_GRAPH", raising=False)


def _schedule(eng, seed=321, shape=SHAPE, n_sig=N_SIG):
    import torch
    dev = "cuda"
    g = np.random.default_rng(17)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    y, noise = tt(g.standard_normal(shape, dtype=np.float32)), tt(g.standard_normal(shape, dtype=np.float32))
    mask = tt(gc.box_mask(shape))
    sig = gc.karras_sigmas(n_sig)[:-1]
    x = y + noise * float(sig[0])
    torch.manual_seed(seed)
    outs, mo = [], {}
    for i in range(len(sig)):
        s = torch.full((shape[0],), float(sig[i]), dtype=torch.float32, device=dev)
        den = eng(x, y, noise, s, mask, gc.times_from_sigma(s, False), mo, seed)
        outs.append(den)
        if i + 1 < len(sig):
            x = torch.lerp(den, x, float(sig[i + 1] / sig[i]))
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs] + [x.cpu().numpy()], torch.cuda.get_rng_state(dev).clone()


def test_default_engine_captures_a_cheap_backbone_and_equals_eager_launches():
    import torch
    from lanpaint_amd import LanPaint
    m_auto, m_eager = LinearTupleModel(), LinearTupleModel()
    auto = LanPaint(m_auto, N_THINK, 15.0, 5.0, 1.0, 0.2)                  # no optional keyword at all
    assert auto.graph == "auto" and auto.rng == "torch"
    got, state_a = _schedule(auto)
    want, state_e = _schedule(LanPaint(m_eager, N_THINK, 15.0, 5.0, 1.0, 0.2, graph=False))
    assert len(auto._graphs) == 1 and not auto._graph_blocked
    cap = next(iter(auto._graphs.values()))
    assert cap.fast and cap.final_in_graph
    assert m_auto.calls < m_eager.calls                       # the Python backbone stopped being called: replays
    assert auto.iterations_run == N_SIG * N_THINK
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    assert torch.equal(state_a, state_e)                      # the generator ends where eager launches leave it


def test_auto_mode_survives_a_backbone_that_cannot_be_captured():
    import torch
    from lanpaint_amd import LanPaint

    class Syncing(LinearTupleModel):
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            scale = 0.9 if float(t.reshape(-1)[0].item()) >= 0.0 else 0.5      # host sync: illegal while capturing
            return scale * x, 0.8 * x

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng = LanPaint(Syncing(), N_THINK, 15.0, 5.0, 1.0, 0.2)
        got, _ = _schedule(eng)
    assert eng._graph_blocked and not eng._graphs
    assert any("stays with eager launches" in str(x.message) for x in w)
    want, _ = _schedule(LanPaint(Syncing(), N_THINK, 15.0, 5.0, 1.0, 0.2, graph=False))
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # and the stream is usable afterwards
    assert torch.isfinite(torch.ones(4, device="cuda") * 2).all()


def test_auto_mode_rejects_a_capture_that_does_not_reproduce_eager():
    from lanpaint_amd import LanPaint

    class Stateful(LinearTupleModel):                       # python-side state a graph cannot see
        def __call__(self, x, t, model_options=None, seed=None):
            self._note(x, t)
            return (0.9 + 1e-3 * (self.calls % 7)) * x, 0.8 * x

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng = LanPaint(Stateful(), N_THINK, 15.0, 5.0, 1.0, 0.2)
        got, _ = _schedule(eng)
    assert eng._graph_blocked and not eng._graphs
    assert any("does not reproduce the eager one" in str(x.message) for x in w)
    assert all(np.isfinite(a).all() for a in got)


def test_auto_mode_leaves_an_expensive_backbone_alone():
    from lanpaint_amd import LanPaint

    class Slow(LinearTupleModel):
        def __call__(self, x, t, model_options=None, seed=None):
            time.sleep(3e-4)                                  # 300 us of host time per call: not a launch-bound loop
            return super().__call__(x, t, model_options=model_options, seed=seed)

    m = Slow()
    eng = LanPaint(m, N_THINK, 15.0, 5.0, 1.0, 0.2)
    got, _ = _schedule(eng, n_sig=4)
    assert not eng._graphs and not eng._graph_blocked and m.calls == 4 * (N_THINK + 1)
    want, _ = _schedule(LanPaint(Slow(), N_THINK, 15.0, 5.0, 1.0, 0.2, graph=False), n_sig=4)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
