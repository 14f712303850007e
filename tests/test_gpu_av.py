"""AV packs (MiniMax-H3 flat audio / video latents; reference lanpaint.py:60-74, 173-180) on the two-row coefficient table
(LP_FL_AV, round 4): a 0/1 stream indicator and per-row times mean every element sits on one of two per-row time sets, so
the kernels read two table rows per batch row and the indicator as bits instead of three full-size blended time tensors
and per-element exp / expm1.  Checked against the CPU oracle (which blends like the reference), against the reference's own
golden AV cases (tests/golden/av_flat_*.npz run through this path by the parity suite), and against the reference-shaped
per-element path of the same engine."""
import numpy as np
import pytest

from oracle import lanpaint_oracle as orc
from tests.helpers import assert_close
from tests.stubs import MODELS

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(shape, split, seed, soft=False, rows_differ=False):
    rng = np.random.default_rng(seed)
    b = shape[0]
    y = rng.standard_normal(shape, dtype=np.float32)
    noise = rng.standard_normal(shape, dtype=np.float32)
    ai = np.zeros(shape, dtype=np.float32)
    ai[..., split:] = 1.0                                    # the audio part of the flat pack
    mask = (rng.random(shape) > 0.45).astype(np.float32)
    if soft:
        mask = rng.random(shape, dtype=np.float32)
    t_v = np.float32([0.55 + (0.1 * r if rows_differ else 0.0) for r in range(b)])
    t_a = np.float32([0.30 + (0.05 * r if rows_differ else 0.0) for r in range(b)])
    if b > 1:            # (the reference blends `times * (1 - ai)`: per-row times of a batch have to broadcast against the latent)
        t_v, t_a = t_v.reshape((b,) + (1,) * (len(shape) - 1)), t_a.reshape((b,) + (1,) * (len(shape) - 1))
    times_v = orc.times_from_sigma(t_v, True)
    times_a = orc.times_from_sigma(t_a, True)
    corr = ((1.0 - ai) + np.float32(0.7) * ai).astype(np.float32)
    bs = (-1,) + (1,) * (len(shape) - 1)
    x = (t_v.reshape(bs) * noise + (1 - t_v.reshape(bs)) * y).astype(np.float32)
    t_v = np.ascontiguousarray(t_v)
    return dict(x=x, y=y, noise=noise, mask=mask, ai=ai, sigma=t_v, times_v=times_v, times_a=times_a, corr=corr)


def _run_oracle(c, n_steps, draws, model_options=None):
    it = iter(draws)
    o = orc.OracleLanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, is_flow=True, randn=lambda like: next(it))
    x = c["x"].copy()
    out = o(x, c["y"], c["noise"], c["sigma"], c["mask"], c["times_v"], model_options, 0,
            current_times_audio=c["times_a"], audio_indicator=c["ai"], audio_correction=c["corr"])
    return x, out, o


def _run_engine(c, n_steps, draws, model_options=None, **kw):
    import torch
    from lanpaint_amd import LanPaint
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    it = iter([tt(d) for d in draws])
    eng = LanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, rng=lambda like: next(it), **kw)
    x = tt(c["x"])
    out = eng(x, tt(c["y"]), tt(c["noise"]), tt(c["sigma"]), tt(c["mask"]), tuple(tt(t) for t in c["times_v"]), model_options, 0,
              current_times_audio=tuple(tt(t) for t in c["times_a"]), audio_indicator=tt(c["ai"]), audio_correction=tt(c["corr"]))
    torch.cuda.synchronize()
    return x.cpu().numpy(), out.cpu().numpy(), eng


@pytest.mark.parametrize("shape,split,soft,rows_differ", [
    ((1, 1, 24), 15, False, False),            # the reference test's flat pack
    ((1, 2, 1000), 333, False, False),          # seam inside a wave, not on a multiple of anything
    ((3, 4, 777), 500, False, True),            # batch rows on their own time pairs
    ((2, 1, 4096), 4096 - 64, True, True),      # soft mask values: per-element branch on top of the table rows
    ((1, 8, 66000), 40001, False, False),       # 528 000 elements: four elements per lane, seam inside a lane's quad
])
def test_av_table_path_matches_the_oracle_and_the_per_element_path(shape, split, soft, rows_differ, monkeypatch):
    n_steps = 3
    c = _case(shape, split, seed=len(shape) * 100 + split, soft=soft, rows_differ=rows_differ)
    rng = np.random.default_rng(9)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(2 * n_steps - 1)]
    x_o, out_o, _ = _run_oracle(c, n_steps, draws)
    x_t, out_t, eng = _run_engine(c, n_steps, draws)
    assert eng._desc.flags & (1 << 17), "the engine did not take the two-row table (LP_FL_AV)"
    assert_close(x_t, x_o, "AV table path: x vs oracle", rel=3e-5)
    assert_close(out_t, out_o, "AV table path: out vs oracle", rel=3e-5)
    monkeypatch.setenv("LANPAINT_AMD_AV_TABLE", "0")            # the reference-shaped path: blended full-size times, per element
    x_p, out_p, eng_p = _run_engine(c, n_steps, draws)
    assert not (eng_p._desc.flags & (1 << 17)) and eng_p._desc.flags & (1 << 8)
    assert_close(x_t, x_p, "AV table path vs per-element path: x", rel=3e-5)
    assert_close(out_t, out_p, "AV table path vs per-element path: out", rel=3e-5)


def test_av_pack_with_the_inner_early_stop_on_the_device():
    """The stop rule of earlystop.py evaluated on the device for an AV pack (round 3 kept the host stopper for per-element
    times): same iteration count, trace and outputs as the oracle's stopper, whose threshold scales with the mean of the
    BLENDED abt tensor (earlystop.py:105-113) -- the device forms it from the two rows' abt and the share of audio elements."""
    shape, split, n_steps = (2, 2, 600), 401, 10
    c = _case(shape, split, seed=4, rows_differ=True)
    rng = np.random.default_rng(10)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(2 * n_steps - 1)]
    ran = {}
    for thr in (5.0, 1e-9):
        mo_o = {"lanpaint_semantic_stop": {"threshold": thr, "patience": 2}}
        mo_e = {"lanpaint_semantic_stop": {"threshold": thr, "patience": 2}, "lanpaint_semantic_trace": []}
        x_o, out_o, o = _run_oracle(c, n_steps, draws, mo_o)
        x_e, out_e, eng = _run_engine(c, n_steps, draws, mo_e)
        assert eng._desc.flags & (1 << 17) and eng._ds is not None          # AV table + device-side stopper buffers in use
        assert eng.iterations_run == o.iterations_run, (thr, eng.iterations_run, o.iterations_run)
        ran[thr] = eng.iterations_run
        assert_close(x_e, x_o, f"thr={thr}: x", rel=3e-5)
        assert_close(out_e, out_o, f"thr={thr}: out", rel=3e-5)
        tr = mo_e["lanpaint_semantic_trace"]
        assert len(tr) == eng.iterations_run and (thr < 1 or tr[-1]["stopped"])
        st = o.last_stopper
        if st is not None:
            assert tr[0]["threshold_eff"] == pytest.approx(st.threshold_eff, rel=1e-5)
    assert ran[5.0] < n_steps and ran[1e-9] == n_steps


def test_indicator_that_is_not_binary_keeps_the_reference_shaped_path():
    shape = (1, 1, 64)
    c = _case(shape, 40, seed=1)
    c["ai"] = (c["ai"] * 0.5).astype(np.float32)              # a soft indicator: blended times really vary per element
    c["corr"] = ((1.0 - c["ai"]) + np.float32(0.7) * c["ai"]).astype(np.float32)
    rng = np.random.default_rng(11)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(3)]
    x_o, out_o, _ = _run_oracle(c, 2, draws)
    x_e, out_e, eng = _run_engine(c, 2, draws)
    assert eng._desc.flags & (1 << 8) and not (eng._desc.flags & (1 << 17))
    assert_close(x_e, x_o, "soft indicator: x", rel=3e-5)
    assert_close(out_e, out_o, "soft indicator: out", rel=3e-5)


def test_av_pack_replays_as_a_graph_bitwise_equal_to_eager_launches():
    """The table form is what makes an AV call capturable (round 3: eager only): its per-call inputs -- the two time pairs, the
    correction tensor -- go through workspace buffers the prologue refreshes.  Three sigma calls with different times and
    corrections, replayed vs eager launches under the same torch seed: bit for bit."""
    import torch
    from lanpaint_amd import LanPaint
    shape, split, n_steps = (1, 4, 3000), 1777, 3
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    c = _case(shape, split, seed=21)
    ai, y, noise, mask = tt(c["ai"]), tt(c["y"]), tt(c["noise"]), tt(c["mask"])
    res = {}
    for graph in (False, True):
        torch.manual_seed(77)
        eng = LanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, graph=graph)
        x, outs = tt(c["x"]), []
        for k, (tv, ta, cc) in enumerate(((0.8, 0.6, 0.9), (0.5, 0.3, 0.7), (0.2, 0.1, 0.5))):
            s_v, s_a = np.float32([tv]), np.float32([ta])
            corr = tt(((1.0 - c["ai"]) + np.float32(cc) * c["ai"]).astype(np.float32))        # a fresh tensor per call, like nodes.py
            out = eng(x, y, noise, tt(s_v), mask, tuple(tt(t) for t in orc.times_from_sigma(s_v, True)), None, 0,
                      current_times_audio=tuple(tt(t) for t in orc.times_from_sigma(s_a, True)), audio_indicator=ai,
                      audio_correction=corr)
            outs.append(out.clone())
            x = x + 0.1 * (out - x)
        torch.cuda.synchronize()
        res[graph] = (x.clone(), outs, len(eng._graphs))
    assert res[True][2] >= 1 and res[False][2] == 0
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("graph", [None, False])
def test_soft_indicator_under_inference_mode_over_several_sigma_calls(graph):
    """a non-0/1 indicator is cached as "soft"; under torch.inference_mode() -- how ComfyUI runs nodes --
    tensors carry no version counter, so the cache cannot tell whether the tensor was rewritten and the next look used to
    dereference the missing bit buffer (AttributeError inside the first sigma call: the eligibility check and the prologue
    both ask).  Three sigma calls with the same soft indicator tensor, default launch mode and eager: the reference-shaped
    per-element path every time, results equal to the oracle's."""
    import torch
    from lanpaint_amd import LanPaint
    shape, n_steps = (1, 2, 96), 2
    c = _case(shape, 60, seed=2)
    c["ai"] = (c["ai"] * 0.5).astype(np.float32)
    c["corr"] = ((1.0 - c["ai"]) + np.float32(0.7) * c["ai"]).astype(np.float32)
    rng = np.random.default_rng(12)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(3 * (2 * n_steps - 1))]
    it_o = iter(draws)
    o = orc.OracleLanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, is_flow=True, randn=lambda like: next(it_o))
    x_o, want = c["x"].copy(), []
    for _ in range(3):
        want.append(o(x_o, c["y"], c["noise"], c["sigma"], c["mask"], c["times_v"], None, 0, current_times_audio=c["times_a"],
                      audio_indicator=c["ai"], audio_correction=c["corr"]).copy())
    with torch.inference_mode():
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
        it = iter([tt(d) for d in draws])
        kw = {} if graph is None else {"graph": graph}
        eng = LanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, rng=lambda like: next(it), **kw)
        x, ai, corr = tt(c["x"]), tt(c["ai"]), tt(c["corr"])
        y, noise, mask, sigma = tt(c["y"]), tt(c["noise"]), tt(c["mask"]), tt(c["sigma"])
        tv, ta = tuple(tt(t) for t in c["times_v"]), tuple(tt(t) for t in c["times_a"])
        for k in range(3):
            out = eng(x, y, noise, sigma, mask, tv, None, 0, current_times_audio=ta, audio_indicator=ai, audio_correction=corr)
            assert eng._desc.flags & (1 << 8) and not (eng._desc.flags & (1 << 17))     # per-element path, not the table
            assert_close(out.cpu().numpy(), want[k], f"sigma call {k}: out", rel=3e-5)
        assert_close(x.cpu().numpy(), x_o, "x after three calls", rel=3e-5)


def test_binary_indicator_rewritten_to_soft_values_under_inference_mode_is_noticed():
    """A 0/1 indicator packed once and then rewritten IN PLACE to soft values while no version counter exists: the in-place
    re-pack of every call reads the "values other than 0 and 1" flag, so the call falls back to the per-element path instead of
    running on bits that binarise the new values at 0.5."""
    import torch
    from lanpaint_amd.lanpaint import pack_indicator
    with torch.inference_mode():
        ai = torch.zeros((1, 2, 128), device=DEV)
        ai[..., 64:] = 1.0
        first = pack_indicator(ai, ai.shape)
        assert first is not None and first[1] == 0.5 and first[2] is True
        again = pack_indicator(ai, ai.shape)
        assert again is not None and again[0] is first[0]                 # same bit buffer, re-derived in place
        ai.mul_(0.5)
        assert pack_indicator(ai, ai.shape) is None
        ai.fill_(1.0)
        back = pack_indicator(ai, ai.shape)                                # and a later binary rewrite is taken up again
        assert back is not None and back[1] == 1.0


def test_rows_with_different_audio_shares_keep_the_host_stopper():
    """the device-side stopper weights every row's (video, audio) abt pair by ONE audio share; the reference's
    threshold is the mean of the blended abt tensor, i.e. per-row shares.  An indicator whose rows differ in their share
    therefore takes the reference-shaped path with the host-side stopper, and matches the oracle's stopper."""
    shape, n_steps = (2, 1, 512), 6
    c = _case(shape, 300, seed=6, rows_differ=True)
    c["ai"][1, :, 100:] = 1.0                                              # row 1 holds more audio elements than row 0
    c["corr"] = ((1.0 - c["ai"]) + np.float32(0.7) * c["ai"]).astype(np.float32)
    rng = np.random.default_rng(13)
    draws = [rng.standard_normal(shape, dtype=np.float32) for _ in range(2 * n_steps - 1)]
    mo_o = {"lanpaint_semantic_stop": {"threshold": 5.0, "patience": 2}}
    mo_e = {"lanpaint_semantic_stop": {"threshold": 5.0, "patience": 2}, "lanpaint_semantic_trace": []}
    x_o, out_o, o = _run_oracle(c, n_steps, draws, mo_o)
    x_e, out_e, eng = _run_engine(c, n_steps, draws, mo_e)
    assert not (eng._desc.flags & (1 << 17))                               # not the two-row table with its single share
    assert eng.iterations_run == o.iterations_run
    assert_close(x_e, x_o, "rows with different audio shares: x", rel=3e-5)
    assert_close(out_e, out_o, "rows with different audio shares: out", rel=3e-5)
    # without a stopper the table path is still taken for such an indicator (the share only enters the stop threshold)
    _, _, eng2 = _run_engine(c, 2, draws[:3])
    assert eng2._desc.flags & (1 << 17)


@pytest.mark.parametrize("shape,split,patience", [((1, 4, 3000), 1777, 1), ((2, 2, 600), 401, 2), ((1, 8, 66000), 40001, 1)],
                         ids=["one_row", "two_rows_own_times", "16B_lanes_seam_in_a_quad"])
def test_av_pack_with_early_stop_replays_as_a_graph_bitwise_equal_to_the_watched_loop(shape, split, patience):
    """Round 5: the one combination round 4 still refused -- a GATED (replayed) early stop on an AV call.  The stopped launches
    of the replay re-emit the committed state with every element's own stream scale (table row 2 r / 2 r + 1 by the indicator
    bit).  Against the eager loop, which asks the mailbox after every iteration (itself pinned to the oracle's stopper above):
    the same iteration counts and trace, the same bits in x / out, torch's generator left in the same place -- over sigma calls
    on which the stop lands on different iterations (and once not at all)."""
    import torch
    from lanpaint_amd import LanPaint
    n_steps = 8
    rows_differ = shape[0] > 1
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    c = _case(shape, split, seed=33, rows_differ=rows_differ)
    ai, y, noise, mask = tt(c["ai"]), tt(c["y"]), tt(c["noise"]), tt(c["mask"])
    bs = (-1,) + (1,) * (len(shape) - 1)
    res = {}
    for graph in (False, True):
        torch.manual_seed(78)
        eng = LanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, graph=graph)
        x, runs = tt(c["x"]), []
        for k, (f, thr) in enumerate(((1.0, 5.0), (0.8, 5.0), (0.6, 1e-9), (0.4, 5.0))):
            s_v = (c["sigma"] * np.float32(f)).astype(np.float32)
            s_a = (np.asarray(c["times_a"][2]) * np.float32(f)).astype(np.float32)
            mo = {"lanpaint_semantic_stop": {"threshold": thr, "patience": patience}, "lanpaint_semantic_trace": []}
            it0 = eng.iterations_run
            out = eng(x, y, noise, tt(s_v), mask, tuple(tt(t) for t in orc.times_from_sigma(s_v, True)), mo, 0,
                      current_times_audio=tuple(tt(t) for t in orc.times_from_sigma(s_a, True)), audio_indicator=ai,
                      audio_correction=tt(c["corr"]))
            tr = mo["lanpaint_semantic_trace"]
            runs.append((out.clone(), x.clone(), eng.iterations_run - it0,
                         [(t["inner_step"], t["patience_counter"], t["stopped"]) for t in tr], [t["dist"] for t in tr]))
            x = x + 0.1 * (out - x)
        torch.cuda.synchronize()
        res[graph] = (runs, int(torch.cuda.default_generators[0].get_offset()), len(eng._graphs), eng._desc.flags)
    assert res[True][2] >= 1 and res[False][2] == 0
    assert res[True][3] & (1 << 17) and res[False][3] & (1 << 17)            # both on the two-row table
    for k, (g, e) in enumerate(zip(res[True][0], res[False][0])):
        assert g[2] == e[2] and g[3] == e[3], (k, g[2], e[2])
        np.testing.assert_allclose(g[4], e[4], rtol=1e-12)
        assert torch.equal(g[0], e[0]) and torch.equal(g[1], e[1]), f"sigma call {k}"
    assert res[True][1] == res[False][1]
    ran = [r[2] for r in res[True][0]]
    assert ran[2] == n_steps and any(r < n_steps for r in ran), ran


def test_av_early_stop_with_the_unfolded_gated_schedule():
    """LP_TUNE_ES_NO_FOLD (a developer switch of the descriptor): the verdict sits in a one-wave kernel between the launches and a
    stopped launch leaves through the kernel's early exit -- the path that used to re-emit with one row's scale."""
    import torch
    from lanpaint_amd import LanPaint, _cabi
    shape, split, n_steps = (1, 4, 3000), 1777, 8
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    c = _case(shape, split, seed=34)
    res = {}
    for tune in (0, _cabi.LP_TUNE_ES_NO_FOLD):
        torch.manual_seed(5)
        eng = LanPaint(MODELS["linear_tuple"](flow=True), n_steps, 15.0, 5.0, 1.0, 0.2, IS_FLOW=True, graph=True)
        eng._desc.tune = tune
        x = tt(c["x"])
        mo = {"lanpaint_semantic_stop": {"threshold": 5.0, "patience": 1}}
        out = eng(x, tt(c["y"]), tt(c["noise"]), tt(c["sigma"]), tt(c["mask"]), tuple(tt(t) for t in c["times_v"]), mo, 0,
                  current_times_audio=tuple(tt(t) for t in c["times_a"]), audio_indicator=tt(c["ai"]),
                  audio_correction=tt(c["corr"]))
        torch.cuda.synchronize()
        assert len(eng._graphs) == 1 and eng._desc.tune == tune
        res[tune] = (out.clone(), x.clone(), eng.iterations_run)
    a, b = res[0], res[_cabi.LP_TUNE_ES_NO_FOLD]
    assert a[2] == b[2] < n_steps
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
