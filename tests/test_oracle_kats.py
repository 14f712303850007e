"""Known-answer pins taken from the reference's own tests and from its functions
run in the build container (tests/golden/kat_*.npz), checked against the oracle."""
import math

import numpy as np
import pytest

from oracle import lanpaint_oracle as orc
from tests.helpers import load_golden


# --- reference tests/test_min_step_frac.py:18-41 -----------------------------------
@pytest.mark.parametrize("n,frac,min_frac,want", [
    (5, 0.1, 0.0, 5), (5, 0.01, 0.0, 5), (5, 0.2, 0.05, 5), (5, 0.05, 0.05, 5),
    (5, 0.04, 0.05, 4), (5, 0.025, 0.05, 2), (5, 0.005, 0.05, 0), (5, 0.0, 0.05, 0), (0, 0.01, 0.05, 0)])
def test_min_step_frac_effective_steps(n, frac, min_frac, want):
    assert orc.min_step_frac_effective_steps(n, frac, min_frac) == want


# --- reference tests/test_av_schedule.py:179-180, 322-324 ---------------------------
def test_step_size_scales_with_remaining_noise():
    eng = orc.OracleLanPaint(None, 1, 15.0, 1.0, 1.0, 0.2)
    eng.ndim = 3
    abt = np.full((1, 1, 8), 0.5, dtype=np.float32)
    abt[..., 5:] = 0.9
    step = 0.2 * np.maximum(1 - abt, 0.0)
    assert step.flatten()[0] == pytest.approx(0.1)
    assert step.flatten()[-1] == pytest.approx(0.02)
    one = np.ones_like(abt)
    _abt, dtx, dty, a_x, a_y, d_x, d_y = eng.step_coefficients((one, abt, one), step, one, one)
    adt = (a_x * dtx).flatten()
    assert adt[0] == pytest.approx(0.2) and adt[-1] == pytest.approx(0.2)


def test_step_coefficients_match_reference_prepare_step_size():
    kat = load_golden("kat_step_coefficients")
    for flow, sigma, msf, abt_ref, dtx, dty, a_x, a_y, d in kat["table"]:
        s = np.asarray([sigma], dtype=np.float32)
        ve, abt, ft = orc.times_from_sigma(s, bool(flow))
        assert float(abt[0]) == pytest.approx(abt_ref, rel=1e-6)
        eng = orc.OracleLanPaint(None, 5, 15.0, 5.0, 1.0, 0.2, is_flow=bool(flow), min_step_frac=msf)
        eng.ndim = 4
        step = eng._bcast(0.2 * np.maximum(1 - abt, np.float32(msf)))
        one = eng._bcast(abt ** 0)
        _abt, o_dtx, o_dty, o_ax, o_ay, o_dx, _ = eng.step_coefficients((ve, abt, ft), step, one, one)
        for got, want in ((o_dtx, dtx), (o_dty, dty), (o_ax, a_x), (o_ay, a_y), (o_dx, d)):
            assert float(np.ravel(got)[0]) == pytest.approx(want, rel=2e-6)


# --- SURVEY.md section 8a coefficient table (float64 closed form) --------------------
@pytest.mark.parametrize("abt,step,region,tag,e,k,std", [
    (0.2, 0.16, 0, "full", 0.818730741, 0.145015407, 0.513560102),
    (0.2, 0.16, 0, "half", 0.904837412, 0.076130071, 0.380808885),
    (0.2, 0.16, 1, "full", 0.301194186, 0.093174108, 0.348191913),
    (0.2, 0.16, 1, "half", 0.548811613, 0.060158452, 0.305244342),
    (0.2, 0.20, 0, "full", 0.778800780, 0.176959376, 0.561048550),
    (0.2, 0.20, 1, "half", 0.472366547, 0.070351127, 0.321842579),
    (0.5, 0.10, 0, "full", 0.818730751, 0.090634625, 0.406004900),
    (0.5, 0.10, 1, "half", 0.548811631, 0.037599031, 0.241316838)])
def test_region_coefficients_known_answers(abt, step, region, tag, e, k, std):
    abt32 = float(np.float32(abt))
    c = orc.region_coefficients(abt32, float(np.float32(step)), 5.0, 1.0)[region]
    assert c["e_" + tag] == pytest.approx(e, rel=2e-7)
    assert c["k_" + tag] == pytest.approx(k, rel=2e-7)
    assert c["std_" + tag] == pytest.approx(std, rel=2e-7)


# --- replace step, reference tests/test_av_schedule.py:203-219 ------------------------
def test_replace_step_values():
    class Ident:
        def __call__(self, x, t, model_options=None, seed=None):
            self.last = x
            return x, x
    m = Ident()
    eng = orc.OracleLanPaint(m, 0, 15.0, 1.0, 1.0, 0.2, is_flow=False)
    x = np.zeros((1, 1, 8), dtype=np.float32)
    ai = np.zeros_like(x)
    ai[..., 5:] = 1.0
    times = (np.float32([1.0]), np.float32([0.5]), np.float32([0.5]))
    times_a = (np.float32([0.25]), np.float32([0.9]), np.float32([0.2]))
    eng(x, np.zeros_like(x), np.ones_like(x), np.float32([0.5]), np.ones_like(x), times, None, 0, n_steps=0,
        current_times_audio=times_a, audio_indicator=ai)
    assert m.last.flatten()[0] == pytest.approx(0.5)
    assert m.last.flatten()[-1] == pytest.approx(0.2)


# --- mask index math: nearest-exact picks (reference tests/test_reshape_mask.py, test_videomask.py) ----
def test_nearest_exact_picks():
    assert list(orc.nearest_exact_src_index(2, 8)) == [2, 6]
    assert list(orc.nearest_exact_src_index(4, 16)) == [2, 6, 10, 14]
    picks = orc.nearest_exact_src_index(37, 124)
    assert 62 in picks and 60 not in picks and picks[18] == 62
    assert list(orc.nearest_exact_src_index(1, 3)) == [1]
    assert orc.nearest_exact_src_index(40, 100)[20] == 51      # inside the [50, 60) stroke of test_videomask.py:678-689


def test_nearest_exact_equals_torch_interpolate():
    import torch
    for n_in, n_out in [(8, 2), (16, 4), (124, 37), (864, 30), (480, 54), (7, 7), (5, 13), (3, 1), (100, 40), (1, 9)]:
        src = torch.arange(n_in, dtype=torch.float32).reshape(1, 1, n_in)
        want = torch.nn.functional.interpolate(src, size=(n_out,), mode="nearest-exact").reshape(-1).numpy()
        assert np.array_equal(orc.nearest_exact_src_index(n_out, n_in), want.astype(np.int64)), (n_in, n_out)


def test_reshape_mask_video_union_kats():
    m = np.zeros((8, 8, 8), dtype=np.float32)
    m[2, 5, 5] = 1.0
    m[6, 7, 7] = 1.0
    out = orc.reshape_mask(m, (1, 16, 2, 4, 4), video_inpainting=True)
    assert out.shape == (1, 16, 2, 4, 4)
    assert out[0, 0, 0].max() == 1.0 and out[0, 0, 1].max() == 1.0
    assert out[0, 0, 0, 2, 2] == 1.0 and out[0, 0, 1, 3, 3] == 1.0
    m = np.zeros((8, 8, 8), dtype=np.float32)
    m[3, 5, 5] = 1.0                      # frame 3 is not picked by 8 -> 2
    assert orc.reshape_mask(m, (1, 16, 2, 4, 4), video_inpainting=True).max() == 0.0
    m = np.zeros((3, 8, 8), dtype=np.float32)
    m[1, 5, 5] = 1.0
    assert orc.reshape_mask(m, (1, 16, 1, 4, 4), video_inpainting=True).max() == 1.0
    m = np.zeros((16, 4, 4), dtype=np.float32)
    m[6, 1, 1] = 1.0
    out = orc.reshape_mask(m, (1, 24, 4, 4, 4), video_inpainting=True)
    assert out[0, 0, 0].max() == 1.0 and out[0, 0, 3].max() == 1.0
    m = np.zeros((8, 1, 6, 8), dtype=np.float32)     # [F,1,H,W] from SetLatentNoiseMask
    m[2, 0, 2, 3] = 1.0
    m[6, 0, 4, 5] = 1.0
    out = orc.reshape_mask(m, (1, 24, 2, 6, 8), video_inpainting=True)
    assert out[0, 0, 0, 2, 3] == 1.0 and out[0, 0, 1, 4, 5] == 1.0


def test_reshape_mask_image_and_audio_kats():
    assert orc.reshape_mask(np.zeros((1, 4, 4)), (1, 16, 1, 8, 8)).shape == (1, 16, 1, 8, 8)
    assert orc.reshape_mask(np.zeros((1, 4, 4)), (1, 16, 1, 8, 8), comfy_060_or_newer=False).shape == (1, 16, 1, 8, 8)
    assert orc.reshape_mask(np.zeros((4, 4)), (2, 3, 8, 8)).shape == (2, 3, 8, 8)
    a = np.zeros(100, dtype=np.float32)
    a[50:60] = 1.0
    out = orc.reshape_mask(a, (1, 32, 2, 40))
    assert out.shape == (1, 32, 2, 40) and (out == 0).mean() > 0.7
    assert out[0, 0, 0, 20] == 1.0 and out[0, 0, 1, 20] == 1.0
    a4 = a.reshape(1, 1, 100, 1)
    out = orc.reshape_mask(a4, (1, 32, 2, 40))
    assert out[0, 0, 0, 20] == 1.0 and out[0, 0, 1, 20] == 1.0
    out = orc.reshape_mask(np.ones((1, 6, 8)), (1, 24, 37, 3, 4), video_inpainting=True)
    assert out.shape == (1, 24, 37, 3, 4) and out.min() == 1.0


def test_reshape_mask_matches_torch_pipeline():
    """Random masks through the oracle vs the same torch ops the reference calls
    (interpolate nearest-exact + max_pool3d), bit-exact."""
    import torch
    rng = np.random.default_rng(3)
    for (f, h, w), out_shape in [((9, 17, 13), (1, 4, 3, 5, 7)), ((124, 20, 12), (2, 3, 37, 6, 5)), ((5, 8, 8), (1, 2, 5, 8, 8))]:
        m = (rng.random((f, h, w)) > 0.7).astype(np.float32)
        got = orc.reshape_mask(m, out_shape, video_inpainting=True)
        t = torch.from_numpy(m)[None, None]
        t = torch.nn.functional.interpolate(t, size=out_shape[2:], mode="nearest-exact")
        t = torch.nn.functional.max_pool3d(t, kernel_size=(5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0))
        want = t.repeat(out_shape[0], out_shape[1], 1, 1, 1).numpy()
        assert np.array_equal(got, want)


# --- early-stop metric (earlystop.py:32-55) -------------------------------------------
def test_boundary_ring_and_wmse_match_reference():
    kat = load_golden("kat_boundary")
    rng = np.random.default_rng(int(kat["seed"]))
    for k in range(kat["masks"].shape[0]):
        m = (rng.random((2, 3, 9, 11)) > (0.3 + 0.15 * k)).astype(np.float32)
        a = rng.standard_normal(m.shape, dtype=np.float32)
        b = rng.standard_normal(m.shape, dtype=np.float32)
        assert np.array_equal(m, kat["masks"][k])
        inp = (1 - m).astype(np.float32)
        ring = orc.boundary_weight(m, inp)
        assert np.array_equal(ring, kat["rings"][k])
        assert orc.weighted_mse(a, b, inp) == pytest.approx(kat["mses"][k][0], rel=1e-5)
        assert orc.weighted_mse(a, b, ring) == pytest.approx(kat["mses"][k][1], rel=1e-5)


def test_abt_scale():
    assert orc.abt_scale(0.5) == 1.0 and orc.abt_scale(0.0) == 0.0 and orc.abt_scale(1.0) == 0.0
    assert orc.abt_scale(0.25) == pytest.approx(0.75)
    assert orc.abt_scale(-3) == 0.0 and orc.abt_scale(7) == 0.0


def test_times_from_sigma_forms():
    ve, abt, ft = orc.times_from_sigma(np.float32([2.0]), False)
    assert float(abt[0]) == pytest.approx(0.2) and float(ve[0]) == 2.0
    assert float(ft[0]) == pytest.approx(math.sqrt(0.8) / (math.sqrt(0.8) + math.sqrt(0.2)))
    ve, abt, ft = orc.times_from_sigma(np.float32([0.5]), True)
    assert float(abt[0]) == pytest.approx(0.5) and float(ve[0]) == pytest.approx(1.0) and float(ft[0]) == 0.5


# --- post-decode mask blend (reference nodes.py:592-647, 1049-1088) ------------------------
def test_mask_blend_and_video_merge_match_reference():
    g = load_golden("kat_mask_blend")
    for k in (1, 3, 7, 51):
        np.testing.assert_allclose(orc.gaussian_kernel_2d(k), g[f"gauss{k}"], rtol=2e-6, atol=1e-12)
    for idx in range(5):
        out = orc.mask_blend(g[f"blend{idx}_i1"], g[f"blend{idx}_i2"], g[f"blend{idx}_mask"], int(g[f"blend{idx}_k"]))
        np.testing.assert_allclose(out, g[f"blend{idx}_out"], atol=2e-6)
    for idx in range(4):
        out = orc.merge_video_with_mask(g[f"merge{idx}_orig"], g[f"merge{idx}_inp"], g[f"merge{idx}_mask"],
                                        int(g[f"merge{idx}_k"]))
        assert out.shape == g[f"merge{idx}_out"].shape
        np.testing.assert_allclose(out, g[f"merge{idx}_out"], atol=2e-6)
    with pytest.raises(ValueError):
        orc.merge_video_with_mask(np.zeros((4, 8, 8, 3), np.float32), np.zeros((4, 8, 8, 3), np.float32),
                                  np.zeros((2, 8, 8), np.float32), 3)
