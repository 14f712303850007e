"""Kernel-level checks through the C ABI: coefficient table, Philox stream, mask-edge ring,
weighted-MSE reduction, mask resample (index math bit-exact with torch's nearest-exact)."""
import ctypes

import numpy as np
import pytest

from oracle import lanpaint_oracle as orc
from tests import golden_cases as gc
from tests.helpers import run_product_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available()
    from lanpaint_amd import _cabi
    return _cabi.load()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def tt(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---------------------------------------------------------------- K1 coefficient table
@pytest.mark.parametrize("flow,sigmas,msf,lam,beta", [
    (False, [2.0], 0.0, 5.0, 1.0), (False, [2.0], 1.0, 5.0, 1.0), (True, [0.5], 0.0, 5.0, 1.0),
    (False, [14.6146, 0.0292, 1.0, 0.3], 0.0, 5.0, 1.0), (True, [0.97, 0.05, 0.5], 0.3, 8.0, 0.5)])
def test_coeffs_table_matches_closed_form(lib, flow, sigmas, msf, lam, beta):
    import torch
    from lanpaint_amd import _cabi
    s = np.asarray(sigmas, dtype=np.float32)
    ve, abt, _ft = orc.times_from_sigma(s, flow)
    rows = len(sigmas)
    h = _cabi.LpHyper()
    h.lambda_, h.beta, h.step_size, h.min_step_frac, h.is_flow, h.one_plus_lambda = lam, beta, 0.2, msf, int(flow), 1 + lam
    table = torch.empty((rows, _cabi.LP_COEF_STRIDE), dtype=torch.float32, device="cuda")
    ve_d, abt_d, s_d = tt(ve.astype(np.float32)), tt(abt.astype(np.float32)), tt(s)
    _cabi.check(lib.lp_coeffs(ctypes.byref(h), ve_d.data_ptr(), 1, abt_d.data_ptr(), 1, s_d.data_ptr(), 1, None, 0,
                              s_d.data_ptr(), 1, rows, table.data_ptr(), _stream()))
    t = table.cpu().numpy()
    for r in range(rows):
        a32 = np.float32(abt[r])
        oma = np.float32(1) - a32
        step = np.float32(0.2) * np.maximum(oma, np.float32(msf))
        want = orc.region_coefficients(float(a32), float(step), lam, beta)
        # region_coefficients uses (1-abt) in double; the table uses the fp32 (1-abt) like the reference
        for g in (0, 1):
            base = _cabi.LP_C_REGION0 if g == 0 else _cabi.LP_C_REGION1
            a = (1.0 + lam * g) / float(oma)
            dt = float(np.float32(step * np.float32(beta))) if g else float(step)
            for tag, tau, off in (("full", dt, 0), ("half", dt / 2, 3)):
                e, k = np.exp(-a * tau), -np.expm1(-a * tau) / a
                sd = np.sqrt(2 * (-np.expm1(-2 * a * tau) / (2 * a)))
                np.testing.assert_allclose(t[r, base + off: base + off + 3], [e, k, sd], rtol=3e-7, atol=1e-37)
            assert t[r, base + _cabi.LP_R_A] == pytest.approx(a, rel=2e-7)
            assert want[g]["A"] == pytest.approx(a, rel=1e-6)
        assert t[r, _cabi.LP_C_ABT] == a32 and t[r, _cabi.LP_C_OMA] == oma
        assert t[r, _cabi.LP_C_RSIGMA] == s[r] and t[r, _cabi.LP_C_VALID] == 1.0 and t[r, _cabi.LP_C_TMODEL] == s[r]
        scale = (np.sqrt(a32) + np.sqrt(np.float32(1) - a32)) if flow else np.sqrt(np.float32(1) + np.float32(ve[r]) ** 2)
        assert t[r, _cabi.LP_C_SCALE] == pytest.approx(float(scale), rel=2e-7)
        assert t[r, _cabi.LP_C_DTX] == step
        assert t[r, _cabi.LP_C_AX] == pytest.approx(1.0 / float(oma), rel=3e-7)


# ---------------------------------------------------------------- Philox
def _philox(lib, n, seed, offset, slot):
    import torch
    from lanpaint_amd import _cabi
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _cabi.check(lib.lp_philox_normal(out.data_ptr(), n, seed, offset, slot, _stream()))
    torch.cuda.synchronize()
    return out


def test_philox_normal_statistics(lib):
    z = _philox(lib, 1 << 22, 1234, 7, 0).double().cpu().numpy()
    n = z.size
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 4 * np.sqrt(2.0 / n)
    assert abs(np.mean(z ** 3)) < 4 * np.sqrt(15.0 / n)
    assert abs(np.mean(z ** 4) - 3.0) < 4 * np.sqrt(96.0 / n)
    assert np.abs(z).max() > 4.5                       # tails are populated
    for lag in (1, 2, 3, 4, 64, 4096):                 # no serial correlation between neighbouring elements
        assert abs(np.mean(z[:-lag] * z[lag:])) < 5 / np.sqrt(n)
    w = _philox(lib, 1 << 22, 1234, 7, 1).double().cpu().numpy()      # sine branch of the same pairs
    assert abs(w.var() - 1.0) < 4 * np.sqrt(2.0 / n) and abs(np.mean(z * w)) < 5 / np.sqrt(n)
    assert abs(np.mean(z * z * w * w) - 1.0) < 0.01                    # independent, not merely uncorrelated
    from scipy import stats
    assert stats.kstest(z[: 1 << 18], "norm").pvalue > 1e-4


def test_philox_streams_are_distinct_and_reproducible(lib):
    a = _philox(lib, 4096, 1, 0, 0).cpu().numpy()
    assert np.array_equal(a, _philox(lib, 4096, 1, 0, 0).cpu().numpy())
    for other in (_philox(lib, 4096, 2, 0, 0), _philox(lib, 4096, 1, 1, 0), _philox(lib, 4096, 1, 0, 1)):
        o = other.cpu().numpy()
        assert not np.array_equal(a, o)
        assert abs(np.corrcoef(a, o)[0, 1]) < 0.08
    assert np.array_equal(_philox(lib, 4099, 1, 0, 0).cpu().numpy()[:4096], a)    # ragged tail keeps the prefix


@pytest.mark.parametrize("name", ["ve_basic", "ve_odd_numel", "flow_batch", "ve_n1"])
def test_fused_in_kernel_noise_equals_host_supplied_philox(lib, name):
    """The noise the fused kernel generates is exactly lp_philox_normal(seed, launch, slot):
    feeding those tensors through the host-xi path gives bit-identical results, for the
    float4 and the scalar (odd numel) kernels alike."""
    case = gc.build_case(name)
    n_steps = case["n_steps"] if case["n_steps"] is not None else case["hyper"]["NSteps"]
    seed = 99
    fused = run_product_case(name, rng="philox", philox_seed=seed, graph=False)   # eager launch counters (a replay takes them from the device)
    calls = []

    def host_xi(like):
        k = len(calls)
        launch, slot = (k // 2, k % 2) if n_steps > 1 else (k, 0)
        calls.append((launch, slot))
        return _philox(lib, like.numel(), seed, (1 << 48) + launch, slot).reshape(like.shape)

    host = run_product_case(name, rng=host_xi, philox_seed=seed)
    assert len(calls) == max(0, 2 * n_steps - 1)
    assert np.array_equal(fused["x"], host["x"]) and np.array_equal(fused["out"], host["out"])


# ---------------------------------------------------------------- K4 ring + weighted MSE
@pytest.mark.parametrize("shape", [(2, 3, 9, 11), (1, 4, 128, 128), (1, 2, 70, 130), (3, 1, 1, 5), (1, 1, 33, 63)])
def test_boundary_ring_bit_exact(lib, shape):
    import torch
    from lanpaint_amd import _cabi
    rng = np.random.default_rng(11)
    m = (rng.random(shape) > 0.6).astype(np.float32)
    if shape[-1] > 64:
        m[..., 61:64] = 1.0          # strokes straddling the 62-column tile seam
        m[..., 31:33, :] = 0.0
    soft = m.copy()
    soft[m == 0] = rng.random(int((m == 0).sum())).astype(np.float32) * 0.5     # soft inpaint weights <= 0.5
    for mask in (m, soft):
        want = orc.boundary_weight(mask, (1 - mask).astype(np.float32))
        md = tt(mask)
        ring = torch.full_like(md, -7.0)
        b, c, h, w = shape
        _cabi.check(lib.lp_boundary_ring(md.data_ptr(), ring.data_ptr(), b * c, h, w, _stream()))
        assert np.array_equal(ring.cpu().numpy(), want)


def test_wmse_pair_matches_oracle(lib):
    import torch
    from lanpaint_amd.earlystop import StopState, WeightedSums, stop_rule
    rng = np.random.default_rng(3)
    # (594 elements: the 4-byte path; 65 536: 64 block sums; 786 432 and the video latent: more block sums than the totalling wave
    # has lanes / than the 1 024-block cap)
    for shape in [(2, 3, 9, 11), (1, 4, 128, 128), (1, 16, 5, 6, 7), (3, 4, 256, 256), (1, 16, 21, 60, 104)]:
        m = (rng.random(shape) > 0.5).astype(np.float32)
        a = rng.standard_normal(shape, dtype=np.float32)
        b = rng.standard_normal(shape, dtype=np.float32)
        met = WeightedSums(tt(m))
        six = met.six((tt(a), tt(b)), (tt(b), tt(a)))                     # pair A and -- as the drift pair -- the same two swapped
        assert six[4] == pytest.approx(six[0], rel=1e-12) and six[5] == pytest.approx(six[2], rel=1e-12)
        _st, rec = stop_rule(six, StopState(), 1e30, 2, have_prev=True, has_ring=met.ring is not None, have_anchor=False)
        d_in, d_ring = rec.dist_inpaint, rec.dist_ring
        inp = (1 - m).astype(np.float32)
        assert d_in == pytest.approx(orc.weighted_mse(a, b, inp), rel=1e-5)
        if len(shape) == 4:
            assert d_ring == pytest.approx(orc.weighted_mse(a, b, orc.boundary_weight(m, inp)), rel=1e-5)
        else:
            assert d_ring is None and met.ring is None
        assert met.inpaint_weight() == pytest.approx(float(inp.sum()), rel=1e-6)


# ---------------------------------------------------------------- K5 mask resample
@pytest.mark.parametrize("src_shape,out_shape,video", [
    ((9, 17, 13), (1, 4, 3, 5, 7), True), ((124, 20, 12), (2, 3, 37, 6, 5), True), ((5, 8, 8), (1, 2, 5, 8, 8), True),
    ((1, 6, 8), (1, 24, 37, 3, 4), True), ((64, 64), (2, 4, 8, 8), False), ((3, 40, 24), (3, 4, 5, 3), False),
    ((1, 37, 53), (4, 16, 1, 11, 7), False)])
def test_reshape_mask_kernel_bit_exact(lib, src_shape, out_shape, video):
    from lanpaint_amd.nodes import reshape_mask
    rng = np.random.default_rng(17)
    m = (rng.random(src_shape) > 0.7).astype(np.float32)
    want = orc.reshape_mask(m, out_shape, video_inpainting=video)
    got = reshape_mask(tt(m), out_shape, video_inpainting=video)
    assert tuple(got.shape) == tuple(out_shape)
    assert np.array_equal(got.cpu().numpy(), want)


# ---------------------------------------------------------------- post-decode mask blend
def test_mask_blend_matches_reference_golden(lib):
    import torch
    from lanpaint_amd import blend
    from tests.helpers import load_golden
    g = load_golden("kat_mask_blend")
    for idx in range(5):
        out = blend.mask_blend(tt(g[f"blend{idx}_i1"]), tt(g[f"blend{idx}_i2"]), tt(g[f"blend{idx}_mask"]), int(g[f"blend{idx}_k"]))
        np.testing.assert_allclose(out.cpu().numpy(), g[f"blend{idx}_out"], atol=3e-6)
    for idx in range(4):
        out = blend.merge_video_with_mask(tt(g[f"merge{idx}_orig"]), tt(g[f"merge{idx}_inp"]), tt(g[f"merge{idx}_mask"]),
                                          int(g[f"merge{idx}_k"]))
        assert tuple(out.shape) == g[f"merge{idx}_out"].shape
        np.testing.assert_allclose(out.cpu().numpy(), g[f"merge{idx}_out"], atol=3e-6)
    for k in (1, 3, 7, 51):
        np.testing.assert_allclose(blend.gaussian_kernel_2d(k).numpy(), g[f"gauss{k}"], rtol=1e-6)
    node_out, = blend.MaskBlend().blend_images(torch.from_numpy(g["blend2_i1"]), torch.from_numpy(g["blend2_i2"]),
                                               torch.from_numpy(g["blend2_mask"]), 7)       # CPU tensors in, like ComfyUI
    assert node_out.device.type == "cpu"
    np.testing.assert_allclose(node_out.numpy(), g["blend2_out"], atol=3e-6)
    with pytest.raises(ValueError):
        blend.mask_blend(tt(g["blend0_i1"]), tt(g["blend1_i1"]), tt(g["blend0_mask"]), 3)
    with pytest.raises(ValueError):
        blend.mask_blend(tt(g["blend0_i1"]), tt(g["blend0_i2"]), tt(g["blend0_mask"]), 4)


@pytest.mark.parametrize("shape,k", [((1, 64, 64, 3), 51), ((2, 70, 130, 4), 9), ((1, 1, 1, 3), 3), ((1, 1080, 1920, 3), 15),
                                     ((3, 37, 53, 1), 21)])
def test_mask_blend_matches_oracle_across_tiles(lib, shape, k):
    """Tile seams, halos wider than the image, both tile geometries, full-HD size."""
    from lanpaint_amd import blend
    rng = np.random.default_rng(k)
    b, h, w, c = shape
    i1, i2 = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    m = (rng.random((b, h, w)) > 0.9).astype(np.float32)
    got, smooth = blend._launch(tt(m), tt(i1), tt(i2), k, want_smooth=True)
    if h * w <= 70 * 130:
        want_smooth = orc.smooth_mask(m, k)
        np.testing.assert_allclose(smooth.cpu().numpy(), want_smooth, atol=3e-6)
        np.testing.assert_allclose(got.cpu().numpy(), orc.mask_blend(i1, i2, m, k), atol=3e-6)
    else:       # full HD: the oracle's k^2 loops are too slow; check a crop that contains a tile seam and an image edge
        crop = (slice(None), slice(0, 96), slice(1800, 1920))
        want = orc.smooth_mask(m[:, :96 + k, 1800 - k:], k)[:, :96, k:]
        np.testing.assert_allclose(smooth.cpu().numpy()[crop], want, atol=3e-6)
        s = smooth.cpu().numpy()[..., None]
        np.testing.assert_allclose(got.cpu().numpy(), i1 * (1 - s) + i2 * s, atol=1e-6)
        assert 0.0 <= s.min() and s.max() <= 1.0 + 1e-5


# ---------------------------------------------------------------- K1a sigma -> times + inner-step scalars
@pytest.mark.parametrize("flow", [False, True])
def test_sigma_times_bit_exact_vs_reference_cpu_ops(lib, flow):
    """lp_sigma_times reproduces the reference's eager fp32 tensor ops (nodes.py:242-252, 286, 299) bit for
    bit as the reference's CPU path evaluates them (IEEE-correct division / sqrt).  torch's own GPU kernels
    use the hardware reciprocal for `1 / t`, so against them the match is to 2 ulp."""
    import torch
    from lanpaint_amd import _cabi
    sched_cpu = torch.from_numpy(np.concatenate([gc.flow_sigmas(30)[:-1] if flow else gc.karras_sigmas(30)[:-1], [0.0]]).astype(np.float32))
    sched = sched_cpu.cuda()
    probes = list(sched_cpu[:-1].tolist()) + [0.37, 0.62, 0.05] + ([] if flow else [3.3, 11.0])

    def ref_ops(sigma):
        if flow:
            ft = sigma
            abt = (1 - ft) ** 2 / ((1 - ft) ** 2 + ft ** 2)
            return ft / (1 - ft), abt, ft
        abt = 1 / (1 + sigma ** 2)
        return sigma, abt, (1 - abt) ** 0.5 / ((1 - abt) ** 0.5 + abt ** 0.5)

    for rows in (1, 4):
        for sv in probes:
            sigma_cpu = torch.full((rows,), sv, dtype=torch.float32)
            sigma = sigma_cpu.cuda()
            buf = torch.empty(3 * rows + 2, dtype=torch.float32, device="cuda")
            _cabi.check(lib.lp_sigma_times(sigma.data_ptr(), rows, sched.data_ptr(), sched.numel(), int(flow), buf.data_ptr(),
                                           buf[3 * rows:].data_ptr(), _stream()))
            got = buf.cpu()
            ve, abt, ft = ref_ops(sigma_cpu)
            assert torch.equal(got[:rows], ve) and torch.equal(got[rows:2 * rows], abt), sv     # what n_eff and the kernels consume
            if flow:
                assert torch.equal(got[2 * rows:3 * rows], ft), sv
            else:   # VE: flow_t is informational (unused by a VE engine); torch's CPU pow(x, 0.5) is not always sqrt-exact
                torch.testing.assert_close(got[2 * rows:3 * rows], ft, rtol=2.5e-7, atol=0)
            assert int(got[3 * rows]) == int(torch.argmin(torch.abs(sched_cpu - torch.mean(sigma_cpu))))
            assert float(got[3 * rows + 1]) == pytest.approx(float((1.0 - abt).mean()), rel=2e-7)
            if rows == 1:
                assert float(got[3 * rows + 1]) == float((1.0 - abt).mean())
            for a, b in zip(ref_ops(sigma), (got[:rows], got[rows:2 * rows], got[2 * rows:3 * rows])):
                torch.testing.assert_close(a.cpu(), b, rtol=2.5e-7, atol=0)


@pytest.mark.parametrize("rows,sched_len", [(1, 1), (3, 64), (64, 65), (70, 200), (130, 1000)])
def test_sigma_times_wave_form_keeps_the_one_thread_order(lib, rows, sched_len):
    """The sigma algebra is run by one full wave (rows and schedule entries loaded lane-parallel): more rows / entries than
    lanes, different sigma per row, a schedule with REPEATED values around the mean.  The sums must be the sequential fp32
    sums of the one-thread form (row after row), the index torch.argmin's first minimum."""
    import torch
    from lanpaint_amd import _cabi
    rng = np.random.default_rng(rows * 1000 + sched_len)
    sig = rng.uniform(0.05, 14.0, rows).astype(np.float32)
    sched = np.sort(rng.uniform(0.0, 14.6, sched_len).astype(np.float32))[::-1].copy()
    seq_mean = np.float32(0.0)
    for v in sig:
        seq_mean = np.float32(seq_mean + v)
    seq_mean = np.float32(seq_mean / np.float32(rows))
    if sched_len >= 8:               # ties: the entry nearest the mean, four times in a row, and its mirror image on the other side
        k = int(np.argmin(np.abs(sched - seq_mean)))
        k = min(max(k, 2), sched_len - 5)
        sched[k:k + 4] = sched[k]
        sched[k + 4] = np.float32(seq_mean - (sched[k] - seq_mean))
    buf = torch.empty(3 * rows + 2, dtype=torch.float32, device="cuda")
    sig_d, sched_d = torch.from_numpy(sig).cuda(), torch.from_numpy(sched).cuda()
    _cabi.check(lib.lp_sigma_times(sig_d.data_ptr(), rows, sched_d.data_ptr(), sched_len, 0, buf.data_ptr(), buf[3 * rows:].data_ptr(),
                                   _stream()))
    got = buf.cpu().numpy()
    abt = (np.float32(1.0) / (np.float32(1.0) + sig * sig)).astype(np.float32)
    assert np.array_equal(got[:rows], sig) and np.array_equal(got[rows:2 * rows], abt)
    seq_oma = np.float32(0.0)
    for v in abt:
        seq_oma = np.float32(seq_oma + np.float32(np.float32(1.0) - v))
    assert np.float32(got[3 * rows + 1]) == np.float32(seq_oma / np.float32(rows))
    dd = np.abs((sched - seq_mean).astype(np.float32))
    assert int(got[3 * rows]) == int(np.argmin(dd))            # numpy's argmin: first minimum, like torch's


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4 * 128 * 128, 16 * 21 * 60 * 104 + 7])
@pytest.mark.parametrize("denoise", [False, True])
def test_pack_mask_bit_exact(lib, n, denoise):
    """lp_pack_mask (wave64 ballot) == the oracle's little-endian bit layout, tail bits zero."""
    import torch
    from lanpaint_amd import _cabi
    from oracle.lanpaint_oracle import pack_mask_bits
    rng = np.random.default_rng(n)
    m = (rng.random(n) < 0.4).astype(np.float32)
    src = torch.from_numpy(m).cuda()
    bits = torch.full((_cabi.mask_bits_bytes(n),), 0xAB, dtype=torch.uint8, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _cabi.check(lib.lp_pack_mask(src.data_ptr(), n, _cabi.LP_FL_MASK_DENOISE if denoise else 0, bits.data_ptr(),
                                 flag.data_ptr(), torch.cuda.current_stream().cuda_stream), "lp_pack_mask")
    assert np.array_equal(bits.cpu().numpy(), pack_mask_bits(m, denoise_mask=denoise))
    assert int(flag.item()) == 0


def test_pack_mask_flags_soft_masks_and_rejects_bad_arguments(lib):
    import torch
    import lanpaint_amd
    from lanpaint_amd import _cabi
    soft = torch.tensor([0.0, 1.0, 0.25, 1.0] * 40, device="cuda")
    with pytest.raises(ValueError):
        lanpaint_amd.pack_mask(soft)
    ok = lanpaint_amd.pack_mask(soft, denoise_mask=True)                 # the threshold makes it binary
    assert torch.equal(ok, 1 - (soft > 0.5).float()) and ok._lp_bits.numel() == _cabi.mask_bits_bytes(160)
    bits = torch.zeros(32, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.lp_pack_mask(soft.data_ptr(), 0, 0, bits.data_ptr(), None, s) < 0
    assert lib.lp_pack_mask(soft.data_ptr(), 160, _cabi.LP_FL_MASK_U8, bits.data_ptr(), None, s) < 0
    assert lib.lp_pack_mask(None, 160, 0, bits.data_ptr(), None, s) < 0


@pytest.mark.parametrize("flow", [False, True])
@pytest.mark.parametrize("shape", [(3, 4, 6, 5), (2, 4, 260, 256)])
def test_replace_launch_with_folded_coefficients_equals_lp_coeffs(lib, flow, shape):
    """LP_PH_COEFFS: the replace launch writes the coefficient table itself.  Table bitwise equal to lp_coeffs',
    x_t / x_in bitwise equal to the two-launch form (both vector widths; tiny rows: fewer groups than table lanes)."""
    import ctypes
    import torch
    from lanpaint_amd import _cabi
    torch.manual_seed(1)
    rows, n_el = shape[0], int(np.prod(shape))
    x, y, noise = (torch.randn(shape, device="cuda") for _ in range(3))
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sig = torch.linspace(0.3, 0.9, rows, device="cuda") if flow else torch.linspace(0.5, 7.0, rows, device="cuda")
    if flow:
        abt = (1 - sig) ** 2 / ((1 - sig) ** 2 + sig ** 2)
        ve, tm = sig / (1 - sig), sig.clone()
    else:
        abt, ve, tm = 1 / (1 + sig ** 2), sig.clone(), sig.clone()
    h = _cabi.LpHyper()
    h.lambda_, h.beta, h.step_size, h.min_step_frac, h.is_flow, h.one_plus_lambda = 5.0, 1.3, 0.2, 0.1, int(flow), 6.0
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for fold in (False, True):
        coef = torch.full((rows, _cabi.LP_COEF_STRIDE), float("nan"), device="cuda")
        x_t, x_in = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
        d = _cabi.LpStepDesc()
        d.n_el, d.el_per_row, d.rows = n_el, n_el // rows, rows
        d.flags = _cabi.LP_FL_FLOW if flow else 0
        d.replace_kind = _cabi.LP_REPLACE_FLOW if flow else _cabi.LP_REPLACE_VE
        d.lambda_, d.one_plus_lambda, d.beta, d.step_size, d.min_step_frac, d.noise_scale = 5.0, 6.0, 1.3, 0.2, 0.1, 1.0
        d.coef, d.x, d.noise, d.y, d.mask = coef.data_ptr(), x.data_ptr(), noise.data_ptr(), y.data_ptr(), mask.data_ptr()
        d.x_t, d.x_in = x_t.data_ptr(), x_in.data_ptr()
        d.phases = _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT
        if fold:
            d.phases |= _cabi.LP_PH_COEFFS
            d.t_ve, d.t_abt, d.t_rsig, d.t_model = ve.data_ptr(), abt.data_ptr(), sig.data_ptr(), tm.data_ptr()
            d.t_ve_stride = d.t_abt_stride = d.t_rsig_stride = d.t_model_stride = 1
            d.coef_out = coef.data_ptr()
        else:
            _cabi.check(lib.lp_coeffs(ctypes.byref(h), ve.data_ptr(), 1, abt.data_ptr(), 1, sig.data_ptr(), 1, None, 0,
                                      tm.data_ptr(), 1, rows, coef.data_ptr(), st), "lp_coeffs")
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        used = [c for c in range(_cabi.LP_COEF_STRIDE) if c < 32 or c == _cabi.LP_C_TMODEL]
        outs.append((coef[:, used].cpu(), x_t.cpu(), x_in.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # the fold is only valid as REPLACE | EMIT | COEFFS of a row-table launch, with its inputs present
    d.t_abt = None
    assert lib.lp_step(ctypes.byref(d), st) < 0


def test_replay_call_sequences_the_launches_and_reports_errors(lib):
    """lp_replay_call = [lp_coeffs] ; lp_step(replace) ; [hipGraphLaunch] ; lp_finalize in one FFI trip: same results
    as the separate entry points (here without a graph: graph_exec = NULL), first failing step's code returned."""
    import ctypes
    import torch
    from lanpaint_amd import _cabi
    torch.manual_seed(2)
    shape = (2, 4, 16, 12)
    rows, n_el = shape[0], int(np.prod(shape))
    x, y, noise, model_out = (torch.randn(shape, device="cuda") for _ in range(4))
    mask = (torch.rand(shape, device="cuda") < 0.5).float()
    sig = torch.tensor([2.0, 0.7], device="cuda")
    abt = 1 / (1 + sig ** 2)
    h = _cabi.LpHyper()
    h.lambda_, h.beta, h.step_size, h.min_step_frac, h.is_flow, h.one_plus_lambda = 5.0, 1.0, 0.2, 0.0, 0, 6.0
    st = torch.cuda.current_stream().cuda_stream

    def run(one_call):
        coef = torch.zeros((rows, _cabi.LP_COEF_STRIDE), device="cuda")
        x_t, x_in, out, x_back = (torch.empty(shape, device="cuda") for _ in range(4))
        d = _cabi.LpStepDesc()
        d.n_el, d.el_per_row, d.rows = n_el, n_el // rows, rows
        d.replace_kind, d.phases = _cabi.LP_REPLACE_VE, _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT
        d.lambda_, d.one_plus_lambda, d.beta, d.step_size, d.noise_scale = 5.0, 6.0, 1.0, 0.2, 1.0
        d.coef, d.x, d.noise, d.y, d.mask = coef.data_ptr(), x.data_ptr(), noise.data_ptr(), y.data_ptr(), mask.data_ptr()
        d.x_t, d.x_in = x_t.data_ptr(), x_in.data_ptr()
        f = _cabi.LpFinalDesc()
        f.n_el, f.model_out, f.y, f.mask = n_el, model_out.data_ptr(), y.data_ptr(), mask.data_ptr()
        f.x_src, f.x_dst, f.out = x_in.data_ptr(), x_back.data_ptr(), out.data_ptr()
        if one_call:
            c = _cabi.LpCallDesc()
            c.hyper, c.replace, c.final = ctypes.pointer(h), ctypes.pointer(d), ctypes.pointer(f)
            c.ve_sigma, c.abt, c.replace_sigma, c.t_model = sig.data_ptr(), abt.data_ptr(), sig.data_ptr(), sig.data_ptr()
            c.ve_stride = c.abt_stride = c.rs_stride = c.t_stride = 1
            c.rows, c.coef_table, c.graph_exec = rows, coef.data_ptr(), None
            _cabi.check(lib.lp_replay_call(ctypes.byref(c), st), "lp_replay_call")
        else:
            _cabi.check(lib.lp_coeffs(ctypes.byref(h), sig.data_ptr(), 1, abt.data_ptr(), 1, sig.data_ptr(), 1, None, 0,
                                      sig.data_ptr(), 1, rows, coef.data_ptr(), st), "lp_coeffs")
            _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
            _cabi.check(lib.lp_finalize(ctypes.byref(f), st), "lp_finalize")
        torch.cuda.synchronize()
        return [t.cpu() for t in (coef, x_t, x_in, out, x_back)], (d, f)

    a, _ = run(False)
    b, (d, f) = run(True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert torch.isfinite(b[3]).all()
    c = _cabi.LpCallDesc()
    assert lib.lp_replay_call(ctypes.byref(c), st) < 0                      # nothing set
    assert lib.lp_replay_call(None, st) < 0
    d.x = None                                                              # the replace step must fail ...
    c.replace, c.final = ctypes.pointer(d), ctypes.pointer(f)
    marker = torch.full((4,), 7.0, device="cuda")
    f.out = marker.data_ptr()
    assert lib.lp_replay_call(ctypes.byref(c), st) < 0
    torch.cuda.synchronize()
    assert torch.equal(marker.cpu(), torch.full((4,), 7.0))                 # ... before lp_finalize is reached


def test_uint8_mask_flag_through_the_c_abi(lib):
    """LP_FL_MASK_U8 (kept for direct C-ABI users; the engine packs such masks to bits): the run-time kernel gives
    bitwise the results of the fp32-mask hot kernels, replace and steady step."""
    import ctypes
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for phase in (_cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT, _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT):
        res = []
        for fmt in ("f32", "u8"):
            d, keep, _n = bench.standalone_step(_cabi, "c1_sd15", dev, phase, mask_format=fmt)
            assert bool(d.flags & _cabi.LP_FL_MASK_U8) == (fmt == "u8")
            _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
            torch.cuda.synchronize()
            bufs = keep[0]
            res.append([bufs[k].clone().cpu() for k in ("x_t", "C", "x_in")])
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_image_encode_decode_nodes_on_the_kernels(lib):
    """LanPaint_ImageEncode snaps the mask with lp_reshape_mask (== torch's nearest-exact on the GPU, bit for bit);
    LanPaint_ImageDecode merges with lp_mask_blend (== the oracle's merge_video_with_mask).  Host tensors in and
    out, as ComfyUI hands them to nodes."""
    import torch
    import torch.nn.functional as F
    from lanpaint_amd import nodes
    from oracle import lanpaint_oracle as orc

    class VAE:
        def encode(self, image):
            b, h, w, _c = image.shape
            return torch.zeros((b, 4, (h + 7) // 8, (w + 7) // 8))

        def decode(self, z):
            g = torch.Generator().manual_seed(1)
            return torch.rand((z.shape[0], z.shape[-2] * 8, z.shape[-1] * 8, 3), generator=g)

    g = torch.Generator().manual_seed(0)
    image = torch.rand((1, 124, 203, 3), generator=g)
    mask = (torch.rand((124, 203), generator=g) < 0.4).float()
    latent, = nodes.LanPaint_ImageEncode().encode(image, VAE(), mask=mask)
    want = F.interpolate(mask.cuda()[None, None], size=(16, 26), mode="nearest-exact")[0, 0].cpu()
    assert latent["noise_mask"].device.type == "cpu" and tuple(latent["noise_mask"].shape) == (1, 1, 16, 26)
    assert torch.equal(latent["noise_mask"][0, 0], want)
    out, = nodes.LanPaint_ImageDecode().decode(latent, VAE(), image=image, mask=mask, blend_overlap=9)
    assert out.device.type == "cpu" and tuple(out.shape) == tuple(image.shape)
    dec = VAE().decode(latent["samples"])
    dec = F.interpolate(dec.movedim(-1, 1), size=(124, 203), mode="bilinear", align_corners=False).movedim(1, -1)
    ref = orc.merge_video_with_mask(image.numpy(), dec.numpy(), mask.numpy(), 9)
    np.testing.assert_allclose(out.numpy(), ref, atol=3e-6)


@pytest.mark.parametrize("n", [1, 5, 255, 256, 257, 1000, 4 * 128 * 128, 16 * 21 * 60 * 104, 4 * 16 * 21 * 60 * 104 + 3])
def test_torch_normal_reproduces_torch_randn_bit_for_bit(lib, n):
    """LP_RNG_TORCH's per-element generator == torch.randn on this device: same (seed, offset) in, the same bits out,
    and the generator-offset increment the engine books per draw == what torch itself consumes."""
    import torch
    from lanpaint_amd import LanPaint
    dev = torch.device("cuda", 0)
    gen = LanPaint._generator(dev)
    bg, inc = LanPaint._randn_policy(dev, n)
    torch.manual_seed(4321 + n)
    st = torch.cuda.current_stream().cuda_stream
    for _rep in range(3):                       # consecutive draws: the offset moves on
        seed, off = gen.initial_seed(), gen.get_offset()
        ref = torch.randn(n, device=dev)
        assert gen.get_offset() - off == inc
        out = torch.empty(n, device=dev)
        _cabi_check = __import__("lanpaint_amd")._cabi.check
        _cabi_check(lib.lp_torch_normal(out.data_ptr(), n, seed, off, bg, st), "lp_torch_normal")
        assert torch.equal(out, ref)
    x = torch.zeros((2, 3, 5, 7), device=dev)
    seed, off = gen.initial_seed(), gen.get_offset()
    ref = torch.randn_like(x)
    out = torch.empty_like(x)
    bg, _ = LanPaint._randn_policy(dev, x.numel())
    assert lib.lp_torch_normal(out.data_ptr(), x.numel(), seed, off, bg, st) == 0
    assert torch.equal(out, ref)
    assert lib.lp_torch_normal(out.data_ptr(), x.numel(), seed, off + 1, bg, st) < 0      # offsets come in fours


def test_torch_normal_equals_torch_randn_over_a_quarter_billion_draws(lib):
    """Round 5 restated the device library's logf / sqrtf inside the Box-Muller of LP_RNG_TORCH for the arguments it can receive
    (lp_common.h::bm_logf / bm_sqrtf: no denormal rescaling, no infinity select).  That is only right if no draw ever needs the
    dropped range handling and the kept arithmetic is the library's own: 4 x 2^26 draws from different generator states against
    torch.randn, BITWISE (torch.equal would let -0.0 pass for +0.0: the sign of a zero root is part of the restatement)."""
    import torch
    from lanpaint_amd import LanPaint, _cabi
    dev = torch.device("cuda", 0)
    gen = LanPaint._generator(dev)
    n = 1 << 26
    bg, inc = LanPaint._randn_policy(dev, n)
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(n, device=dev)
    for seed in (0, 1, 20250924, 2 ** 40 + 7):
        torch.manual_seed(seed)
        gen.set_offset(4 * (seed % 1000))
        s, off = gen.initial_seed(), gen.get_offset()
        ref = torch.randn(n, device=dev)
        assert gen.get_offset() - off == inc
        _cabi.check(lib.lp_torch_normal(out.data_ptr(), n, s, off, bg, st), "lp_torch_normal")
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), seed
        assert torch.isfinite(ref).all() and float(ref.abs().max()) > 5.0          # the tails were visited


@pytest.mark.parametrize("rng", ["philox", "torch"])
@pytest.mark.parametrize("kind", ["temporal", "box", "blob"])
@pytest.mark.parametrize("phase", ["steady", "first", "last"])
def test_region_aware_streams_change_nothing_but_the_traffic(lib, kind, phase, rng):
    """Bit-packed mask, streaming size (VEC = 4): waves whose 256 mask bits are all 0 / all 1 skip the streams their
    region does not read (x0_BIG + y / x0).  Same launch with LP_FL_NO_REGION_SKIP: bitwise equal x_t, C, x_in --
    on a mask of large uniform regions (temporal), one with mixed waves only (box: 52-element runs) and a disc.
    rng = "torch": the ATen-strided kernels, which take the decision per slot (64 consecutive elements of a wave)."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    ph = {"steady": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
          "first": _cabi.LP_PH_POST_FIRST | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
          "last": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_EMIT}[phase]
    d, keep, n_el = bench.standalone_step(_cabi, "c5_wan", dev, ph, mask_kind=kind, mask_format="bits")
    bufs = keep[0]
    if rng == "torch":
        from lanpaint_amd import LanPaint
        d.rng_kind = _cabi.LP_RNG_TORCH
        d.rng_bg, d.rng_inc = LanPaint._randn_policy(dev, n_el)
        d.rng_seed = 99
    st = torch.cuda.current_stream().cuda_stream
    x_t0, c0 = bufs["x_t"].clone(), bufs["C"].clone()
    res = []
    for extra in (0, _cabi.LP_FL_NO_REGION_SKIP):
        bufs["x_t"].copy_(x_t0)
        bufs["C"].copy_(c0)
        bufs["x_in"].zero_()
        d.flags = (d.flags & ~_cabi.LP_FL_NO_REGION_SKIP) | extra
        d.rng_offset = 8 if rng == "torch" else 7
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        res.append([bufs[k].clone() for k in ("x_t", "C", "x_in")])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][0], x_t0) and torch.isfinite(res[0][0]).all()


@pytest.mark.parametrize("kind", ["temporal", "box", "blob"])
def test_region_aware_replace_and_finalize_change_nothing_but_the_traffic(lib, kind):
    """Round 4: the two launches around the think loop follow the same rule at streaming sizes.  The replace launch (with the
    coefficient table folded in, lanpaint.py:89-92): a wave of 256 inpaint elements keeps its x and reads neither noise nor known
    latent.  lp_finalize (lanpaint.py:154,156): such a wave never reads y, a wave of known elements never reads the model
    output.  Same launches with LP_FL_NO_REGION_SKIP: bitwise equal results, on uniform regions, mixed waves only, and a disc."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    d, keep, n_el = bench.standalone_step(_cabi, "c5_wan", dev,
                                              _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT | _cabi.LP_PH_COEFFS, mask_kind=kind, mask_format="bits")
    bufs, mask, coef, sig, ve, abt = keep
    d.t_ve, d.t_abt, d.t_rsig, d.t_ve_stride, d.t_abt_stride, d.t_rsig_stride = ve.data_ptr(), abt.data_ptr(), sig.data_ptr(), 1, 1, 1
    d.coef_out = coef.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for extra in (0, _cabi.LP_FL_NO_REGION_SKIP):
        for k in ("x_t", "C", "x_in"):
            bufs[k].fill_(-7.0)
        d.flags = (d.flags & ~_cabi.LP_FL_NO_REGION_SKIP) | extra
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        res.append([bufs[k].clone() for k in ("x_t", "C", "x_in")])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    m = mask.reshape(bufs["x"].shape)
    x_want = bufs["x"] * (1 - m) + (bufs["y"] + bufs["noise"] * sig.view(-1, *([1] * (m.ndim - 1)))) * m
    scale = torch.sqrt(1.0 + ve * ve).view(-1, *([1] * (m.ndim - 1))) if not bench.WORKLOADS["c5_wan"][1] else None
    assert torch.isfinite(res[0][0]).all() and not (res[0][0] == -7.0).any()
    if scale is not None:
        assert torch.allclose(res[0][0], x_want / scale, rtol=1e-5, atol=1e-6)

    out = [torch.full_like(bufs["x"], -3.0) for _ in range(2)]
    xd = [torch.full_like(bufs["x"], -3.0) for _ in range(2)]
    for j, extra in enumerate((0, _cabi.LP_FL_NO_REGION_SKIP)):
        f = _cabi.LpFinalDesc()
        f.n_el, f.flags = n_el, _cabi.LP_FL_MASK_BITS | extra
        f.model_out, f.y, f.mask = bufs["x0"].data_ptr(), bufs["y"].data_ptr(), mask._lp_bits.data_ptr()
        f.x_src, f.x_dst, f.out = bufs["x_t"].data_ptr(), xd[j].data_ptr(), out[j].data_ptr()
        _cabi.check(lib.lp_finalize(ctypes.byref(f), st), "lp_finalize")
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and torch.equal(xd[0], xd[1]) and torch.equal(xd[0], bufs["x_t"])
    assert torch.equal(out[0], bufs["x0"] * (1 - m) + bufs["y"] * m)


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("kind", ["temporal", "box", "blob"])
@pytest.mark.parametrize("phase", ["steady", "first", "last"])
def test_half_width_streams_in_lane_pairs_equal_the_eight_byte_path(lib, kind, phase, half):
    """Round 4: bf16 / fp16 backbone outputs (and a half-width x_in) of a streaming launch are read / written 16 bytes per
    LANE PAIR (even lane loads, DPP hands the odd lane its half).  The same launch on head / x_in buffers that sit 8 bytes off
    a 16-byte boundary cannot take that path (lp_step routes it to the run-time kernel with 8-byte accesses): x_t, C and x_in
    must come out bit for bit the same -- with the region-aware stream skipping (which predicates the pair loads) and
    without it, on a mask of large uniform regions, one with mixed waves only and a disc."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16 if half == "bf16" else torch.float16
    ph = {"steady": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
          "first": _cabi.LP_PH_POST_FIRST | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
          "last": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_EMIT}[phase]
    d, keep, n_el = bench.standalone_step(_cabi, "c5_wan", dev, ph, model_dtype=dt, mask_kind=kind, mask_format="bits")
    bufs = keep[0]
    # copies of the two heads and an x_in buffer 8 bytes (4 half elements) off a 16-byte boundary
    pad = {k: torch.zeros(n_el + 8, dtype=dt, device=dev) for k in ("x0", "x0b", "x_in")}
    off = {k: pad[k][4:4 + n_el] for k in pad}
    off["x0"].copy_(bufs["x0"].reshape(-1))
    off["x0b"].copy_(bufs["x0b"].reshape(-1))
    assert all(t.data_ptr() % 16 == 8 for t in off.values()) and all(bufs[k].data_ptr() % 16 == 0 for k in ("x0", "x0b", "x_in"))
    st = torch.cuda.current_stream().cuda_stream
    x_t0, c0 = bufs["x_t"].clone(), bufs["C"].clone()
    res = {}
    for skip in (0, _cabi.LP_FL_NO_REGION_SKIP):
        for path in ("pair", "eight"):
            bufs["x_t"].copy_(x_t0)
            bufs["C"].copy_(c0)
            src = bufs if path == "pair" else off
            src["x_in"].zero_()
            d.x0, d.x0_big, d.x_in = src["x0"].data_ptr(), src["x0b"].data_ptr(), src["x_in"].data_ptr()
            d.flags = (d.flags & ~_cabi.LP_FL_NO_REGION_SKIP) | skip
            d.rng_offset = 11
            _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
            torch.cuda.synchronize()
            res[(skip, path)] = [bufs["x_t"].clone(), bufs["C"].clone(), src["x_in"].reshape(-1).clone()]
    ref = res[(0, "eight")]
    for key, got in res.items():
        for a, b in zip(got, ref):
            assert torch.equal(a, b), key
    assert not torch.equal(ref[0], x_t0) and torch.isfinite(ref[0]).all() and torch.isfinite(ref[2].float()).all()
    assert float(ref[2].float().abs().max()) > 0.0


@pytest.mark.parametrize("n", [1, 2, 4])
@pytest.mark.parametrize("stops", [False, True])
def test_self_closing_gated_loop_through_the_c_abi(lib, stops, n):
    """LP_FL_ES_CLOSE straight through lp_step: a gated loop of four launches on a latent small enough for the folded
    verdict (also loops of one and two launches).  Without the flag the last launch is followed by the closing decision kernel, with it the launch accounts
    its own iteration and posts "done" itself -- same x_t / C / x_in bits, same n_ran and running count in the state and
    in the mailbox, for a loop that runs to its end (the backbone output moves every iteration) and for one that stops
    after three iterations (a constant output: distance exactly 0); the flag without LP_FL_ES_GATED, or LP_FL_ES_GATED
    without LP_FL_ES, is refused."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    from lanpaint_amd.lanpaint import _DeviceStop
    dev = torch.device("cuda", 0)
    S, F, P, E = _cabi.LP_PH_POST_STEADY, _cabi.LP_PH_POST_FIRST, _cabi.LP_PH_PRE_HALF, _cabi.LP_PH_EMIT
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for close in (0, _cabi.LP_FL_ES_CLOSE):
        d, keep, n_el = bench.standalone_step(_cabi, "c1_sd15", dev, S | P | E)
        bufs = keep[0]
        torch.manual_seed(3)
        for k in ("x_t", "C"):
            bufs[k].copy_(torch.randn_like(bufs[k]))
        ds = _DeviceStop(bufs["x_t"], n)
        mailbox = torch.zeros(_cabi.LP_ES_TRACE0 + 8 * n, dtype=torch.float64, device=dev)
        base = d.flags
        d.es, d.es_partials, d.es_host = ds.state.data_ptr(), ds.partials.data_ptr(), mailbox.data_ptr()
        d.es_xte = ds.x_te.data_ptr()
        for k in range(3):
            d.es_x0s[k] = ds.x0s[k].data_ptr()
        d.es_threshold, d.es_patience_eff, d.es_n_steps, d.es_seq_base = 1e-30, 2, n, 1000
        # a replace-like launch that only resets the stop state (phases EMIT, no flags needed)
        d.phases, d.flags, d.es_reset = E, base, 1
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "reset")
        d.es_reset = 0
        for i in range(n):
            last = i == n - 1
            d.phases = (F if i == 0 else S) | (0 if last else P) | E
            d.flags = base | _cabi.LP_FL_ES | _cabi.LP_FL_ES_GATED | (close if last else 0)
            d.es_index, d.rng_offset = i, 10 + i
            if not stops:
                bufs["x0"].add_(0.05 * torch.randn_like(bufs["x0"]))
            _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        raw = ds.state.cpu().numpy().tobytes()
        per = ctypes.sizeof(_cabi.LpEsState)
        slots = [_cabi.LpEsState.from_buffer_copy(raw[k * per:(k + 1) * per]) for k in range(2)]
        n_ran, total = max(s.n_ran for s in slots), max(s.total_ran for s in slots)     # the slot the loop ended in
        mb = mailbox.cpu().numpy()
        res.append(([bufs[k].clone() for k in ("x_t", "C", "x_in")], n_ran, total, int(mb[1]), int(mb[6]),
                    int(mb.view(np.int64)[0])))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    ran = min(3, n) if stops else n          # (patience: the constant output stops the loop after three iterations)
    assert res[0][1:] == (ran, ran, ran, ran, 1000 + _cabi.LP_ES_SEQ_DONE), res[0][1:]
    assert res[1][1:] == res[0][1:], res[1][1:]
    d.flags = base | _cabi.LP_FL_ES | _cabi.LP_FL_ES_CLOSE                      # CLOSE without GATED
    assert lib.lp_step(ctypes.byref(d), st) == _cabi.LP_E_INVALID
    d.flags = base | _cabi.LP_FL_ES_GATED                                        # GATED without ES
    assert lib.lp_step(ctypes.byref(d), st) == _cabi.LP_E_INVALID


@pytest.mark.parametrize("phase", ["steady", "last"])
def test_shared_divisor_emit_equals_ieee_division(lib, phase):
    """the flow-model emit `x_t / c` of the 16-byte-per-lane kernels goes through lp_common.h::div_shared (one
    reciprocal per lane + one residual correction per element), which equals IEEE division only while `1.0f / c` is correctly
    rounded -- a property of the build flags (pinned in lanpaint_amd/build.py: -fhip-fp32-correctly-rounded-divide-sqrt,
    -fno-fast-math).  The one-element-per-lane kernels still divide.  Same launch, both vector widths forced through the tune
    switches: bitwise equal x_t, C and model-space x_in on the video latent (2 M quotients per launch, flow scale c)."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    ph = {"steady": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_PRE_HALF | _cabi.LP_PH_EMIT,
          "last": _cabi.LP_PH_POST_STEADY | _cabi.LP_PH_EMIT}[phase]
    d, keep, n_el = bench.standalone_step(_cabi, "c5_wan", dev, ph, mask_kind="temporal", mask_format="bits")
    bufs = keep[0]
    bufs["x_t"].mul_(37.5)                      # quotients over a few binades
    st = torch.cuda.current_stream().cuda_stream
    x_t0, c0 = bufs["x_t"].clone(), bufs["C"].clone()
    res = []
    for tune in (_cabi.LP_TUNE_VEC4, _cabi.LP_TUNE_VEC1):
        bufs["x_t"].copy_(x_t0)
        bufs["C"].copy_(c0)
        bufs["x_in"].zero_()
        d.tune, d.rng_offset = tune, 5
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        res.append([bufs[k].clone() for k in ("x_t", "C", "x_in")])
    for name, a, b in zip(("x_t", "C", "x_in"), *res):
        assert torch.equal(a, b), name
    assert torch.isfinite(res[0][2]).all() and not torch.equal(res[0][2], torch.zeros_like(res[0][2]))


def test_ve_replace_launch_shared_divisor_equals_ieee_division(lib):
    """The VE replace launch `x_t = x / sqrt(1 + sigma^2)` (lanpaint.py:96-99) at 16 bytes per lane divides through
    lp_common.h::div_shared with the reciprocal of the row's scale (round 5); one element per lane divides.  The same fused
    replace + coefficient-table launch on an SDXL batch, both widths forced: bitwise equal x_t, model-space x_in and table."""
    import torch
    import bench
    from lanpaint_amd import _cabi
    dev = torch.device("cuda", 0)
    ph = _cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT | _cabi.LP_PH_COEFFS
    d, keep, n_el = bench.standalone_step(_cabi, "c3_sdxl_b4", dev, ph, mask_kind="box", mask_format="bits")
    bufs, _m, coef, sig, ve, abt = keep
    bufs["x"].mul_(23.0)
    d.t_ve, d.t_abt, d.t_rsig, d.t_ve_stride, d.t_abt_stride, d.t_rsig_stride = ve.data_ptr(), abt.data_ptr(), sig.data_ptr(), 1, 1, 1
    d.coef_out = coef.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for tune in (_cabi.LP_TUNE_VEC4, _cabi.LP_TUNE_VEC1):
        bufs["x_t"].zero_()
        bufs["x_in"].zero_()
        coef.zero_()
        d.tune = tune
        _cabi.check(lib.lp_step(ctypes.byref(d), st), "lp_step")
        torch.cuda.synchronize()
        res.append([bufs["x_t"].clone(), bufs["x_in"].clone(), coef.clone()])
    for name, a, b in zip(("x_t", "x_in", "coefficient table"), *res):
        assert torch.equal(a, b), name
    assert torch.isfinite(res[0][0]).all() and float(res[0][0].abs().max()) > 1.0
    assert float(coef[0, _cabi.LP_C_RSCALE]) == 1.0 / float(coef[0, _cabi.LP_C_SCALE]) or abs(float(coef[0, _cabi.LP_C_RSCALE]) * float(coef[0, _cabi.LP_C_SCALE]) - 1.0) < 1e-6
