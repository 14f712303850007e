"""Property tests (hypothesis) of the oracle's integer mask math against the torch ops the reference calls,
and of algebraic invariants of the Langevin restatement.  CPU only."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import lanpaint_oracle as orc


RULE_NAME = {0: "scalar", 1: "generic_fma", 2: "generic"}       # include/lanpaint_hip.h LP_NN_ATEN_*


def _cpu_fma():
    """Does this torch build's CPU TensorIterator kernel contract its multiply-subtract on this host?  (2 -> 41, output 20)"""
    probe = torch.nn.functional.interpolate(torch.tensor([[[0.0, 1.0]]]), size=(41,), mode="nearest-exact")
    return float(probe[0, 0, 20]) == 1.0


def _torch_cpu_index(n_in, n_out, nd, axis=0, other_out=3, channels=1, channels_last=False):
    """The source index torch's CPU kernel picks along one axis of an nd-dimensional nearest-exact call."""
    interp = torch.nn.functional.interpolate
    shp, view, size = [1, channels] + [2] * nd, [1, 1] + [1] * nd, [other_out] * nd
    shp[2 + axis], view[2 + axis], size[axis] = n_in, n_in, n_out
    x = torch.arange(n_in, dtype=torch.float32).view(view).expand(shp).contiguous()
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last if nd == 2 else torch.channels_last_3d)
    y = interp(x, size=tuple(size), mode="nearest-exact")
    idx = [0, 0] + [0] * nd
    idx[2 + axis] = slice(None)
    return y[tuple(idx)].numpy().astype(np.int64)


def test_nearest_exact_index_equals_torch_cpu_on_every_pair():
    """torch-CPU `nearest-exact` is what the reference runs (nodes.py:110-127, 159-160: reshape_mask before
    `.to(device)`), so it is the arbiter of "bit-exact mask index math" -- in BOTH directions.  The oracle's three index forms +
    ATen's dispatch (aten_nearest_exact_rule) against torch on the CPU, exhaustively: 1-D every (in <= 256, out <= 512) pair;
    2-D in both regimes of _use_vectorized_kernel_cond_2d (out_h + out_w <= 128: scalar rule; above: TensorIterator kernel);
    3-D on each axis.  0 mismatching pairs (round 4's single fp32 form: 94 mismatching up-sampling pairs up to 512)."""
    fma = _cpu_fma()
    bad = []
    for n_in in range(1, 257):                                          # 1-D: the audio [F] -> tokens path (nodes.py:78-82)
        for n_out in range(1, 513):
            rule = orc.aten_nearest_exact_rule("cpu", 1, (n_out,), cpu_fma=fma)
            if not np.array_equal(orc.nearest_exact_src_index(n_out, n_in, rule), _torch_cpu_index(n_in, n_out, 1)):
                bad.append((1, n_in, n_out))
    assert not bad, bad[:10]
    for n_in in list(range(1, 80)) + [124, 128, 259]:                   # 2-D: image masks; (T, 1) audio masks (nodes.py:88)
        for n_out in range(1, 300):
            for other in (1, 3, 128 - n_out, 129 - n_out, 200):         # both sides of the out_h + out_w = 128 boundary
                if other < 1:
                    continue
                rule = orc.aten_nearest_exact_rule("cpu", 2, (n_out, other), cpu_fma=fma)
                assert rule == ("scalar" if n_out + other <= 128 else ("generic_fma" if fma else "generic"))
                if not np.array_equal(orc.nearest_exact_src_index(n_out, n_in, rule), _torch_cpu_index(n_in, n_out, 2, 0, other)):
                    bad.append((2, n_in, n_out, other))
    assert not bad, bad[:10]
    for axis in range(3):                                               # 3-D: the video path (nodes.py:110-114)
        for n_in in list(range(1, 40)) + [81, 121]:
            for n_out in range(1, 200, 1 if n_in < 16 else 3):
                rule = orc.aten_nearest_exact_rule("cpu", 3, (n_out, 3, 3), cpu_fma=fma)
                if not np.array_equal(orc.nearest_exact_src_index(n_out, n_in, rule), _torch_cpu_index(n_in, n_out, 3, axis)):
                    bad.append((3, axis, n_in, n_out))
    assert not bad, bad[:10]
    # channels-last inputs with more than 3 channels go to the scalar-rule kernels whatever the output size
    for nd, cl_c, want in ((2, 4, "scalar"), (2, 3, None), (3, 8, "scalar"), (3, 1, None)):
        rule = orc.aten_nearest_exact_rule("cpu", nd, (141, 3) if nd == 2 else (41, 3, 3), cl_c, True, fma)
        assert want is None or rule == want
        n_in, n_out = (2, 141) if nd == 2 else (2, 41)
        assert np.array_equal(orc.nearest_exact_src_index(n_out, n_in, rule), _torch_cpu_index(n_in, n_out, nd, 0, 3, cl_c, True))
    assert orc.nearest_exact_src_index(201, 14)[100] == 6               # exact-rational math would say 7
    assert orc.nearest_exact_src_index(41, 2, "scalar")[20] == 0 and orc.nearest_exact_src_index(41, 2, "generic_fma")[20] == 1
    assert orc.nearest_exact_src_index(41, 2, "generic")[20] == 0


def test_the_three_index_rules_agree_wherever_a_mask_is_brought_down_to_a_latent_grid():
    """Every DOWN-sampling pair up to 512 (pixel mask -> latent grid: x8 in space, 81 -> 21 frames): the three forms give
    the same indices, so the production masks were already the reference's on either device; up-sampling they differ on
    94 (scalar vs contracted) of the 130 816 pairs."""
    differ_up = 0
    for n_in in range(1, 513):
        for n_out in range(1, 513):
            a = orc.nearest_exact_src_index(n_out, n_in, "scalar")
            b = orc.nearest_exact_src_index(n_out, n_in, "generic_fma")
            if n_out <= n_in:
                assert np.array_equal(a, b) and np.array_equal(a, orc.nearest_exact_src_index(n_out, n_in, "generic")), (n_in, n_out)
            elif not np.array_equal(a, b):
                differ_up += 1
    assert differ_up == 94


def test_product_rule_dispatch_is_the_oracles_and_torchs():
    """lanpaint_amd.interp_rule (what the HIP launch is told) == the oracle's restatement of ATen's dispatch == torch on the CPU,
    on the tensors the reference's call sites build (host masks; a CUDA mask always takes the GPU kernels' scalar rule)."""
    from lanpaint_amd import interp_rule as ir
    fma = _cpu_fma()
    assert RULE_NAME[ir.cpu_generic_rule()] == ("generic_fma" if fma else "generic")
    host = torch.zeros(3)
    cases = [(torch.zeros(1, 1, 7), (41,)), (torch.zeros(1, 1, 7, 1), (41, 1)), (torch.zeros(1, 1, 7, 1), (200, 1)),
             (torch.zeros(2, 1, 30, 30), (64, 64)), (torch.zeros(2, 1, 30, 30), (128, 128)), (torch.zeros(1, 1, 5, 9, 9), (21, 60, 104)),
             (torch.zeros(1, 8, 6, 6).contiguous(memory_format=torch.channels_last), (100, 100)),
             (torch.zeros(1, 3, 6, 6).contiguous(memory_format=torch.channels_last), (100, 100))]
    for view, size in cases:
        nd = len(size)
        cl = nd in (2, 3) and view.shape[1] > 3 and view.is_contiguous(memory_format=torch.channels_last if nd == 2 else torch.channels_last_3d)
        want = orc.aten_nearest_exact_rule("cpu", nd, size, view.shape[1], cl, fma)
        assert RULE_NAME[ir.rule_for(host, view, size)] == want, (tuple(view.shape), size)
        n_in = view.shape[2]
        src = torch.arange(n_in, dtype=torch.float32).view([1, 1, n_in] + [1] * (nd - 1)).expand(view.shape).contiguous()
        if cl:
            src = src.contiguous(memory_format=torch.channels_last)
        got = torch.nn.functional.interpolate(src, size=size, mode="nearest-exact")
        idx = got[(0, 0, slice(None)) + (0,) * (nd - 1)].numpy().astype(np.int64)
        assert np.array_equal(idx, orc.nearest_exact_src_index(size[0], n_in, want)), (tuple(view.shape), size)
    assert ir.aten_rule(True, 1, (41,)) == 0 and ir.aten_rule(True, 3, (21, 60, 104)) == 0


@settings(max_examples=60, deadline=None, derandomize=True, database=None)
@given(st.integers(1, 40), st.integers(1, 300), st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
def test_reshape_mask_audio_upsampling_equals_torch_cpu_pipeline(f, t, ch, seed):
    """The reference's audio paths -- [F] -> tokens (1-D call) and [1, 1, F, 1] (2-D call, size (T, 1)) -- UP-sample; on a host
    mask the oracle must give what torch's CPU kernels give (nodes.py:74-89, 123-130)."""
    rng = np.random.default_rng(seed)
    m = rng.random(f).astype(np.float32)
    fma = _cpu_fma()
    interp = torch.nn.functional.interpolate
    want = interp(torch.from_numpy(m)[None, None], size=(t,), mode="nearest-exact").expand(1, 1, ch, t)
    want = interp(want, size=(ch, t), mode="nearest-exact")
    assert np.array_equal(orc.reshape_mask(m, (1, 1, ch, t), mask_on="cpu", cpu_fma=fma), want.numpy())
    m4 = torch.from_numpy(m).reshape(1, 1, f, 1)
    want = interp(m4, size=(t, 1), mode="nearest-exact").permute(0, 1, 3, 2).expand(1, 1, ch, t)
    want = interp(want, size=(ch, t), mode="nearest-exact")
    assert np.array_equal(orc.reshape_mask(m4.numpy(), (1, 1, ch, t), mask_on="cpu", cpu_fma=fma), want.numpy())


@settings(max_examples=40, deadline=None, derandomize=True, database=None)
@given(st.integers(1, 20), st.integers(1, 12), st.integers(1, 12), st.integers(1, 9), st.integers(1, 7), st.integers(1, 7),
       st.integers(1, 3), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_reshape_mask_video_equals_torch_pipeline(f, h, w, tf, th, tw, b, c, seed):
    rng = np.random.default_rng(seed)
    m = (rng.random((f, h, w)) > 0.6).astype(np.float32)
    got = orc.reshape_mask(m, (b, c, tf, th, tw), video_inpainting=True)
    t = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None], size=(tf, th, tw), mode="nearest-exact")
    t = torch.nn.functional.max_pool3d(t, kernel_size=(5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0))
    assert np.array_equal(got, t.repeat(b, c, 1, 1, 1).numpy())


@settings(max_examples=40, deadline=None, derandomize=True, database=None)
@given(st.integers(1, 16), st.integers(1, 16), st.integers(1, 9), st.integers(1, 9), st.integers(1, 3), st.integers(1, 4),
       st.integers(0, 2 ** 31 - 1))
def test_reshape_mask_image_equals_torch_pipeline(h, w, th, tw, b, c, seed):
    rng = np.random.default_rng(seed)
    m = (rng.random((h, w)) > 0.5).astype(np.float32)
    got = orc.reshape_mask(m, (b, c, th, tw))
    t = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None], size=(th, tw), mode="nearest-exact")
    assert np.array_equal(got, t.repeat(b, c, 1, 1).numpy())


@settings(max_examples=60, deadline=None, derandomize=True, database=None)
@given(st.integers(1, 12), st.integers(1, 12), st.integers(0, 2 ** 31 - 1))
def test_boundary_ring_properties(h, w, seed):
    """ring pixels are inpaint pixels with a known 4-neighbour; nothing else; known pixels never in the ring."""
    rng = np.random.default_rng(seed)
    m = (rng.random((1, 2, h, w)) > 0.5).astype(np.float32)
    ring = orc.boundary_weight(m, (1 - m).astype(np.float32))
    known = m > 0.5
    pad = np.pad(known, ((0, 0), (0, 0), (1, 1), (1, 1)))
    nb = pad[:, :, :-2, 1:-1] | pad[:, :, 2:, 1:-1] | pad[:, :, 1:-1, :-2] | pad[:, :, 1:-1, 2:]
    assert np.array_equal(ring > 0, (~known) & nb)
    assert not (ring[known] > 0).any()


@settings(max_examples=25, deadline=None, derandomize=True, database=None)
@given(st.floats(0.03, 14.0), st.floats(0.5, 10.0), st.floats(0.2, 1.0), st.booleans(), st.integers(0, 2 ** 31 - 1))
def test_think_iteration_is_affine_in_state_and_noise(sigma, lamb, beta, flow, seed):
    """For a fixed mask / sigma the update is affine in (x_t, score inputs, xi): superposition holds exactly in
    exact arithmetic, to fp32 rounding here (the property the full-size GPU test relies on)."""
    rng = np.random.default_rng(seed)
    if flow:
        sigma = min(sigma / 15.0, 0.97)
    shape = (1, 2, 4, 4)
    s = np.float32([sigma])
    times = orc.times_from_sigma(s, flow)
    mask = (rng.random(shape) > 0.5).astype(np.float32)

    def run(x, y, xi):
        it = iter(xi)
        o = orc.OracleLanPaint(None, 1, 15.0, lamb, beta, 0.2, is_flow=flow, randn=lambda like: next(it))
        o.ndim = 4
        one = np.ones((1, 1, 1, 1), dtype=np.float32)
        step = o._bcast(np.float32(0.2) * (1 - times[1]))
        sc = lambda xt: (0.9 * xt - xt) * (1 - mask) + (-(1 + lamb) * (xt - y) + lamb * (xt - 0.8 * xt)) * mask   # noqa: E731
        x1, st1 = o.think_iteration(x, sc, mask, step, times, one, one * np.float32(beta), None)
        x2, _ = o.think_iteration(x1, sc, mask, step, times, one, one * np.float32(beta), st1)
        return x2

    def rnd():
        return rng.standard_normal(shape).astype(np.float32)

    A = (rnd(), rnd(), [rnd() for _ in range(3)])
    B = (rnd(), rnd(), [rnd() for _ in range(3)])
    a, b = np.float32(0.6), np.float32(-1.3)
    AB = (a * A[0] + b * B[0], a * A[1] + b * B[1], [a * p + b * q for p, q in zip(A[2], B[2])])
    want = a * run(*A) + b * run(*B)
    got = run(*AB)
    assert np.abs(got - want).max() <= 2e-4 * max(1.0, float(np.abs(want).max()))


@settings(max_examples=60, deadline=None, derandomize=True, database=None)
@given(st.integers(1, 700), st.booleans(), st.integers(0, 2**31 - 1))
def test_mask_bit_packing_round_trip(n, denoise, seed):
    """pack -> unpack is the identity on binary masks; word / bit positions follow the header's layout."""
    from oracle.lanpaint_oracle import pack_mask_bits, unpack_mask_bits
    rng = np.random.default_rng(seed)
    m = (rng.random(n) < 0.5).astype(np.float32)
    bits = pack_mask_bits(m, denoise_mask=denoise)
    assert bits.dtype == np.uint8 and bits.size == ((n + 63) // 64) * 8
    want = 1.0 - m if denoise else m
    assert np.array_equal(unpack_mask_bits(bits, n), want)
    words = bits.view("<u4")
    for i in rng.integers(0, n, size=min(n, 16)):
        assert (int(words[i >> 5]) >> (int(i) & 31)) & 1 == int(want[i])
    assert not np.unpackbits(bits, bitorder="little")[n:].any()          # tail bits are zero
