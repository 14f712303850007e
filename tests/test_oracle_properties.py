"""Property tests (hypothesis) of the oracle's integer mask math against the torch ops the reference calls,
and of algebraic invariants of the Langevin restatement.  CPU only."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import lanpaint_oracle as orc


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 400), st.integers(1, 400))
def test_nearest_exact_index_equals_torch(n_in, n_out):
    """The oracle restates ATen's formula (fp32 scale, fp32 product), which torch's GPU kernels follow exactly
    (checked on the MI355X for 8288 (in,out) pairs, all three ranks).  torch's CPU kernels deviate from their
    own formula at a few indices where (i + 0.5) * scale lands one ulp below an integer k (they return k):
    39 of 8288 pairs on the 1-D/3-D paths, 26 on the 2-D path, e.g. (in=2, out=47, i=23).  Anything else is a bug."""
    src = torch.arange(n_in, dtype=torch.float32).reshape(1, 1, n_in)
    want = torch.nn.functional.interpolate(src, size=(n_out,), mode="nearest-exact").reshape(-1).numpy().astype(np.int64)
    got = orc.nearest_exact_src_index(n_out, n_in)
    scale = np.float32(n_in) / np.float32(n_out)
    for i in np.nonzero(got != want)[0]:
        p = (np.float32(i) + np.float32(0.5)) * scale
        k = np.float32(np.round(p))
        assert p == np.nextafter(k, np.float32(-np.inf)) and want[i] == min(int(k), n_in - 1), (n_in, n_out, int(i))


def test_nearest_exact_known_cpu_kernel_deviation_is_the_only_one():
    """Exhaustive over a grid that contains the known deviating pairs."""
    dev = 0
    for n_in in (1, 2, 3, 4, 6, 7, 14, 54):
        for n_out in range(1, 260):
            src = torch.arange(n_in, dtype=torch.float32).reshape(1, 1, 1, n_in)
            want = torch.nn.functional.interpolate(src, size=(1, n_out), mode="nearest-exact").reshape(-1).numpy().astype(np.int64)
            got = orc.nearest_exact_src_index(n_out, n_in)
            bad = np.nonzero(got != want)[0]
            dev += len(bad)
            scale = np.float32(n_in) / np.float32(n_out)
            for i in bad:
                p = (np.float32(i) + np.float32(0.5)) * scale
                assert p == np.nextafter(np.float32(np.round(p)), np.float32(-np.inf))
    assert dev < 40
    assert orc.nearest_exact_src_index(201, 14)[100] == 6        # exact-rational math would say 7


def test_nearest_exact_cpu_kernels_agree_wherever_a_mask_is_brought_to_a_latent_grid():
    """Which device's rule is "the reference"?  reshape_mask (nodes.py:59-133) runs torch's interpolate on whatever
    device ComfyUI holds the mask on.  The formula restated here (ATen's nearest_exact_idx, followed exactly by
    torch's GPU kernels and by lp_reshape_mask) and torch's CPU 2-D / 3-D kernels -- the ones a CPU-resident mask
    goes through -- can only differ when UPSAMPLING from an even size <= 14 (one index per pair, where the exact
    position is an integer).  A pixel mask is always DOWNSAMPLED to the latent grid (x8 in space, 81 -> 21 frames):
    there the two agree on every pair, so the GPU path gives the masks the reference gives on either device."""
    interp = torch.nn.functional.interpolate

    def cpu2d(a, b):
        src = torch.arange(a, dtype=torch.float32).reshape(1, 1, a, 1).expand(1, 1, a, 3).contiguous()
        return interp(src, size=(b, 3), mode="nearest-exact")[0, 0, :, 0].numpy().astype(np.int64)

    def cpu3d(a, b):
        src = torch.arange(a, dtype=torch.float32).reshape(1, 1, a, 1, 1).expand(1, 1, a, 2, 2).contiguous()
        return interp(src, size=(b, 2, 2), mode="nearest-exact")[0, 0, :, 0, 0].numpy().astype(np.int64)

    # every downsampling pair up to 96, and the pixel sizes of real workflows down to any latent size
    pairs = [(a, b) for a in range(1, 97) for b in range(1, a + 1)]
    pairs += [(a, b) for a in (81, 121, 124, 480, 512, 720, 832, 864, 1024, 2048) for b in range(1, a // 4 + 1, 3)]
    for a, b in pairs:
        want = orc.nearest_exact_src_index(b, a)
        assert np.array_equal(cpu2d(a, b), want) and np.array_equal(cpu3d(a, b), want), (a, b)
    # upsampling: the deviation exists, and only from small even sizes
    dev = set()
    for a in range(1, 40):
        for b in range(a + 1, 200):
            want = orc.nearest_exact_src_index(b, a)
            for got in (cpu2d(a, b), cpu3d(a, b)):
                bad = np.nonzero(got != want)[0]
                assert len(bad) <= 1
                if len(bad):
                    dev.add(a)
                    assert (int(bad[0]) + 0.5) * a / b == float(got[bad[0]])    # torch-CPU returns the exact integer position
    assert dev and dev <= {2, 4, 6, 8, 10, 12, 14}


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 20), st.integers(1, 12), st.integers(1, 12), st.integers(1, 9), st.integers(1, 7), st.integers(1, 7),
       st.integers(1, 3), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_reshape_mask_video_equals_torch_pipeline(f, h, w, tf, th, tw, b, c, seed):
    rng = np.random.default_rng(seed)
    m = (rng.random((f, h, w)) > 0.6).astype(np.float32)
    got = orc.reshape_mask(m, (b, c, tf, th, tw), video_inpainting=True)
    t = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None], size=(tf, th, tw), mode="nearest-exact")
    t = torch.nn.functional.max_pool3d(t, kernel_size=(5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0))
    assert np.array_equal(got, t.repeat(b, c, 1, 1, 1).numpy())


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 16), st.integers(1, 16), st.integers(1, 9), st.integers(1, 9), st.integers(1, 3), st.integers(1, 4),
       st.integers(0, 2 ** 31 - 1))
def test_reshape_mask_image_equals_torch_pipeline(h, w, th, tw, b, c, seed):
    rng = np.random.default_rng(seed)
    m = (rng.random((h, w)) > 0.5).astype(np.float32)
    got = orc.reshape_mask(m, (b, c, th, tw))
    t = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None], size=(th, tw), mode="nearest-exact")
    assert np.array_equal(got, t.repeat(b, c, 1, 1).numpy())


@settings(max_examples=60, deadline=None)
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


@settings(max_examples=25, deadline=None)
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


@settings(max_examples=60, deadline=None)
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
