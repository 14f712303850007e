// aux_kernels.hip -- the small kernels around the fused step (gfx950, wave64):
//   K1 lp_coeffs        per-row coefficient table on the device (no host sync)
//   K3 lp_finalize      known-region reprojection + in-place write-back
//      lp_philox_normal standalone N(0,1) fill (same generator as the fused step)
//   K4 lp_boundary_ring / lp_wmse_pair   inner early-stop metric: mask-edge stencil
//                        via wave ballots + an LDS tile of row bitmasks, and a
//                        deterministic weighted-MSE reduction
//   K5 lp_reshape_mask  exact-integer nearest-exact resample + temporal union
#include "lp_common.h"

namespace lp {

// ---------------------------------------------------------------------------------
// K1: FOUR lanes per batch row, one per (region, tau in {dt, dt/2}): the closed-form factors are 12
// dependent double-precision exp / expm1 evaluations per row, and the table sits on the critical path of
// every sigma call, so the chain is cut to 3 per lane.  fp32 fields mirror the reference's own op order
// (prepare_step_size, lanpaint.py:295-328); the per-region factors are evaluated in double from those
// fp32 inputs (lanpaint.py:241-252 in exact arithmetic).
// ---------------------------------------------------------------------------------
__global__ void lp_coeffs_kernel(lp_hyper h, const float* __restrict__ ve, int ve_stride,
                                 const float* __restrict__ abt, int abt_stride, const float* __restrict__ rs,
                                 int rs_stride, const float* __restrict__ step_ov, int step_stride,
                                 const float* __restrict__ t_model, int t_stride, int rows, float* __restrict__ table) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t >> 2;
    if (r >= rows) return;
    const float abt_f = abt[static_cast<int64_t>(r) * abt_stride];
    const float ve_f = ve ? ve[static_cast<int64_t>(r) * ve_stride] : 0.0f;
    const float rs_f = rs ? rs[static_cast<int64_t>(r) * rs_stride] : 0.0f;
    const float tm_f = t_model ? t_model[static_cast<int64_t>(r) * t_stride] : 0.0f;
    const float oma = 1.0f - abt_f;
    const float step = step_ov ? step_ov[static_cast<int64_t>(r) * step_stride]
                               : h.step_size * fmaxf(oma, h.min_step_frac);  // lanpaint.py:81
    coeffs_lane(h, abt_f, ve_f, rs_f, tm_f, step, (t >> 1) & 1, t & 1, table + static_cast<int64_t>(r) * LP_COEF_STRIDE);
}

// ---------------------------------------------------------------------------------
// K1a: per-sigma scalar algebra of KSamplerX0Inpaint.__call__ (nodes.py:242-252, 286-299).
// One thread: B and the schedule are tiny.  __f*_rn keep every op separately rounded like the
// reference's eager fp32 tensor ops (an FMA here could flip a round() in the n_eff rule).
// ---------------------------------------------------------------------------------
// The inner-step rule evaluated on the device too (round 3): with `rule.valid_out` the kernel compares the count the host
// SPECULATED (`rule.guess`; it queued the whole sigma call for it before this kernel ran) with the true one and writes the
// word the captured lp_finalize looks at: 0 voids that run -- nothing of the caller's is written, no generator state moves --
// and the host, which reads the true count from the mailbox, queues the call again.  (Body shared with the LP_PH_SIGMA form
// of the replace launch: lp_common.h.)
__global__ void lp_sigma_times_kernel(const float* __restrict__ sigma, int rows, const float* __restrict__ schedule,
                                      int schedule_len, int is_flow, float* __restrict__ times,
                                      float* __restrict__ scalars, int32_t* __restrict__ seq_out, int32_t seq,
                                      const SigmaRule rule) {
    if (blockIdx.x != 0 || threadIdx.x >= kWave) return;             // one full wave (launched as such)
    sigma_rows_and_rule(sigma, rows, schedule, schedule_len, is_flow != 0, times, scalars, seq_out, seq, rule);
}

int sigma_times_rule_dispatch(const float* sigma, int rows, const float* schedule, int schedule_len, int is_flow,
                              float* times, float* scalars, int32_t* seq_out, int32_t seq, int32_t n_steps, int32_t early_stop,
                              int32_t total_steps, double min_step_frac, int32_t guess, uint64_t* valid_out, hipStream_t stream) {
    if (!sigma || !schedule || !times || !scalars || rows <= 0 || schedule_len <= 0) return LP_E_INVALID;
    const SigmaRule rule{n_steps, early_stop, total_steps, guess, min_step_frac, valid_out};
    hipLaunchKernelGGL(lp_sigma_times_kernel, dim3(1), dim3(64), 0, stream, sigma, rows, schedule, schedule_len, is_flow,
                       times, scalars, seq_out, seq, rule);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

int sigma_times_dispatch(const float* sigma, int rows, const float* schedule, int schedule_len, int is_flow,
                         float* times, float* scalars, int32_t* seq_out, int32_t seq, hipStream_t stream) {
    return sigma_times_rule_dispatch(sigma, rows, schedule, schedule_len, is_flow, times, scalars, seq_out, seq, 0, 0, 0, 0.0, -1,
                                     nullptr, stream);
}

int coeffs_dispatch(const lp_hyper* h, const float* ve, int ve_stride, const float* abt, int abt_stride,
                    const float* rs, int rs_stride, const float* step_ov, int step_stride, const float* t_model,
                    int t_stride, int rows, float* table, hipStream_t stream) {
    if (!h || !abt || !table || rows <= 0) return LP_E_INVALID;
    if (!h->is_flow && !ve) return LP_E_INVALID;
    const int block = 64, lanes = rows * 4;
    hipLaunchKernelGGL(lp_coeffs_kernel, dim3((lanes + block - 1) / block), dim3(block), 0, stream, *h, ve, ve_stride,
                       abt, abt_stride, rs, rs_stride, step_ov, step_stride, t_model, t_stride, rows, table);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// K3: out = model_out*(1-m) + y*m ; x_dst <- x_src      (lanpaint.py:154,156)
// ---------------------------------------------------------------------------------
// The leading arguments are preloaded into SGPRs by the command processor (see lp_step_kernel in step_kernel.hip):
// the four input streams, the I/O table, the length and the flags are there at entry instead of behind a cold read of
// the argument segment (a replayed finalize was argument segment -> table -> stores, three dependent round trips).
template <int VEC, int BLOCK>
__global__ __launch_bounds__(256) void lp_finalize_kernel(const void* a_mask, const void* a_model_out, const float* a_y,
                                                          const float* a_x_src, const uint64_t* a_io_table, int64_t a_n_el,
                                                          uint32_t a_flags, const lp_final_desc d_arg) {
    lp_final_desc d = d_arg;
    d.mask = a_mask; d.model_out = a_model_out; d.y = a_y; d.x_src = a_x_src; d.io_table = a_io_table;
    d.n_el = a_n_el; d.flags = a_flags;
    const int64_t groups = d.n_el / VEC;
    const int dt = x0_dtype(d.flags);
    const int64_t g = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x;          // one group per lane
    // a finalize replayed from a hipGraph takes the caller's two tensors of THIS call from the table the replace
    // launch published (scalar loads, issued first; only the stores at the end of the body wait for them)
    float* const x_dst = d.io_table ? reinterpret_cast<float*>(d.io_table[0]) : d.x_dst;
    float* const out = d.io_table ? reinterpret_cast<float*>(d.io_table[1]) : d.out;
    // word 2 of the table: 0 voids this launch (a sigma call queued for a speculated inner-step count that turned out
    // wrong, lp_node_call): nothing of the caller's is written and the replayed generator counter does not move
    if (d.io_table && d.io_table[2] == 0ull) return;
    // Region-aware streams at streaming sizes (bit-packed mask, 16 B per lane; same rule as lp_step_kernel): a wave whose 256 mask
    // bits are all 0 (inpaint) never reads y, one whose bits are all 1 (known) never reads the model output -- 20.1 -> 16.1 B per
    // element in those waves.  The decision waits for the mask word, so the latency-bound sizes (BLOCK = 64) keep issuing every load
    // at once.  (m in {0, 1}: out = model_out * 1 + 0 * 0 resp. 0 * 0 + y * 1, the value of the full expression for finite operands.)
    constexpr bool RA = VEC == 4 && BLOCK == 256;
    const bool in_range = g < groups;
    if (RA || in_range) {
        const int64_t i = (in_range ? g : groups - 1) * VEC;
        float m[VEC], mo[VEC], yv[VEC], o[VEC];
        Raw<VEC> m_raw, mo_raw, un_raw;                 // issue every load, decode afterwards (lp_common.h)
        load_mask_raw<VEC>(d.mask, d.flags, i, m_raw);
        float xs[VEC];
        bool need_mo = true, need_y = true;
        if constexpr (RA) {
            if (x_dst) load_f32<VEC>(d.x_src, i, xs);
            if ((d.flags & LP_FL_MASK_BITS) && !(d.flags & (LP_FL_CFG_FUSED | LP_FL_NO_REGION_SKIP))) {
                const uint32_t nib = (m_raw.w[0] >> (static_cast<uint32_t>(i) & 31u)) & 0xFu;
                need_y = __ballot(in_range && nib != 0u) != 0ull;
                need_mo = __ballot(in_range && nib != 0xFu) != 0ull;
            }
            if (!in_range) return;                      // (after the ballots: every lane of the wave takes part in them)
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) mo_raw.w[k] = 0u, yv[k] = 0.0f;
        if (need_mo) load_raw<VEC>(d.model_out, dt, i, mo_raw);
        if (d.flags & LP_FL_CFG_FUSED) load_raw<VEC>(d.uncond, dt, i, un_raw);
        if (need_y) load_f32<VEC>(d.y, i, yv);
        if constexpr (!RA) {
            if (x_dst) load_f32<VEC>(d.x_src, i, xs);
        }
        cvt_mask<VEC>(d.flags, i, m_raw, m);
        cvt_raw<VEC>(dt, mo_raw, mo);
        if (d.flags & LP_FL_CFG_FUSED) {
            float un[VEC];
            cvt_raw<VEC>(dt, un_raw, un);
#pragma unroll
            for (int k = 0; k < VEC; ++k) mo[k] = un[k] + (mo[k] - un[k]) * d.cfg_scale;
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = mo[k] * (1.0f - m[k]) + yv[k] * m[k];
        store_f32<VEC>(out, i, o);
        if (x_dst) store_f32<VEC>(x_dst, i, xs);
    }
    if (d.rng_bump_ptr && blockIdx.x == 0 && threadIdx.x == 0) *d.rng_bump_ptr += d.rng_bump;
}

static bool aligned(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int finalize_dispatch(const lp_final_desc* dp, hipStream_t stream) {
    if (!dp) return LP_E_INVALID;
    const lp_final_desc& d = *dp;
    if (d.n_el <= 0 || !d.model_out || !d.y || !d.mask || (!d.out && !d.io_table)) return LP_E_INVALID;
    if ((d.x_dst || d.io_table) && !d.x_src) return LP_E_INVALID;
    if ((d.flags & LP_FL_CFG_FUSED) && !d.uncond) return LP_E_INVALID;
    if ((d.flags & LP_FL_MASK_BITS) && ((d.flags & (LP_FL_MASK_U8 | LP_FL_MASK_DENOISE)) || !aligned(d.mask, 4)))
        return LP_E_INVALID;
    const bool half = x0_dtype(d.flags) != DT_F32;
    const bool vec4 = (d.n_el % 4 == 0) && aligned(d.model_out, half ? 8 : 16) && aligned(d.uncond, half ? 8 : 16) &&
                      aligned(d.y, 16) &&
                      aligned(d.mask, (d.flags & (LP_FL_MASK_U8 | LP_FL_MASK_BITS)) ? 4 : 16) && aligned(d.x_src, 16) &&
                      (d.io_table ? true : (aligned(d.x_dst, 16) && aligned(d.out, 16)));
    const int vec = vec4 ? 4 : 1;
    const int64_t groups = d.n_el / vec;
    const int block = groups <= 64 * 1024 ? 64 : 256;
    const int64_t bx = (groups + block - 1) / block;
    if (bx > 0x7fffffff) return LP_E_INVALID;
#define LP_FINAL_ARGS d.mask, d.model_out, d.y, d.x_src, d.io_table, d.n_el, d.flags, d
    const dim3 grid(static_cast<unsigned>(bx));
    if (vec4 && block == 64) hipLaunchKernelGGL((lp_finalize_kernel<4, 64>), grid, dim3(64), 0, stream, LP_FINAL_ARGS);
    else if (vec4) hipLaunchKernelGGL((lp_finalize_kernel<4, 256>), grid, dim3(256), 0, stream, LP_FINAL_ARGS);
    else if (block == 64) hipLaunchKernelGGL((lp_finalize_kernel<1, 64>), grid, dim3(64), 0, stream, LP_FINAL_ARGS);
    else hipLaunchKernelGGL((lp_finalize_kernel<1, 256>), grid, dim3(256), 0, stream, LP_FINAL_ARGS);
#undef LP_FINAL_ARGS
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// standalone fill with the fused kernel's generator: element e of launch `offset` gets the
// cosine (slot 0, POST stream) or sine (slot 1, PRE stream) branch of its Box-Muller pair
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lp_philox_kernel(float* __restrict__ out, int64_t n_el, uint64_t seed,
                                                        uint64_t offset, uint32_t slot) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_el;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float za, zb;
        normal_pair(static_cast<uint64_t>(e), offset, seed, za, zb);
        out[e] = slot ? zb : za;
    }
}

int philox_dispatch(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t slot, hipStream_t stream) {
    if (!out || n_el <= 0) return LP_E_INVALID;
    if (slot > 1) return LP_E_INVALID;
    int64_t bx = (n_el + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(lp_philox_kernel, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, out, n_el, seed,
                       offset, slot);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// K4a: mask-edge ring (earlystop.py:32-49).  A block owns a 32-row x 62-column tile
// of one H x W plane.  Each wave turns a 64-pixel row segment (tile + 1-pixel halo)
// into ONE 64-bit word with __ballot(known); the 34 row words of the tile (+halo)
// are staged in LDS; the 4-neighbour stencil then runs on whole words:
//   ring = ~k & ((k << 1) | (k >> 1) | up | down)
// ring value = (1 - mask) on ring pixels (boundary.float() * inpaint_weight), else 0.
// ---------------------------------------------------------------------------------
constexpr int kRingRows = 32, kRingCols = 62;

__global__ __launch_bounds__(256) void lp_ring_kernel(const float* __restrict__ mask, float* __restrict__ ring,
                                                      int height, int width) {
    __shared__ unsigned long long rowbits[kRingRows + 2];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x0 = blockIdx.x * kRingCols - 1, y0 = blockIdx.y * kRingRows - 1;
    const int64_t plane = static_cast<int64_t>(blockIdx.z) * height * width;
    const int x = x0 + lane;
    const bool x_in = x >= 0 && x < width;

    for (int j = wave; j < kRingRows + 2; j += 4) {
        const int y = y0 + j;
        bool known = false;
        if (x_in && y >= 0 && y < height) known = mask[plane + static_cast<int64_t>(y) * width + x] > 0.5f;
        const unsigned long long bits = __ballot(known);
        if (lane == 0) rowbits[j] = bits;
    }
    __syncthreads();
    for (int j = 1 + wave; j <= kRingRows; j += 4) {
        const int y = y0 + j;
        if (y >= height) break;
        const unsigned long long k = rowbits[j];
        const unsigned long long nb = (k << 1) | (k >> 1) | rowbits[j - 1] | rowbits[j + 1];
        const unsigned long long rb = ~k & nb;
        if (lane >= 1 && lane <= kRingCols && x_in) {
            const int64_t idx = plane + static_cast<int64_t>(y) * width + x;
            ring[idx] = ((rb >> lane) & 1ull) ? (1.0f - mask[idx]) : 0.0f;
        }
    }
}

int ring_dispatch(const float* mask, float* ring, int64_t planes, int height, int width, hipStream_t stream) {
    if (!mask || !ring || planes <= 0 || height <= 0 || width <= 0) return LP_E_INVALID;
    if (planes > 65535) return LP_E_UNSUPPORTED;
    dim3 grid((width + kRingCols - 1) / kRingCols, (height + kRingRows - 1) / kRingRows, static_cast<unsigned>(planes));
    if (grid.y > 65535) return LP_E_UNSUPPORTED;
    hipLaunchKernelGGL(lp_ring_kernel, grid, dim3(256), 0, stream, mask, ring, height, width);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// K4b: { sum(w1 d^2), sum(w1), sum(w2 d^2), sum(w2) }, d = a-b, w1 = 1-mask, w2 = ring.
// fp32 per-thread partials, double from the wave reduction upward, fixed order.
// ---------------------------------------------------------------------------------
// V = 4: sixteen bytes per lane and stream (rows of 4 k elements, 16-byte aligned tensors -- every latent); V = 1: the rest.
// Host-stopper configurations (custom distance_fn, a batch sharded over ranks) run this once per think iteration.
template <int V>
__global__ __launch_bounds__(256) void lp_wmse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ mask,
                                                              const float* __restrict__ ring, int64_t n_el,
                                                              double* __restrict__ scratch) {
    __shared__ double part[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t groups = n_el / V;
    for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < groups;
         g += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float av[V], bv[V], mv[V], rv[V];
        load_f32<V>(a, g * V, av);
        load_f32<V>(b, g * V, bv);
        load_f32<V>(mask, g * V, mv);
        if (ring) load_f32<V>(ring, g * V, rv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float dv = av[k] - bv[k];
            const float d2 = dv * dv;
            const float w1 = 1.0f - mv[k];
            s[0] += d2 * w1;
            s[1] += w1;
            if (ring) {
                s[2] += d2 * rv[k];
                s[3] += rv[k];
            }
        }
    }
    double v[4] = {static_cast<double>(s[0]), static_cast<double>(s[1]), static_cast<double>(s[2]), static_cast<double>(s[3])};
    wave_sum_dpp(v);                                            // fixed order; the wave's sums in its last lane
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == kWave - 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) part[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        scratch[static_cast<int64_t>(blockIdx.x) * 4 + k] = part[0][k] + part[1][k] + part[2][k] + part[3][k];
    }
}

// One wave: lane l adds the block sums l, l + 64, ... in that order, then the fixed DPP tree over the lanes (round 3 had four
// threads walk up to 1 024 block sums one after the other).
__global__ __launch_bounds__(64) void lp_wmse_final_kernel(const double* __restrict__ scratch, int blocks, double* __restrict__ acc) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int bidx = threadIdx.x; bidx < blocks; bidx += kWave) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += scratch[static_cast<int64_t>(bidx) * 4 + k];
    }
    wave_sum_dpp(v);
    if (threadIdx.x == kWave - 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = v[k];
    }
}

int wmse_dispatch(const float* a, const float* b, const float* mask, const float* ring, int64_t n_el, double* acc,
                  double* scratch, int scratch_blocks, hipStream_t stream) {
    if (!a || !b || !mask || !acc || !scratch || n_el <= 0 || scratch_blocks <= 0) return LP_E_INVALID;
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec4 = n_el % 4 == 0 && al16(a) && al16(b) && al16(mask) && (!ring || al16(ring));
    int64_t bx = (n_el / (vec4 ? 4 : 1) + 255) / 256;
    if (bx > scratch_blocks) bx = scratch_blocks;
    if (bx > 1024) bx = 1024;
    if (vec4)
        hipLaunchKernelGGL(lp_wmse_partial_kernel<4>, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, a, b, mask, ring, n_el,
                           scratch);
    else
        hipLaunchKernelGGL(lp_wmse_partial_kernel<1>, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, a, b, mask, ring, n_el,
                           scratch);
    if (hipGetLastError() != hipSuccess) return LP_E_LAUNCH;
    hipLaunchKernelGGL(lp_wmse_final_kernel, dim3(1), dim3(kWave), 0, stream, scratch, static_cast<int>(bx), acc);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// K5: reshape_mask (nodes.py:59-133) as an output-indexed gather.  The source index is ATen's nearest-exact rule of the
// kernel the reference's device would run (lp_common.h::nearest_exact_index: three fp32 forms; `flags` bits 8..9 choose) --
// the fp32 rounding IS the reference behaviour (the exact-rational index differs, e.g. in=14,out=201,i=100), so each form
// is reproduced op for op.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lp_reshape_mask_kernel(const float* __restrict__ src, int sb, int sc, int sf,
                                                              int sh, int sw, float* __restrict__ dst, int db, int dc,
                                                              int df, int dh, int dw, int taps, int flags) {
    const int binarize = flags & LP_RESHAPE_BINARIZE;
    const int rule = (flags >> LP_RESHAPE_RULE_SHIFT) & 3;
    const int64_t total = static_cast<int64_t>(db) * dc * df * dh * dw;
    const int half = taps / 2;
    for (int64_t o = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; o < total;
         o += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int64_t t = o;
        const int w = static_cast<int>(t % dw); t /= dw;
        const int h = static_cast<int>(t % dh); t /= dh;
        const int f = static_cast<int>(t % df); t /= df;
        const int c = static_cast<int>(t % dc); t /= dc;
        const int b = static_cast<int>(t);
        const int ws = nearest_exact_index(w, sw, dw, rule), hs = nearest_exact_index(h, sh, dh, rule);
        const int64_t plane = (static_cast<int64_t>(b % sb) * sc + (c % sc)) * sf;
        float v = -INFINITY;
        for (int k = -half; k <= half; ++k) {           // max_pool3d (5,1,1), pad (2,0,0) with -inf
            const int ff = f + k;
            if (ff < 0 || ff >= df) continue;
            const int fs = nearest_exact_index(ff, sf, df, rule);
            const float s = src[((plane + fs) * sh + hs) * sw + ws];
            v = (s > v || s != s) ? s : v;              // NaN propagates like torch's max_pool
        }
        if (binarize) v = 1.0f - ((v > 0.5f) ? 1.0f : 0.0f);
        dst[o] = v;
    }
}

int reshape_mask_dispatch(const float* src, int sb, int sc, int sf, int sh, int sw, float* dst, int db, int dc, int df,
                          int dh, int dw, int taps, int flags, hipStream_t stream) {
    if (!src || !dst || sb <= 0 || sc <= 0 || sf <= 0 || sh <= 0 || sw <= 0 || db <= 0 || dc <= 0 || df <= 0 ||
        dh <= 0 || dw <= 0)
        return LP_E_INVALID;
    if (taps < 1 || (taps % 2) == 0) return LP_E_INVALID;
    if (flags < 0 || (flags & ~(LP_RESHAPE_BINARIZE | (3 << LP_RESHAPE_RULE_SHIFT))) != 0 ||
        ((flags >> LP_RESHAPE_RULE_SHIFT) & 3) > LP_NN_ATEN_CPU_GENERIC)
        return LP_E_INVALID;
    const int64_t total = static_cast<int64_t>(db) * dc * df * dh * dw;
    int64_t bx = (total + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(lp_reshape_mask_kernel, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, src, sb, sc, sf,
                       sh, sw, dst, db, dc, df, dh, dw, taps, flags);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// bit-packed mask: one wave64 ballot turns 64 consecutive elements into one 64-bit word
// ---------------------------------------------------------------------------------
// (`mask` and `latent_out` carry no __restrict__: the header allows them to be ONE buffer when flags == 0 -- an in-place
// refresh of a mask that is its own fp32 form; every element is read and written by the same lane)
__global__ __launch_bounds__(256) void lp_pack_mask_kernel(const float* mask, int64_t n_el, uint32_t flags,
                                                           unsigned long long* __restrict__ bits,
                                                           int32_t* __restrict__ nonbinary, float* latent_out) {
    const int64_t words = (n_el + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    bool soft = false;
    for (int64_t w = wave; w < words; w += n_waves) {
        const int64_t i = w * 64 + lane;
        const float v = (i < n_el) ? mask[i] : ((flags & LP_FL_MASK_DENOISE) ? 1.0f : 0.0f);
        soft |= !(v == 0.0f || v == 1.0f);
        const bool hi = v > 0.5f;
        const bool known = (flags & LP_FL_MASK_DENOISE) ? !hi : hi;
        const unsigned long long word = __ballot(known);
        if (lane == 0) bits[w] = word;
        if (latent_out && i < n_el) latent_out[i] = known ? 1.0f : 0.0f;       // the fp32 latent_mask of nodes.py:281-283
    }
    if (nonbinary && !(flags & LP_FL_MASK_DENOISE) && __ballot(soft) != 0ull && lane == 0) *nonbinary = 1;
}

int pack_mask_dispatch(const float* mask, int64_t n_el, uint32_t flags, void* bits, int32_t* nonbinary, float* latent_out,
                       hipStream_t stream) {
    if (!mask || !bits || n_el <= 0 || (flags & ~LP_FL_MASK_DENOISE) || !aligned(bits, 8)) return LP_E_INVALID;
    const int64_t words = (n_el + 63) / 64;
    int64_t bx = (words + 3) / 4;                       // 4 waves per block
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(lp_pack_mask_kernel, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, mask, n_el, flags,
                       static_cast<unsigned long long*>(bits), nonbinary, latent_out);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// ---------------------------------------------------------------------------------
// LP_RNG_TORCH test hook: the whole torch.randn(n) tensor from the per-element function the step kernel uses
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lp_torch_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                              uint64_t offset, uint32_t bg) {
    const int64_t li = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (li < n) out[li] = torch_normal(static_cast<uint64_t>(li), seed, offset, bg, n <= static_cast<int64_t>(bg));
}

int torch_normal_dispatch(float* out, int64_t n, uint64_t seed, uint64_t offset, uint32_t bg, hipStream_t stream) {
    if (!out || n <= 0 || bg == 0 || (offset & 3ull)) return LP_E_INVALID;
    const int64_t bx = (n + 255) / 256;
    if (bx > 0x7fffffff) return LP_E_INVALID;
    hipLaunchKernelGGL(lp_torch_normal_kernel, dim3(static_cast<unsigned>(bx)), dim3(256), 0, stream, out, n, seed, offset,
                       bg);
    return hipGetLastError() == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

}  // namespace lp
