// lp_common.h -- device helpers shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// LP_RNG_TORCH restates what torch's own kernels inline for torch.randn (ATen's Philox mapping over rocRAND's normal
// transform).  The library is built with -ffp-contract=on (lanpaint_amd/build.py): torch's build contracts a*b+c only
// inside one expression, and under hipcc's default "fast" the Box-Muller results differ in the last bit in ~15 % of the
// draws -- measured on the MI355X against torch.randn, 0 mismatches with "on".
#include "lanpaint_hip.h"

namespace lp {

constexpr int kWave = 64;   // CDNA wavefront width; hard-coded per the gfx950 programming guide

// ---- nearest-exact source index (nodes.py:78,88,110,125-127; 1079, 1278-1287: every F.interpolate(mode="nearest-exact")) ----
// "bit-exact for mask index math" needs the rule of the kernel torch runs ON THE DEVICE THE REFERENCE HOLDS THE MASK ON, and ATen
// has more than one (measured against torch 2.10 over every (in, out) <= 512, tests/test_oracle_properties.py):
//   LP_NN_ATEN_SCALAR           min(int(floorf((i + 0.5f) * scale)), in - 1), all fp32 -- nearest_neighbor_exact_compute_source_index:
//                               torch's GPU kernels; on the CPU the 2-D kernel with out_h + out_w <= 128 and the
//                               channels-last kernels with more than 3 channels (cpu_upsample_nearest*);
//   LP_NN_ATEN_CPU_GENERIC_FMA  the CPU's TensorIterator kernel (upsample_generic_Nd_kernel_impl / HelperInterpNearestExact: 1-D,
//                               3-D, 2-D with out_h + out_w > 128): src = max(scale * (i + 0.5f) - 0.5f, 0) with the multiply-subtract
//                               CONTRACTED to one fma (the AVX2 / AVX512 builds of that file), then floorf(float(double(src) + 0.5));
//   LP_NN_ATEN_CPU_GENERIC      the same kernel as a CPU without FMA runs it (ATEN_CPU_CAPABILITY=default): product and
//                               subtraction rounded separately.
// scale = float(in) / float(out) in all three.  They agree on every DOWN-sampling pair (a pixel mask brought to a latent
// grid); up-sampling they differ on ties -- e.g. 2 -> 41 at i = 20: the scalar rule reads source 0, the generic one source 1.
__device__ __forceinline__ int nearest_exact_index(int i, int in_size, int out_size, int rule) {
#pragma clang fp contract(off)
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    const float at = static_cast<float>(i) + 0.5f;
    int s;
    if (rule == LP_NN_ATEN_SCALAR) {
        s = static_cast<int>(floorf(at * scale));
    } else {
        float src = (rule == LP_NN_ATEN_CPU_GENERIC_FMA) ? __builtin_fmaf(scale, at, -0.5f) : (scale * at - 0.5f);
        src = src < 0.0f ? 0.0f : src;
        s = static_cast<int>(floorf(static_cast<float>(static_cast<double>(src) + 0.5)));
    }
    return s < in_size - 1 ? s : in_size - 1;
}

// ---- Philox2x32-10 (Salmon et al. 2011, Random123) ---------------------------------
// Counter-based, ONE block per latent element: counter = (element index, launch sequence
// number), key = seed folded with the high words.  The two 32-bit outputs make one
// Box-Muller pair whose cosine branch feeds the POST half-step and whose sine branch feeds
// the PRE half-step of the same launch.  No state is carried and the mapping is per
// element, so the stream is independent of grid shape, vector width and block size.
// WIDE: the round's 64-bit product as ONE v_mad_u64_u32 instead of v_mul_hi_u32 + v_mul_lo_u32 -- ten multiply instructions
// less per element of the VALU-bound streaming kernels (round 4).  The instruction also writes a carry into an SGPR pair, so
// the one-element-per-lane kernels (latency-bound, and at their SGPR limit in the run-time-phase forms) keep the two-multiply
// form; the values are the same either way.
template <bool WIDE = false>
__device__ __forceinline__ void philox2x32_10(uint32_t& c0, uint32_t& c1, uint32_t key) {
    constexpr uint32_t M = 0xD256D193u, W = 0x9E3779B9u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi, lo;
        if constexpr (WIDE) {
            const uint64_t p = static_cast<uint64_t>(M) * c0;
            hi = static_cast<uint32_t>(p >> 32); lo = static_cast<uint32_t>(p);
        } else {
            hi = __umulhi(M, c0); lo = M * c0;
        }
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += W;
    }
}

__host__ __device__ __forceinline__ uint32_t philox_key(uint64_t seed, uint64_t seq, uint64_t elem) {
    return static_cast<uint32_t>(seed) ^ (static_cast<uint32_t>(seed >> 32) * 0x85EBCA6Bu) ^
           (static_cast<uint32_t>(seq >> 32) * 0xC2B2AE35u) ^ (static_cast<uint32_t>(elem >> 32) * 0x27D4EB2Fu);
}

// u32 -> uniform in (0, 1]: (x + 0.5) * 2^-32 (never 0, so log is finite).
__device__ __forceinline__ float u01(uint32_t x) {
    return fmaf(static_cast<float>(x), 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// Two independent standard normals for element `elem` of launch `seq`.  v_sin_f32 /
// v_cos_f32 take their argument in revolutions, which is exactly 2*pi*u.
// (normal_pair_key: the same pair with the block's key given -- for element indices < 2^32 the key is philox_key(seed, seq, 0),
// wave-uniform, and the compiler keeps the round keys in SGPRs)
template <bool WIDE>
__device__ __forceinline__ void normal_pair_key(uint32_t c0, uint32_t c1, uint32_t key, float& z_post, float& z_pre) {
    philox2x32_10<WIDE>(c0, c1, key);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c0)));
    const float t = u01(c1);
    z_post = r * __builtin_amdgcn_cosf(t);
    z_pre = r * __builtin_amdgcn_sinf(t);
}

// The same pairs for the FOUR elements of a streaming lane, the float arithmetic around the transcendentals written on
// 2-vectors so that it issues as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per instruction, the same
// values as normal_pair_key element by element -- u01's fma, the -2 ln 2 scaling, r * cos, r * sin: 20 VALU instructions
// become 10 per lane of a launch that is VALU-bound at streaming sizes; round 5).  v_cvt / v_log / v_sqrt / v_sin / v_cos
// have no packed form.
template <bool WIDE>
__device__ __forceinline__ void normal_pairs4_key(const uint32_t (&elem)[4], uint32_t seq, uint32_t key, float (&z_post)[4],
                                                  float (&z_pre)[4]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint32_t c0[4], c1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c0[k] = elem[k]; c1[k] = seq;
        philox2x32_10<WIDE>(c0[k], c1[k], key);
    }
    const f32x2 scale = {2.3283064365386963e-10f, 2.3283064365386963e-10f}, half = {1.1641532182693481e-10f, 1.1641532182693481e-10f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 u = {static_cast<float>(c0[2 * p]), static_cast<float>(c0[2 * p + 1])};
        f32x2 t = {static_cast<float>(c1[2 * p]), static_cast<float>(c1[2 * p + 1])};
        u = __builtin_elementwise_fma(u, scale, half);                     // u01, both elements
        t = __builtin_elementwise_fma(t, scale, half);
        f32x2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
        lg = lg * -1.3862943611198906f;
        const f32x2 r = {__builtin_amdgcn_sqrtf(lg.x), __builtin_amdgcn_sqrtf(lg.y)};
        const f32x2 cs = {__builtin_amdgcn_cosf(t.x), __builtin_amdgcn_cosf(t.y)};
        const f32x2 sn = {__builtin_amdgcn_sinf(t.x), __builtin_amdgcn_sinf(t.y)};
        const f32x2 zc = r * cs, zs = r * sn;
        z_post[2 * p] = zc.x; z_post[2 * p + 1] = zc.y;
        z_pre[2 * p] = zs.x; z_pre[2 * p + 1] = zs.y;
    }
}

__device__ __forceinline__ void normal_pair(uint64_t elem, uint64_t seq, uint64_t seed, float& z_post, float& z_pre) {
    uint32_t c0 = static_cast<uint32_t>(elem), c1 = static_cast<uint32_t>(seq);
    philox2x32_10(c0, c1, philox_key(seed, seq, elem));
    // r = sqrt(-2 ln u) with u in (2^-33, 1]: v_log_f32 (log2) and v_sqrt_f32 directly -- the argument is never
    // denormal, so the library sqrtf's rescaling / refinement sequence (~12 instructions) buys nothing here
    // (c5_wan steady launch 13.2 -> 12.8 us)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c0)));
    const float t = u01(c1);
    z_post = r * __builtin_amdgcn_cosf(t);
    z_pre = r * __builtin_amdgcn_sinf(t);
}

// ---- LP_RNG_TORCH: the value torch.randn_like(t).view(-1)[li] takes for generator state (seed, offset) ------
// ATen (native/cuda/DistributionTemplates.h, distribution_elementwise_grid_stride_kernel with unroll 4): thread
// idx = li % bg initialises Philox4x32-10 with (seed, subsequence = idx, offset), its k-th normal4 call serves
// the elements li = idx + bg * (4 k + ii), ii = 0..3, and normal_() stores rand * std + mean with std 1, mean 0.
// Philox4x32-10 as in Random123 / rocRAND's philox4x32_10_engine (same constants, same output order): the counter
// is (offset / 4 [64 bit], subsequence [64 bit]), the key the seed.  Written out so that a launch evaluates exactly
// one block per call (rocrand_init + rocrand4 carry state bookkeeping the compiler does not always drop).
// rocRAND's Box-Muller (rocrand_normal.h, box_muller(unsigned, unsigned)) restated expression for expression -- the
// same constants, the same fused a + x*a shapes, library logf / sqrtf, the native sine / cosine of __sincosf: the
// library no longer reaches into rocrand_device::detail.  (sine first: .x of the pair is the sine branch.)
// The library logf / correctly rounded sqrtf of that expression, restated for the ARGUMENTS Box-Muller can hand them (round 5):
// what the ROCm device library computes for a normal-range argument, minus its range handling.  u = 2^-32 (x + 1) lies in
// [2^-32, 1]: never denormal, its logarithm finite -- ocml's logf is v_log_f32 (log2) times ln 2 in two pieces (head
// 0x3f317217, tail 0x3377d1cf, the product's rounding error recovered with an fma); the x < 2^-126 rescaling by 2^32 and the
// |log2| < inf select around it never act.  -2 ln u lies in {-0} u [1.19e-7, 44.4]: ocml's sqrtf is v_sqrt_f32 followed by
// one step down / one step up in the last place against the exact residual (fma); its x < 2^-96 rescaling never acts, its
// "zero or infinity is its own root" select is kept (u = 1 gives -0).  12 -> 5 and 16 -> 11 VALU instructions, four of each
// per lane and draw of a kernel that is VALU-bound (C5 with the reference's noise stream: 585 instructions per wave in round
// 4).  Bit-equality with torch.randn is what the engine's load-time self-check and tests/test_gpu_kernels.py::
// test_torch_normal_reproduces_torch_randn_bit_for_bit verify on the hardware.
__device__ __forceinline__ float bm_logf(float u) {
#pragma clang fp contract(off)
    const float r = __builtin_amdgcn_logf(u);
    const float head = __uint_as_float(0x3f317217u), tail = __uint_as_float(0x3377d1cfu);
    const float p = r * head;
    float e = __builtin_fmaf(r, head, -p);
    e = __builtin_fmaf(r, tail, e);
    return p + e;
}

__device__ __forceinline__ float bm_sqrtf(float x) {
#pragma clang fp contract(off)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_down = __builtin_fmaf(-down, s, x), e_up = __builtin_fmaf(-up, s, x);
    float t = (0.0f >= e_down) ? down : s;
    t = (0.0f < e_up) ? up : t;
    return __builtin_amdgcn_classf(x, 0x260) ? x : t;          // (+-0, +inf)
}

__device__ __forceinline__ float2 box_muller_u32(uint32_t x, uint32_t y) {
#pragma clang fp contract(on)
    float2 r;
    const float u = 2.3283064e-10f + (x * 2.3283064e-10f);
    const float v = 1.46291807e-09f + (y * 1.46291807e-09f);
    const float s = bm_sqrtf(-2.0f * bm_logf(u));
    __sincosf(v, &r.x, &r.y);
    r.x *= s;
    r.y *= s;
    return r;
}

template <bool WIDE = false>          // (WIDE: see philox2x32_10)
__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t subseq, uint64_t seed) {
    uint32_t c0 = static_cast<uint32_t>(ctr), c1 = static_cast<uint32_t>(ctr >> 32);
    uint32_t c2 = static_cast<uint32_t>(subseq), c3 = static_cast<uint32_t>(subseq >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if constexpr (WIDE) {
            const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
            hi0 = static_cast<uint32_t>(p0 >> 32); lo0 = static_cast<uint32_t>(p0);
            hi1 = static_cast<uint32_t>(p1 >> 32); lo1 = static_cast<uint32_t>(p1);
        } else {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        }
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// Which Philox block and which of its four normals element li of the tensor takes.  n_el <= bg (every image-sized
// latent: bg is n_el rounded up to 256 while the grid is uncapped) means one ATen thread per element -- no 64-bit
// division; past the grid cap bg is 256 * 2048 on this chip, a power of two.
template <bool WIDE = false>          // (WIDE: one 32 x 32 -> 64 multiply per product of a Philox round, see philox2x32_10)
__device__ __forceinline__ float torch_normal(uint64_t li, uint64_t seed, uint64_t offset, uint32_t bg, bool small) {
#pragma clang fp contract(on)
    uint32_t idx, q;
    if (small) {
        idx = static_cast<uint32_t>(li);
        q = 0u;
    } else if ((bg & (bg - 1u)) == 0u) {
        idx = static_cast<uint32_t>(li) & (bg - 1u);
        q = static_cast<uint32_t>(li >> (31 - __builtin_clz(bg)));
    } else {
        idx = static_cast<uint32_t>(li % bg);
        q = static_cast<uint32_t>(li / bg);
    }
    const uint4 c = philox4x32_10<WIDE>((offset >> 2) + (q >> 2), idx, seed);
    // rocrand_normal4 (what ATen calls) = Box-Muller on (c.x, c.y) and on (c.z, c.w); only the pair this element's value comes
    // from is transformed (the other three values belong to elements bg, 2 bg, 3 bg away)
    const uint32_t ii = q & 3u;
    const float2 r = box_muller_u32(ii < 2 ? c.x : c.z, ii < 2 ? c.y : c.w);
    const float v = (ii & 1u) ? r.y : r.x;
    return v * 1.0f + 0.0f;
}

// The four values thread `idx` of ATen's kernel produces with one normal4 call: elements idx, idx + bg, idx + 2 bg,
// idx + 3 bg of the tensor.  The streaming kernels evaluate it per ATen thread and hand the values to the lanes that hold
// those elements through LDS (step_kernel.hip, ST).
__device__ __forceinline__ void torch_normal4(uint32_t idx, uint64_t seed, uint64_t offset, float (&o)[4]) {
#pragma clang fp contract(on)
    const uint4 c = philox4x32_10<true>(offset >> 2, idx, seed);
    const float2 a = box_muller_u32(c.x, c.y), b = box_muller_u32(c.z, c.w);
    o[0] = a.x * 1.0f + 0.0f; o[1] = a.y * 1.0f + 0.0f; o[2] = b.x * 1.0f + 0.0f; o[3] = b.y * 1.0f + 0.0f;
}

// x / s with the reciprocal y = RN(1 / s) of a divisor that several quotients share: q0 = RN(x y), r = x - s q0 (exact in an FMA),
// q = RN(q0 + r y).  With a correctly rounded reciprocal the one residual correction gives the correctly rounded quotient
// (Markstein 1990) -- the value of IEEE division, which the reference's `x_t / c` is -- in 3 instructions per quotient instead
// of the 11 of v_div_scale / v_rcp / 4 x fma / v_div_fmas / v_div_fixup (four elements per lane: 44 -> 12 + one reciprocal, of a
// kernel that is VALU-bound at streaming sizes).  No range scaling: operands and quotients of normal magnitude (latents; 2e7
// random pairs against IEEE division: 0 differences); a non-finite x gives NaN where the division gives inf, -0 gives +0.
__device__ __forceinline__ float div_shared(float x, float s, float y) {
    const float q0 = x * y;
    const float r = __builtin_fmaf(-s, q0, x);
    return __builtin_fmaf(r, y, q0);
}

// ---- 16/32-bit float conversions ----------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }
// fp32 -> bf16, round to nearest even: gfx950's own conversion instruction (v_cvt_pk_bf16_f32, one VALU op per PAIR; the
// shift / add / compare sequence it replaces was ~7 ops per element of a kernel that is VALU-bound at streaming sizes)
typedef __bf16 lp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 b = static_cast<__bf16>(f);
    uint16_t h;
    __builtin_memcpy(&h, &b, 2);
    return h;
}
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {      // lo in bits 0..15
    const lp_f32x2 v = {lo, hi};
    const lp_bf16x2 b = __builtin_convertvector(v, lp_bf16x2);
    uint32_t w;
    __builtin_memcpy(&w, &b, 4);
    return w;
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return static_cast<float>(v);
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    const _Float16 v = static_cast<_Float16>(f);
    uint16_t h;
    __builtin_memcpy(&h, &v, 2);
    return h;
}
// two fp32 -> one dword of two half-width values of storage type `dt` (DT_BF16 / DT_F16), `lo` in bits 0..15
__device__ __forceinline__ uint32_t pack_half2(int dt, float lo, float hi);

// ---- vector loads / stores ------------------------------------------------------
enum : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
__device__ __forceinline__ uint32_t pack_half2(int dt, float lo, float hi) {
    if (dt == DT_BF16) return f32x2_to_bf16x2(lo, hi);
    return static_cast<uint32_t>(f32_to_f16(lo)) | (static_cast<uint32_t>(f32_to_f16(hi)) << 16);
}

// Where the V elements of one lane live: V consecutive elements from index i (an int64_t).
__device__ __forceinline__ int64_t elem_index(int64_t i, int k) { return i + k; }


template <int V>
__device__ __forceinline__ void load_f32(const float* __restrict__ p, int64_t i, float (&o)[V]) {
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + i));
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = p[i + k];
    }
}

template <int V>
__device__ __forceinline__ void store_f32(float* __restrict__ p, int64_t i, const float (&v)[V]) {
    if constexpr (V == 4) {
        // the 16 B/lane path only runs on large latents (streaming, nothing is re-read inside a
        // launch): non-temporal accesses measured +11 % on c5_wan (14.4 -> 12.8 us per launch)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p + i));
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) p[i + k] = v[k];      // (non-temporal here: no effect at the latency-bound sizes, profiles/r04_ab_nt1.log)
    }
}

// Loads of tensors whose storage type is a launch flag come in two halves: `load_raw*` only ISSUES the load
// (the untouched bits stay in registers), `cvt_raw*` decodes them later.  Decoding inside the dtype branch
// would put an s_waitcnt behind every such load -- the bit-packed mask alone cost a full memory round trip
// ahead of all other loads of the step kernel.
template <int V>
struct Raw {
    uint32_t w[V];
};
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <int V>
__device__ __forceinline__ void load_raw(const void* __restrict__ p, int dt, int64_t i, Raw<V>& r) {
    if (dt == DT_F32) {
        float t[V];
        load_f32<V>(static_cast<const float*>(p), i, t);
#pragma unroll
        for (int k = 0; k < V; ++k) r.w[k] = __float_as_uint(t[k]);
    } else {
        const uint16_t* q = static_cast<const uint16_t*>(p) + i;
        if constexpr (V == 4) {      // streaming data like the fp32 path: non-temporal (round 3; bf16 heads past L3)
            const u2 t = __builtin_nontemporal_load(reinterpret_cast<const u2*>(q));
            r.w[0] = t.x; r.w[1] = t.y;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) r.w[k] = q[k];
        }
    }
}



template <int V>
__device__ __forceinline__ void cvt_raw(int dt, const Raw<V>& r, float (&o)[V]) {
    if (dt == DT_F32) {
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = __uint_as_float(r.w[k]);
    } else {
        uint16_t h[V];
        if constexpr (V == 4) {
            h[0] = r.w[0] & 0xffffu; h[1] = r.w[0] >> 16; h[2] = r.w[1] & 0xffffu; h[3] = r.w[1] >> 16;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) h[k] = static_cast<uint16_t>(r.w[k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (dt == DT_BF16) ? bf16_to_f32(h[k]) : f16_to_f32(h[k]);
    }
}

template <int V>
__device__ __forceinline__ void cvt_raw(int dt, const Raw<V>& r, float (&o)[V], int64_t) {
    cvt_raw<V>(dt, r, o);
}

// Same with the storage width fixed at compile time (W = 4: fp32, W = 2: bf16 / fp16, W = 0: run-time `dt`).
template <int V, int W, typename IX>
__device__ __forceinline__ void load_raw_w(const void* __restrict__ p, int dt, const IX& i, Raw<V>& r) {
    if constexpr (W == 4) load_raw<V>(p, DT_F32, i, r);
    else if constexpr (W == 2) load_raw<V>(p, DT_BF16, i, r);        // bf16 and fp16 load alike; cvt_raw tells them apart
    else load_raw<V>(p, dt, i, r);
}

template <int V>
__device__ __forceinline__ void load_any(const void* __restrict__ p, int dt, int64_t i, float (&o)[V]) {
    Raw<V> r;
    load_raw<V>(p, dt, i, r);
    cvt_raw<V>(dt, r, o);
}


template <int V>
__device__ __forceinline__ void store_any(void* __restrict__ p, int dt, int64_t i, const float (&v)[V]) {
    if (dt == DT_F32) {
        store_f32<V>(static_cast<float*>(p), i, v);
    } else {
        uint16_t* q = static_cast<uint16_t*>(p) + i;
        if constexpr (V == 4) {
            u2 t;
            t.x = pack_half2(dt, v[0], v[1]); t.y = pack_half2(dt, v[2], v[3]);
            __builtin_nontemporal_store(t, reinterpret_cast<u2*>(q));
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) q[k] = (dt == DT_BF16) ? f32_to_bf16(v[k]) : f32_to_f16(v[k]);
        }
    }
}

// ---- half-width streams at 16 bytes per access (round 4) -------------------------------------------------------------
// A bf16 / fp16 stream read 8 bytes per lane (4 elements) moves half the bytes of an fp32 stream with the same number of
// memory instructions, and 8-byte accesses run at 0.54-0.70x the per-byte rate of 16-byte ones on this part
// (MI355X_MICROARCH.md, L1-bypassing loads): the 30 B / element launch was no faster than the 36 B one.  So the streaming
// kernels (four elements per lane) fetch half-width data per LANE PAIR: the even lane loads the 16 bytes that hold the
// eight elements of both lanes, the odd lane takes its half through one DPP move per dword (quad_perm, VALU -- no LDS);
// stores go the other way.  Needs the stream 16-byte aligned and rows of a multiple of 8 elements (lp_step checks; other
// launches take the 8-byte path).  Issue / decode split as above: load_raw_pair only issues, cvt_raw_pair exchanges.
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, false));
}

// i: this lane's first element (a multiple of 4; the pair's first element is a multiple of 8); pair_on: either lane of the pair
// reads the stream (region-aware launches leave lines out that no lane reads)
__device__ __forceinline__ void load_raw_pair(const void* __restrict__ p, int64_t i, bool pair_on, Raw<4>& r) {
    if ((threadIdx.x & 1u) == 0u && pair_on) {
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(static_cast<const uint16_t*>(p) + i));
        r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
    }
}

__device__ __forceinline__ void cvt_raw_pair(int dt, const Raw<4>& r, float (&o)[4]) {
    // odd lanes: words 2, 3 of the even partner (quad_perm [0,0,2,2]); even lanes keep their own words 0, 1
    const uint32_t hi0 = dpp_quad<0xA0>(r.w[2]), hi1 = dpp_quad<0xA0>(r.w[3]);
    const bool odd = (threadIdx.x & 1u) != 0u;
    const uint32_t w0 = odd ? hi0 : r.w[0], w1 = odd ? hi1 : r.w[1];
    const uint16_t h[4] = {static_cast<uint16_t>(w0 & 0xffffu), static_cast<uint16_t>(w0 >> 16),
                           static_cast<uint16_t>(w1 & 0xffffu), static_cast<uint16_t>(w1 >> 16)};
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (dt == DT_BF16) ? bf16_to_f32(h[k]) : f16_to_f32(h[k]);
}

// every lane of the wave that is still running calls this (both lanes of a pair are active or neither: rows of 8 k elements)
__device__ __forceinline__ void store_half_pair(void* __restrict__ p, int dt, int64_t i, const float (&v)[4]) {
    const uint32_t w0 = pack_half2(dt, v[0], v[1]), w1 = pack_half2(dt, v[2], v[3]);
    // even lanes: the two words of the odd partner (quad_perm [1,1,3,3]) behind their own
    const uint32_t p0 = dpp_quad<0xF5>(w0), p1 = dpp_quad<0xF5>(w1);
    if ((threadIdx.x & 1u) == 0u) {
        const u4 t = {w0, w1, p0, p1};
        __builtin_nontemporal_store(t, reinterpret_cast<u4*>(static_cast<uint16_t*>(p) + i));
    }
}

// mask -> latent_mask value m (1 = known), same issue / decode split.  LP_FL_MASK_DENOISE applies the
// reference's `1 - (denoise_mask > 0.5)` (nodes.py:281-283) on the fly.
template <int V>
__device__ __forceinline__ void load_mask_raw(const void* __restrict__ p, uint32_t flags, int64_t i, Raw<V>& r) {
    if (flags & LP_FL_MASK_BITS) {
        // 64 lanes x V elements share 2*V words: the loads broadcast out of one cache line
        r.w[0] = static_cast<const uint32_t*>(p)[i >> 5];
    } else if (flags & LP_FL_MASK_U8) {
        const uint8_t* q = static_cast<const uint8_t*>(p) + i;
        if constexpr (V == 4) {
            r.w[0] = *reinterpret_cast<const uint32_t*>(q);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) r.w[k] = q[k];
        }
    } else {
        float t[V];
        load_f32<V>(static_cast<const float*>(p), i, t);
#pragma unroll
        for (int k = 0; k < V; ++k) r.w[k] = __float_as_uint(t[k]);
    }
}



template <int V>
__device__ __forceinline__ void cvt_mask(uint32_t flags, int64_t i, const Raw<V>& r, float (&m)[V]) {
    if (flags & LP_FL_MASK_BITS) {
        const uint32_t sh = static_cast<uint32_t>(i) & 31u;          // V == 4: i % 4 == 0, the nibble never straddles
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = static_cast<float>((r.w[0] >> (sh + k)) & 1u);
        return;
    }
    if (flags & LP_FL_MASK_U8) {
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = static_cast<float>(V == 4 ? ((r.w[0] >> (8 * k)) & 0xffu) : r.w[k]);
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = __uint_as_float(r.w[k]);
    }
    if (flags & LP_FL_MASK_DENOISE) {
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = 1.0f - ((m[k] > 0.5f) ? 1.0f : 0.0f);
    }
}

template <int V>
__device__ __forceinline__ void load_mask(const void* __restrict__ p, uint32_t flags, int64_t i, float (&m)[V]) {
    Raw<V> r;
    load_mask_raw<V>(p, flags, i, r);
    cvt_mask<V>(flags, i, r, m);
}

// ---- per-row coefficient table (K1) ---------------------------------------------------
// One of the FOUR lanes that build a row: lane (g, s) = (region, tau in {dt, dt/2}) evaluates its 3 dependent
// double-precision exp / expm1 factors; lane (0, 0) also writes the row header.  fp32 fields mirror the
// reference's own op order (prepare_step_size, lanpaint.py:295-328); the per-region closed-form factors are
// evaluated in double from those fp32 inputs (lanpaint.py:241-252 in exact arithmetic).  Shared by
// lp_coeffs_kernel and the LP_PH_COEFFS form of the replace launch.
__device__ __forceinline__ float row_scale(bool flow, float abt_f, float ve_f) {        // lanpaint.py:96-99
#pragma clang fp contract(off)
    return flow ? (sqrtf(abt_f) + sqrtf(1.0f - abt_f)) : sqrtf(1.0f + ve_f * ve_f);
}

__device__ inline void coeffs_lane(const lp_hyper& h, float abt_f, float ve_f, float rs_f, float tm_f, float step,
                                   int g, int s, float* __restrict__ c) {
#pragma clang fp contract(off)
    const float oma = 1.0f - abt_f;
    const float dtx2 = 2.0f * step * 1.0f, dty2 = 2.0f * step * h.beta;        // :300-301
    const float dtx = dtx2 / 2.0f, dty = dty2 / 2.0f;                          // :328
    if (g == 0 && s == 0) {                  // the row header
        const float atx = (1.0f / oma) * dtx2 / 2.0f;                              // :315
        const float aty = (h.one_plus_lambda / oma) * dty2 / 2.0f;                 // :316
        const bool valid = step > 0.0f;                                            // :205 (per row)
        c[LP_C_SCALE] = row_scale(h.is_flow, abt_f, ve_f);
        c[LP_C_SQRT_ABT] = sqrtf(abt_f);
        c[LP_C_OMA] = oma;
        c[LP_C_ABT] = abt_f;
        c[LP_C_RSIGMA] = rs_f;
        c[LP_C_DTX] = dtx;
        c[LP_C_DTY] = dty;
        c[LP_C_AX] = atx / dtx;                                                    // :319
        c[LP_C_AY] = aty / dty;                                                    // :320
        c[LP_C_DX] = sqrtf(2.0f);                                                  // :326-327
        c[LP_C_DY] = sqrtf(2.0f);
        c[LP_C_VALID] = valid ? 1.0f : 0.0f;
        c[LP_C_TMODEL] = tm_f;
        c[LP_C_RSCALE] = static_cast<float>(1.0 / static_cast<double>(c[LP_C_SCALE]));      // = 1.0f / scale, see the header
    }
    const double oma_d = static_cast<double>(oma);
    float* q = c + (g ? LP_C_REGION1 : LP_C_REGION0);
    const double a = (g ? static_cast<double>(h.one_plus_lambda) : 1.0) / oma_d;
    const double dt = static_cast<double>(g ? dty : dtx);
    const double tau = s ? dt * 0.5 : dt;
    const double e = exp(-a * tau);
    const double k = -expm1(-a * tau) / a;
    const double k2 = -expm1(-2.0 * a * tau) / (2.0 * a);
    const double sd = sqrt(fmax(2.0 * k2, 0.0));
    q[s ? LP_R_E_HALF : LP_R_E_FULL] = static_cast<float>(e);
    q[s ? LP_R_K_HALF : LP_R_K_FULL] = static_cast<float>(k);
    q[s ? LP_R_STD_HALF : LP_R_STD_FULL] = static_cast<float>(sd);
    if (s == 0) {
        const double cx0 = sqrt(static_cast<double>(abt_f)) / oma_d;
        q[LP_R_DT] = static_cast<float>(dt);
        q[LP_R_A] = static_cast<float>(a);
        q[LP_R_CX0] = static_cast<float>(cx0);
        q[LP_R_CXT] = g ? static_cast<float>(a - 1.0 / oma_d) : 0.0f;
    }
}

// fixed-order wave reduction in double (deterministic)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// N sums at once through DPP lane moves (VALU, no LDS crossbar): a six-sum __shfl_down tree is 72 ds_bpermute and
// measured ~0.5 us per call inside the early-stop step kernel; this is ~0.1 us (scripts/shader_clock.py).  Fixed order
// (butterfly inside each row of 16 lanes, then row 0 -> 1, 2 -> 3, {0,1} -> {2,3}); the total is valid in the LAST
// lane of the wave (kWave - 1) only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<long long>(static_cast<uint32_t>(lo)));
}
template <int CTRL, int ROW_MASK, typename T, int N>
__device__ __forceinline__ void dpp_level(T (&v)[N]) {
    T t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = dpp_move<CTRL, ROW_MASK>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += t[k];
}
template <typename T, int N>
__device__ __forceinline__ void wave_sum_dpp(T (&v)[N]) {
    dpp_level<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    dpp_level<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    dpp_level<0x124, 0xf>(v);     // row_ror 4
    dpp_level<0x128, 0xf>(v);     // row_ror 8: every lane of a row holds the row's sum
    dpp_level<0x142, 0xa>(v);     // row_bcast 15 into rows 1 and 3 (rows 0, 2 add the 0.0 of a masked-off move)
    dpp_level<0x143, 0xc>(v);     // row_bcast 31 into rows 2 and 3: lane 63 = the wave's sum
}

// sigma -> (VE sigma, abt, flow t), nodes.py:243-245 / :250-252, every operation rounded on its own like the reference's
// eager fp32 tensor ops (an FMA here could flip a round() in the n_eff rule).  ONE source for lp_sigma_times_kernel and
// the LP_PH_SIGMA form of the replace launch.
__device__ __forceinline__ void sigma_to_times(float s, bool is_flow, float& ve, float& abt, float& ft) {
#pragma clang fp contract(off)
    if (is_flow) {
        const float a = 1.0f - s;
        const float a2 = a * a;
        const float s2 = s * s;
        const float den = a2 + s2;
        abt = a2 / den;
        ve = s / a;
        ft = s;
    } else {
        ve = s;
        const float s2 = s * s;
        const float den = 1.0f + s2;
        abt = 1.0f / den;
        const float b = 1.0f - abt;
        const float sb = sqrtf(b), sa = sqrtf(abt);
        const float sden = sb + sa;
        ft = sb / sden;
    }
}

// What the first wave of lp_sigma_times_kernel (or of the first block of an LP_PH_SIGMA replace launch) does after the rows:
// the two scalars of the inner-step rule, the rule against a speculated count, the mailbox.
struct SigmaRule {
    int32_t n_steps, early_stop, total_steps, guess;       // guess < 0: not speculating (the word is set to 1)
    double min_step_frac;
    uint64_t* valid_out;                                   // device word, or nullptr: no rule on the device
};

__host__ __device__ inline int32_t effective_inner_steps(int32_t n_steps, double step_f, double frac, int32_t total_steps,
                                                         int32_t early_stop, double min_step_frac);

// Run by ONE FULL WAVE (every lane alive): the rows and the schedule entries are loaded lane-parallel -- one memory round
// trip each instead of a scalar load + wait per entry (a 30-entry schedule: ~3 us of a thread-0 walk on the critical path of
// every sigma call of the node path) -- while every floating-point sum keeps the sequential order of the one-thread form
// (readlane by readlane), and the arg-min keeps torch.argmin's first-minimum rule (ties go to the lower index).  Lane 0
// then applies the rule and posts the mailbox.
__device__ __forceinline__ void sigma_rows_and_rule(const float* __restrict__ sigma, int rows, const float* __restrict__ schedule,
                                                    int schedule_len, bool is_flow, float* __restrict__ times,
                                                    float* __restrict__ scalars, int32_t* __restrict__ seq_out, int32_t seq,
                                                    const SigmaRule& rule) {
#pragma clang fp contract(off)
    const int lane = static_cast<int>(threadIdx.x) & (kWave - 1);
    float sum_sigma = 0.0f, sum_oma = 0.0f;
    for (int r0 = 0; r0 < rows; r0 += kWave) {
        const int r = r0 + lane;
        float s = 0.0f, oma = 0.0f;
        if (r < rows) {
            s = sigma[r];
            float ve, abt, ft;
            sigma_to_times(s, is_flow, ve, abt, ft);
            if (times) {
                times[r] = ve;
                times[rows + r] = abt;
                times[2 * rows + r] = ft;
            }
            oma = 1.0f - abt;
        }
        const int n = rows - r0 < kWave ? rows - r0 : kWave;
        for (int k = 0; k < n; ++k) {                                // the one-thread order: row after row
            sum_sigma = sum_sigma + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), k));
            sum_oma = sum_oma + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, oma), k));
        }
    }
    const float mean_sigma = sum_sigma / static_cast<float>(rows);
    int best = 0;
    float best_d = INFINITY;
    for (int i = lane; i < schedule_len; i += kWave) {               // this lane's entries, first minimum among them
        const float diff = schedule[i] - mean_sigma;
        const float dd = fabsf(diff);
        if (dd < best_d) {
            best_d = dd;
            best = i;
        }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {                      // first minimum over the wave (every lane ends with it)
        const float od = __shfl_xor(best_d, off, kWave);
        const int ob = __shfl_xor(best, off, kWave);
        if (od < best_d || (od == best_d && ob < best)) {
            best_d = od;
            best = ob;
        }
    }
    if (lane != 0) return;
    const float frac = sum_oma / static_cast<float>(rows);
    scalars[0] = static_cast<float>(best);
    scalars[1] = frac;
    if (rule.valid_out) {
        const int32_t n_eff = effective_inner_steps(rule.n_steps, static_cast<double>(static_cast<float>(best)),
                                                    static_cast<double>(frac), rule.total_steps, rule.early_stop,
                                                    rule.min_step_frac);
        *rule.valid_out = (rule.guess < 0 || rule.guess == n_eff) ? 1ull : 0ull;
        scalars[3] = static_cast<float>(n_eff);            // (word 2 of the mailbox is the sequence number)
    }
    if (seq_out) {       // mailbox in pinned host memory: the sequence number lands after the scalars
        __threadfence_system();
        __hip_atomic_store(seq_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The inner-step rule of KSamplerX0Inpaint.__call__ (nodes.py:286-299 + min_step_frac_effective_steps, :134-144) as Python
// evaluates it: int(step) compare, then doubles and round-half-even.  ONE source for the host (lp_node_call) and the device
// (lp_sigma_times_kernel, which checks a speculated count): n * frac is exact in double (a 24-bit by a 31-bit integer
// mantissa), the division is correctly rounded on both, rint ties to even on both -- the two cannot disagree.
__host__ __device__ inline int32_t effective_inner_steps(int32_t n_steps, double step_f, double frac, int32_t total_steps,
                                                         int32_t early_stop, double min_step_frac) {
    if (total_steps - static_cast<int32_t>(step_f) <= early_stop) return 0;
    if (min_step_frac <= 0.0 || frac >= min_step_frac || n_steps <= 0) return n_steps;
    const double r = rint(static_cast<double>(n_steps) * frac / min_step_frac);
    return r > 0.0 ? static_cast<int32_t>(r) : 0;
}

__host__ __device__ inline int x0_dtype(uint32_t flags) {
    return (flags & LP_FL_X0_BF16) ? DT_BF16 : (flags & LP_FL_X0_F16) ? DT_F16 : DT_F32;
}
__host__ __device__ inline int xin_dtype(uint32_t flags) {
    return (flags & LP_FL_XIN_BF16) ? DT_BF16 : (flags & LP_FL_XIN_F16) ? DT_F16 : DT_F32;
}

}  // namespace lp
