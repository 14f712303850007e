// lp_common.h -- device helpers shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "lanpaint_hip.h"

namespace lp {

constexpr int kWave = 64;   // CDNA wavefront width; hard-coded per the gfx950 programming guide

// ---- Philox4x32-10 (Salmon et al. 2011) -------------------------------------
// Counter-based: one call -> 4 x u32, keyed on (seed), indexed by
// (element quad, launch sequence number, draw slot).  No state is carried, so the
// stream is independent of grid shape, vector width and block size.
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += W0;
        k1 += W1;
    }
    return c;
}

// u32 -> uniform in (0, 1]: (x + 0.5) * 2^-32 (never 0, so log is finite).
__device__ __forceinline__ float u01(uint32_t x) {
    return fmaf(static_cast<float>(x), 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// 4 standard normals from one Philox block (two Box-Muller pairs).  v_sin_f32 /
// v_cos_f32 take their argument in revolutions, which is exactly 2*pi*u.
__device__ __forceinline__ void normal4(uint64_t quad, uint64_t seq, uint32_t slot, uint64_t seed, float (&z)[4]) {
    // 44 bits of quad index (2^46 elements) and 52 bits of launch sequence share the
    // 128-bit counter with the slot, so distinct (quad, seq, slot) never collide.
    const u32x4 ctr{static_cast<uint32_t>(quad), static_cast<uint32_t>(seq),
                    static_cast<uint32_t>(seq >> 32) | (static_cast<uint32_t>(quad >> 32) << 20), slot};
    const u32x4 r = philox4x32_10(ctr, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    const float r1 = sqrtf(-2.0f * __logf(u01(r.x)));
    const float r2 = sqrtf(-2.0f * __logf(u01(r.z)));
    const float t1 = u01(r.y), t2 = u01(r.w);
    z[0] = r1 * __builtin_amdgcn_cosf(t1);
    z[1] = r1 * __builtin_amdgcn_sinf(t1);
    z[2] = r2 * __builtin_amdgcn_cosf(t2);
    z[3] = r2 * __builtin_amdgcn_sinf(t2);
}

// ---- 16/32-bit float conversions ----------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
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

// ---- vector loads / stores ------------------------------------------------------
enum : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

template <int V>
__device__ __forceinline__ void load_f32(const float* __restrict__ p, int64_t i, float (&o)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + i);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = p[i + k];
    }
}

template <int V>
__device__ __forceinline__ void store_f32(float* __restrict__ p, int64_t i, const float (&v)[V]) {
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) p[i + k] = v[k];
    }
}

// dtype is wave-uniform (a launch flag), so the branch costs one scalar compare.
template <int V>
__device__ __forceinline__ void load_any(const void* __restrict__ p, int dt, int64_t i, float (&o)[V]) {
    if (dt == DT_F32) {
        load_f32<V>(static_cast<const float*>(p), i, o);
    } else {
        const uint16_t* q = static_cast<const uint16_t*>(p) + i;
        uint16_t h[V];
        if constexpr (V == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(q);
            h[0] = t.x & 0xffffu; h[1] = t.x >> 16; h[2] = t.y & 0xffffu; h[3] = t.y >> 16;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) h[k] = q[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (dt == DT_BF16) ? bf16_to_f32(h[k]) : f16_to_f32(h[k]);
    }
}

template <int V>
__device__ __forceinline__ void store_any(void* __restrict__ p, int dt, int64_t i, const float (&v)[V]) {
    if (dt == DT_F32) {
        store_f32<V>(static_cast<float*>(p), i, v);
    } else {
        uint16_t h[V];
#pragma unroll
        for (int k = 0; k < V; ++k) h[k] = (dt == DT_BF16) ? f32_to_bf16(v[k]) : f32_to_f16(v[k]);
        uint16_t* q = static_cast<uint16_t*>(p) + i;
        if constexpr (V == 4) {
            *reinterpret_cast<uint2*>(q) = make_uint2(h[0] | (uint32_t(h[1]) << 16), h[2] | (uint32_t(h[3]) << 16));
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) q[k] = h[k];
        }
    }
}

// mask -> latent_mask value m (1 = known).  LP_FL_MASK_DENOISE applies the
// reference's `1 - (denoise_mask > 0.5)` (nodes.py:281-283) on the fly.
template <int V>
__device__ __forceinline__ void load_mask(const void* __restrict__ p, uint32_t flags, int64_t i, float (&m)[V]) {
    if (flags & LP_FL_MASK_U8) {
        const uint8_t* q = static_cast<const uint8_t*>(p) + i;
        if constexpr (V == 4) {
            const uint32_t t = *reinterpret_cast<const uint32_t*>(q);
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = static_cast<float>((t >> (8 * k)) & 0xffu);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) m[k] = static_cast<float>(q[k]);
        }
    } else {
        load_f32<V>(static_cast<const float*>(p), i, m);
    }
    if (flags & LP_FL_MASK_DENOISE) {
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = 1.0f - ((m[k] > 0.5f) ? 1.0f : 0.0f);
    }
}

__host__ __device__ inline int x0_dtype(uint32_t flags) {
    return (flags & LP_FL_X0_BF16) ? DT_BF16 : (flags & LP_FL_X0_F16) ? DT_F16 : DT_F32;
}
__host__ __device__ inline int xin_dtype(uint32_t flags) {
    return (flags & LP_FL_XIN_BF16) ? DT_BF16 : (flags & LP_FL_XIN_F16) ? DT_F16 : DT_F32;
}

}  // namespace lp
