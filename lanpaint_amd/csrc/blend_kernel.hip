// blend_kernel.hip -- post-decode mask blend for gfx950: dilate (max-pool) + Gaussian blur of the
// mask and the image lerp in ONE launch.  Restates MaskBlend.blend_images (reference
// nodes.py:610-638), merge_video_with_mask (nodes.py:1060-1088) and gaussian_kernel_2d (:1049-1057).
//
// A block owns a TH x TW tile of one image.  The mask tile plus a 2R halo (R = k/2: R for the
// dilation, R for the blur) is staged in LDS once; four separable passes run entirely in LDS
//   A (raw, -inf outside the image)  --row max-->  B  --col max, 0 outside the image-->  C
//   C  --row blur-->  D  --col blur--> m (registers)  --> out = image1*(1-m) + image2*m
// The reference's 2-D kernel exp(-(x^2+y^2)/(2 s^2))/sum is exactly the outer product of the
// normalised 1-D profile, so the separable form differs from conv2d only in summation order.
#include "lp_common.h"

namespace lp {

template <int TH, int TW>
__global__ __launch_bounds__(256) void lp_mask_blend_kernel(const lp_blend_desc d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = d.k, R = k / 2;
    const int AW = TW + 4 * R, AH = TH + 4 * R, BW = TW + 2 * R, CH = TH + 2 * R;
    float* A = lds;                   // AH x AW raw mask; later C: CH x BW dilated
    float* B = A + AH * AW;           // AH x BW row-max;  later D: CH x TW row-blurred
    float* g = B + AH * BW;           // k normalised 1-D Gaussian weights
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int H = d.height, W = d.width;

    if (tid < k) {                    // gaussian_kernel_2d: sigma = (k-1)/4; identity for k <= 1
        float w = 1.0f;
        if (k > 1) {
            const float sigma = static_cast<float>(k - 1) / 4.0f, inv = 1.0f / (2.0f * sigma * sigma);
            float sum = 0.0f;
            for (int j = 0; j < k; ++j) sum += expf(-static_cast<float>((j - R) * (j - R)) * inv);
            w = expf(-static_cast<float>((tid - R) * (tid - R)) * inv) / sum;
        }
        g[tid] = w;
    }
    const bool resample = d.mask_h != H || d.mask_w != W;
    const float* mplane = d.mask + static_cast<int64_t>(d.mask_batch == 1 ? 0 : b) * d.mask_h * d.mask_w;
    for (int idx = tid; idx < AH * AW; idx += 256) {
        const int ay = idx / AW, ax = idx - ay * AW;
        const int y = y0 - 2 * R + ay, x = x0 - 2 * R + ax;
        float v = -INFINITY;                                   // max_pool2d pads with -inf
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const int sy = resample ? nearest_exact_index(y, d.mask_h, H, d.nn_rule) : y;
            const int sx = resample ? nearest_exact_index(x, d.mask_w, W, d.nn_rule) : x;
            v = mplane[static_cast<int64_t>(sy) * d.mask_w + sx];
        }
        A[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < AH * BW; idx += 256) {           // row max over the k-wide window
        const int ay = idx / BW, bx = idx - ay * BW;
        const float* row = A + ay * AW + bx;
        float v = row[0];
        for (int j = 1; j < k; ++j) v = fmaxf(v, row[j]);
        B[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < CH * BW; idx += 256) {           // column max; conv2d pads with ZERO
        const int cy = idx / BW, bx = idx - cy * BW;
        const int y = y0 - R + cy, x = x0 - R + bx;
        float v = 0.0f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            v = B[cy * BW + bx];
            for (int j = 1; j < k; ++j) v = fmaxf(v, B[(cy + j) * BW + bx]);
        }
        A[idx] = v;                                            // C
    }
    __syncthreads();
    for (int idx = tid; idx < CH * TW; idx += 256) {           // row blur
        const int cy = idx / TW, tx = idx - cy * TW;
        const float* row = A + cy * BW + tx;
        float v = 0.0f;
        for (int j = 0; j < k; ++j) v += g[j] * row[j];
        B[idx] = v;                                            // D
    }
    __syncthreads();
    const int C = d.channels;
    for (int idx = tid; idx < TH * TW; idx += 256) {           // column blur + blend
        const int ty = idx / TW, tx = idx - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        float m = 0.0f;
        for (int j = 0; j < k; ++j) m += g[j] * B[(ty + j) * TW + tx];
        const int64_t pix = (static_cast<int64_t>(b) * H + y) * W + x;
        if (d.smooth_out) d.smooth_out[pix] = m;
        if (d.out) {
            const float* p1 = d.image1 + pix * C;
            const float* p2 = d.image2 + pix * C;
            float* po = d.out + pix * C;
            for (int c = 0; c < C; ++c) po[c] = p1[c] * (1.0f - m) + p2[c] * m;
        }
    }
}

template <int TH, int TW>
static hipError_t launch_blend(const lp_blend_desc& d, hipStream_t stream) {
    const int R = d.k / 2;
    const size_t lds = sizeof(float) * (static_cast<size_t>(TH + 4 * R) * (TW + 4 * R) +
                                        static_cast<size_t>(TH + 4 * R) * (TW + 2 * R) + d.k);
    // above the default 64 KiB cap the dynamic-LDS limit has to be raised (160 KiB per CU on gfx950).  The attribute is
    // per DEVICE, so it is set on whatever device this launch goes to -- no process-wide "done" flag (the library
    // keeps no state; a second GPU of the same process would otherwise fail to launch)
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lp_mask_blend_kernel<TH, TW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const dim3 grid((d.width + TW - 1) / TW, (d.height + TH - 1) / TH, d.batch);
    hipLaunchKernelGGL((lp_mask_blend_kernel<TH, TW>), grid, dim3(256), lds, stream, d);
    return hipGetLastError();
}

int blend_dispatch(const lp_blend_desc* dp, hipStream_t stream) {
    if (!dp) return LP_E_INVALID;
    const lp_blend_desc& d = *dp;
    if (d.batch <= 0 || d.height <= 0 || d.width <= 0 || d.channels <= 0 || !d.mask) return LP_E_INVALID;
    if (d.k < 1 || d.k > 51 || (d.k % 2) == 0) return LP_E_INVALID;
    if (d.mask_batch != 1 && d.mask_batch != d.batch) return LP_E_INVALID;
    if (d.mask_h <= 0 || d.mask_w <= 0) return LP_E_INVALID;
    if (d.nn_rule < LP_NN_ATEN_SCALAR || d.nn_rule > LP_NN_ATEN_CPU_GENERIC) return LP_E_INVALID;
    if (!d.out && !d.smooth_out) return LP_E_INVALID;
    if (d.out && (!d.image1 || !d.image2)) return LP_E_INVALID;
    if (d.batch > 65535 || (d.height + 7) / 8 > 65535) return LP_E_UNSUPPORTED;
    // small halos: wide tiles; large halos: the tile shrinks so tile + halo stays inside 160 KiB of LDS
    const hipError_t err = (d.k <= 15) ? launch_blend<16, 64>(d, stream) : launch_blend<8, 32>(d, stream);
    return err == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

}  // namespace lp
