// cabi_host.cpp -- a host WITHOUT Python or torch drives one sigma call through the C ABI (include/lanpaint_hip.h)
// and checks it against the scalar C restatement of the reference (oracle/langevin_oracle.c, linked into this TEST
// binary only).  What a C / C++ / Go-cgo / JNI host would do: hipMalloc'd buffers, raw pointers, POD descriptors, the
// caller's stream, int status codes.  Built and run by tests/test_cabi_host.py (hipcc; needs an MI355X to run).
//
//   sigma call = lp_coeffs ; lp_step(REPLACE|EMIT) ; n x [ backbone (here: host lambda x -> (0.9x, 0.8x)),
//                lp_step(POST|PRE_HALF|EMIT) ] ; backbone ; lp_finalize            (lanpaint.py:56-157)
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lanpaint_hip.h"

extern "C" {   // oracle/langevin_oracle.c
typedef struct {
    float abt, ve_sigma, step;
    float lambda, one_plus_lambda, beta;
    int is_flow;
} orc_row;
void orc_replace_rescale(const orc_row* r, int64_t n, const float* x, const float* known, const float* mask, float* x_t);
void orc_to_model_space(const orc_row* r, int64_t n, const float* x_t, float* x);
void orc_first_step(const orc_row* r, int64_t n, float* x_t, const float* x0, const float* x0b, const float* y,
                    const float* mask, const float* xi, float* C, float* x0s);
void orc_half_step(const orc_row* r, int64_t n, float* x_t, const float* mask, const float* xi, const float* C);
void orc_steady_post(const orc_row* r, int64_t n, float* x_t, const float* x0, const float* x0b, const float* y,
                     const float* mask, const float* xi, float* C, float* x0s);
void orc_finalize(int64_t n, const float* model_out, const float* y, const float* mask, float* out);
}

#define HIP_OK(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));      \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define LP_CHECK(call)                                                                  \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != LP_OK) {                                                             \
            std::fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, lp_strerror(rc_));      \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static float gauss() {   // Box-Muller over a 64-bit LCG: any fixed stream will do, both sides consume the same numbers
    auto u = [] {
        lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
        return (static_cast<double>(lcg_state >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    };
    return static_cast<float>(std::sqrt(-2.0 * std::log(u())) * std::cos(6.283185307179586 * u()));
}

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) { n = count; return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(float)); }
    hipError_t put(const std::vector<float>& h) { return hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice); }
    hipError_t get(std::vector<float>& h) const { h.resize(n); return hipMemcpy(h.data(), p, n * sizeof(float), hipMemcpyDeviceToHost); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int main(int argc, char** argv) {
    const int is_flow = argc > 1 ? std::atoi(argv[1]) : 0;
    const int n_steps = argc > 2 ? std::atoi(argv[2]) : 4;
    const int64_t C = 4, H = 24, W = 20, n = C * H * W;                 // one batch row, 1920 elements
    const float sigma = is_flow ? 0.6f : 1.7f, lambda = 5.0f, beta = 1.0f, step_size = 0.2f;
    if (lp_abi_version() != LP_ABI_VERSION) {
        std::fprintf(stderr, "ABI mismatch: library %d, header %d\n", lp_abi_version(), LP_ABI_VERSION);
        return 4;
    }

    // ---- inputs (host) -------------------------------------------------------------------------------------
    std::vector<float> y(n), noise(n), x(n), mask(n);
    for (int64_t i = 0; i < n; ++i) {
        y[i] = gauss();
        noise[i] = gauss();
        x[i] = is_flow ? sigma * noise[i] + (1.0f - sigma) * y[i] : y[i] + noise[i] * sigma;
        mask[i] = (i % W) < W / 2 ? 1.0f : 0.0f;                         // 50 % box, 1 = known
    }
    std::vector<std::vector<float>> xi(2 * n_steps, std::vector<float>(n));
    for (auto& v : xi)
        for (auto& e : v) e = gauss();
    const float abt = is_flow ? (1 - sigma) * (1 - sigma) / ((1 - sigma) * (1 - sigma) + sigma * sigma) : 1.0f / (1.0f + sigma * sigma);
    const float ve = is_flow ? sigma / (1.0f - sigma) : sigma;

    // ---- the CPU restatement ---------------------------------------------------------------------------------
    orc_row r{abt, ve, step_size * std::fmax(1.0f - abt, 0.0f), lambda, 1.0f + lambda, beta, is_flow};
    std::vector<float> known(n), xt_o(n), xin_o(n), c_o(n), x0s_o(n), h0(n), h1(n), out_o(n), x_o(n);
    for (int64_t i = 0; i < n; ++i) known[i] = is_flow ? sigma * noise[i] + (1.0f - sigma) * y[i] : y[i] + noise[i] * sigma;
    orc_replace_rescale(&r, n, x.data(), known.data(), mask.data(), xt_o.data());
    int draw = 0;
    for (int it = 0; it < n_steps; ++it) {
        if (it > 0) orc_half_step(&r, n, xt_o.data(), mask.data(), xi[draw++].data(), c_o.data());
        orc_to_model_space(&r, n, xt_o.data(), xin_o.data());
        for (int64_t i = 0; i < n; ++i) { h0[i] = 0.9f * xin_o[i]; h1[i] = 0.8f * xin_o[i]; }
        if (it == 0) orc_first_step(&r, n, xt_o.data(), h0.data(), h1.data(), y.data(), mask.data(), xi[draw++].data(), c_o.data(), x0s_o.data());
        else orc_steady_post(&r, n, xt_o.data(), h0.data(), h1.data(), y.data(), mask.data(), xi[draw++].data(), c_o.data(), x0s_o.data());
    }
    orc_to_model_space(&r, n, xt_o.data(), x_o.data());
    for (int64_t i = 0; i < n; ++i) h0[i] = 0.9f * x_o[i];
    orc_finalize(n, h0.data(), y.data(), mask.data(), out_o.data());

    // ---- the same sigma call through the C ABI ---------------------------------------------------------------
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    DevBuf dx, dy, dnoise, dmask, dxt, dC, dxin, dx0, dx0b, dxia, dxib, dout, dcoef, dsig, dve, dabt;
    for (DevBuf* b : {&dx, &dy, &dnoise, &dmask, &dxt, &dC, &dxin, &dx0, &dx0b, &dxia, &dxib, &dout}) HIP_OK(b->alloc(n));
    HIP_OK(dcoef.alloc(LP_COEF_STRIDE));
    HIP_OK(dsig.alloc(1)); HIP_OK(dve.alloc(1)); HIP_OK(dabt.alloc(1));
    HIP_OK(dx.put(x)); HIP_OK(dy.put(y)); HIP_OK(dnoise.put(noise)); HIP_OK(dmask.put(mask));
    HIP_OK(dsig.put({sigma})); HIP_OK(dve.put({ve})); HIP_OK(dabt.put({abt}));

    lp_hyper hy{lambda, beta, step_size, 0.0f, is_flow, 1.0f + lambda};
    LP_CHECK(lp_coeffs(&hy, dve.p, 0, dabt.p, 0, dsig.p, 0, nullptr, 0, is_flow ? dsig.p : dve.p, 0, 1, dcoef.p, stream));
    lp_step_desc d{};
    d.n_el = n; d.el_per_row = n; d.rows = 1;
    d.flags = is_flow ? LP_FL_FLOW : 0;
    d.lambda = lambda; d.one_plus_lambda = 1.0f + lambda; d.beta = beta; d.step_size = step_size; d.noise_scale = 1.0f;
    d.coef = dcoef.p; d.x = dx.p; d.noise = dnoise.p; d.y = dy.p; d.mask = dmask.p; d.x_t = dxt.p; d.C = dC.p; d.x_in = dxin.p;
    d.replace_kind = is_flow ? LP_REPLACE_FLOW : LP_REPLACE_VE;
    d.phases = LP_PH_REPLACE | LP_PH_EMIT;
    LP_CHECK(lp_step(&d, stream));                                        // lanpaint.py:94-99
    std::vector<float> host(n), g0(n), g1(n);
    draw = 0;
    for (int it = 0; it < n_steps; ++it) {
        const bool last = it == n_steps - 1;
        HIP_OK(hipStreamSynchronize(stream));                             // the "backbone" of this host runs on the CPU
        HIP_OK(dxin.get(host));
        for (int64_t i = 0; i < n; ++i) { g0[i] = 0.9f * host[i]; g1[i] = 0.8f * host[i]; }
        HIP_OK(dx0.put(g0)); HIP_OK(dx0b.put(g1));
        // the reference's draw order: the POST draw of iteration it, then the PRE draw of iteration it + 1
        const int post_draw = it == 0 ? 0 : 2 * it, pre_draw = 2 * it + 1;
        HIP_OK(dxia.put(xi[post_draw]));
        if (!last) HIP_OK(dxib.put(xi[pre_draw]));
        d.x0 = dx0.p; d.x0_big = dx0b.p; d.xi_post = dxia.p; d.xi_pre = last ? nullptr : dxib.p;
        d.phases = (it == 0 ? LP_PH_POST_FIRST : LP_PH_POST_STEADY) | (last ? 0u : LP_PH_PRE_HALF) | LP_PH_EMIT;
        LP_CHECK(lp_step(&d, stream));                                    // lanpaint.py:159-184, 212-254, 274-286
    }
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(dxin.get(host));                                               // the final model-space x
    for (int64_t i = 0; i < n; ++i) g0[i] = 0.9f * host[i];
    HIP_OK(dx0.put(g0));
    lp_final_desc f{};
    f.n_el = n; f.model_out = dx0.p; f.y = dy.p; f.mask = dmask.p; f.x_src = dxin.p; f.x_dst = dx.p; f.out = dout.p;
    LP_CHECK(lp_finalize(&f, stream));                                    // lanpaint.py:154, 156
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> out_g, x_g;
    HIP_OK(dout.get(out_g)); HIP_OK(dx.get(x_g));

    // ---- compare -----------------------------------------------------------------------------------------------
    // the oracle's draw bookkeeping: iteration 0 consumed draw 0; iteration it >= 1 consumed 2 it - 1 (PRE) and 2 it (POST)
    double err_out = 0.0, err_x = 0.0, scale = 1.0;
    for (int64_t i = 0; i < n; ++i) {
        err_out = std::fmax(err_out, std::fabs(static_cast<double>(out_g[i]) - out_o[i]));
        err_x = std::fmax(err_x, std::fabs(static_cast<double>(x_g[i]) - x_o[i]));
        scale = std::fmax(scale, std::fabs(static_cast<double>(x_o[i])));
    }
    std::printf("flow=%d n_steps=%d n=%lld  max|out - oracle|=%.3e  max|x - oracle|=%.3e  (scale %.2f)\n", is_flow, n_steps,
                static_cast<long long>(n), err_out, err_x, scale);
    (void)hipStreamDestroy(stream);
    return (err_out <= 5e-5 * scale && err_x <= 5e-5 * scale) ? 0 : 1;
}
