// step_kernel.hip -- the fused Langevin "think" step for gfx950 (CDNA4, wave64).
//
// One launch = [everything after backbone call i] + [everything before call i+1],
// one pass over the latent, straight HBM -> VGPR -> HBM with coalesced loads (no element
// is reused inside the pass, so an LDS stage would only add latency; LDS is used where
// there IS reuse: the mask-edge stencil in aux_kernels.hip).
//   - blockIdx.y = batch row: the per-row coefficient table lands in SGPRs via scalar
//     loads, and with a binary mask no transcendental is evaluated per element.
//   - the hot phase combinations are compile-time specialisations (PH), so every load of
//     the launch is issued up front and the Philox / Box-Muller arithmetic runs while they
//     are in flight; PH = 0 is the generic kernel that reads the phases at run time.
//   - VEC = 4 (16 B/lane) streams large latents; VEC = 1 spreads a small latent over 4x
//     more waves so that all four SIMDs of every CU work on the (latency-bound) launch.
//
// Math restated from the reference (file:line = /root/reference/src/LanPaint/):
//   replace + VP rescale        lanpaint.py:94-99
//   masked score split, Coef_C  lanpaint.py:159-184, 212-220
//   exact OU step + noise       lanpaint.py:232-254
//   overdamped scheme           lanpaint.py:274-286 (second half-step uses the OLD C)
//   back to model space         lanpaint.py:144-147, 163, 168
#include <cstddef>

#include "lp_common.h"

// No implicit a*b+c contraction: the table path spells its fused multiply-adds out (fmaf), the general path
// rounds every product and sum on its own as the reference's eager PyTorch ops do.  The many instantiations
// of lp_step_kernel then compute bit-identical results whichever one a launch is routed to (fp32 / uint8 /
// bit-packed mask, fused or unfused schedule) instead of depending on what the optimiser fused where.
#pragma clang fp contract(off)

namespace lp {

struct RegionCoef {
    float e_full, k_full, std_full, e_half, k_half, std_half, dt, a, cx0, cxt;
};

struct RowCoef {
    float scale, sqrt_abt, oma, abt, rsigma, dtx, dty, ax, ay, dx, dy, valid;
    RegionCoef reg[2];
    float rscale;          // RN(1 / scale) (LP_C_RSCALE; load_row only: the one-element-per-lane kernels divide)
};

__device__ __forceinline__ RowCoef load_row(const float* __restrict__ coef, int row) {
    const float* c = coef + static_cast<int64_t>(row) * LP_COEF_STRIDE;   // wave-uniform -> s_load
    RowCoef r;
    r.scale = c[LP_C_SCALE]; r.sqrt_abt = c[LP_C_SQRT_ABT]; r.oma = c[LP_C_OMA]; r.abt = c[LP_C_ABT];
    r.rsigma = c[LP_C_RSIGMA]; r.dtx = c[LP_C_DTX]; r.dty = c[LP_C_DTY]; r.ax = c[LP_C_AX]; r.ay = c[LP_C_AY];
    r.dx = c[LP_C_DX]; r.dy = c[LP_C_DY]; r.valid = c[LP_C_VALID];
    r.rscale = c[LP_C_RSCALE];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* q = c + (g ? LP_C_REGION1 : LP_C_REGION0);
        r.reg[g] = RegionCoef{q[LP_R_E_FULL], q[LP_R_K_FULL], q[LP_R_STD_FULL], q[LP_R_E_HALF], q[LP_R_K_HALF],
                              q[LP_R_STD_HALF], q[LP_R_DT], q[LP_R_A], q[LP_R_CX0], q[LP_R_CXT]};
    }
    return r;
}

// The same row through VECTOR loads of one address (eight dwordx4, every lane the same words).  Scalar loads are
// "invariant" to the compiler, which sinks them to their first use -- behind the first wait for the argument segment --
// and then waits for them there: a second dependent round trip in front of the arithmetic.  A vector load stays where
// it is written, i.e. it is in flight from kernel entry when the table pointer is preloaded (VEC = 1 kernels).
__device__ __forceinline__ RowCoef load_row_early(const float* coef, int row) {
    uint32_t lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));              // opaque: keeps the address in a VGPR
    const float4* c = reinterpret_cast<const float4*>(coef + static_cast<int64_t>(row) * LP_COEF_STRIDE) + lane_zero;
    float w[32];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 q = c[k];
        w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
    }
    static_assert(LP_COEF_STRIDE % 4 == 0 && LP_C_REGION1 + 10 <= 32, "coefficient row layout");
    RowCoef r;
    r.scale = w[LP_C_SCALE]; r.sqrt_abt = w[LP_C_SQRT_ABT]; r.oma = w[LP_C_OMA]; r.abt = w[LP_C_ABT];
    r.rsigma = w[LP_C_RSIGMA]; r.dtx = w[LP_C_DTX]; r.dty = w[LP_C_DTY]; r.ax = w[LP_C_AX]; r.ay = w[LP_C_AY];
    r.dx = w[LP_C_DX]; r.dy = w[LP_C_DY]; r.valid = w[LP_C_VALID];
    // slot 33 (LP_C_RSCALE) lies past the eight quads fetched here.  Only the VEC == 4 paths divide through rc.rscale (div_shared
    // in the replace and emit phases, both under `VEC == 4 && PH != 0`) and they load their row with load_row; this loader is
    // reached under `VEC == 1 && PH != 0` alone.  Poisoned, so that a future VEC == 1 use shows up as NaN in the first parity test
    // instead of as a silent division by zero.
    r.rscale = __builtin_nanf("");
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* q = w + (g ? LP_C_REGION1 : LP_C_REGION0);
        r.reg[g] = RegionCoef{q[LP_R_E_FULL], q[LP_R_K_FULL], q[LP_R_STD_FULL], q[LP_R_E_HALF], q[LP_R_K_HALF],
                              q[LP_R_STD_HALF], q[LP_R_DT], q[LP_R_A], q[LP_R_CX0], q[LP_R_CXT]};
    }
    return r;
}

// per-element coefficient set of the GENERAL path (soft masks, per-element times);
// op order mirrors prepare_step_size (lanpaint.py:295-328) + lanpaint.py:212-214.
struct ElemCoef {
    float a, d, dt, sqrt_abt, oma, scale;
    bool valid;
};

__device__ __forceinline__ ElemCoef elem_from_row(const RowCoef& rc, float m) {
    const float om = 1.0f - m;
    ElemCoef e;
    e.a = rc.ax * om + rc.ay * m;
    e.d = rc.dx * om + rc.dy * m;
    e.dt = rc.dtx * om + rc.dty * m;
    e.sqrt_abt = rc.sqrt_abt;
    e.oma = rc.oma;
    e.scale = rc.scale;
    e.valid = rc.valid != 0.0f;
    return e;
}

__device__ __forceinline__ ElemCoef elem_from_times(float abt, float ve, float m, bool flow, float opl, float beta,
                                                    float step_size, float min_step_frac) {
    const float oma = 1.0f - abt;
    const float step = step_size * fmaxf(oma, min_step_frac);        // lanpaint.py:81
    const float dtx2 = 2.0f * step * 1.0f, dty2 = 2.0f * step * beta;  // :300-301 (sigma_x = 1, sigma_y = beta)
    const float atx = (1.0f / oma) * dtx2 / 2.0f;                     // :315
    const float aty = (opl / oma) * dty2 / 2.0f;                      // :316
    const float ax = atx / (dtx2 / 2.0f), ay = aty / (dty2 / 2.0f);   // :319-320
    const float dxy = sqrtf(2.0f);                                    // :326-327
    const float om = 1.0f - m;
    ElemCoef e;
    e.a = ax * om + ay * m;
    e.d = dxy * om + dxy * m;
    e.dt = (dtx2 / 2.0f) * om + (dty2 / 2.0f) * m;
    e.sqrt_abt = sqrtf(abt);
    e.oma = oma;
    e.scale = flow ? (sqrtf(abt) + sqrtf(1.0f - abt)) : sqrtf(1.0f + ve * ve);
    e.valid = step > 0.0f;
    return e;
}

// advance_time_overdamped, lanpaint.py:232-254 (general form, fp32 like the reference)
__device__ __forceinline__ float ou_general(float x, float tau, float a, float c, float d, float xi) {
    const float adt = a * tau;
    const float e = expf(-adt);
    float k, k2;
    if (fabsf(a) < 1e-8f) {
        k = tau;
        k2 = tau;
    } else {
        k = (-expm1f(-adt)) / a;
        k2 = (-expm1f(-2.0f * adt)) / (2.0f * a);
    }
    const float mean = e * x + k * c;
    const float var = (d * d) * k2;
    return mean + xi * sqrtf(fmaxf(var, 0.0f));
}

// ---- inner early stop on the device (earlystop.py:58-336, default metric) ----------------------------------
constexpr int kEsSums = 6;    // { sum w1 dA^2, sum w1, sum w2 dA^2, sum w2, sum w1 dB^2, sum w2 dB^2 }

// Cross-block accumulation of the six sums (round 3): every block adds its fp32 block sums -- as doubles, with
// device-scope hardware atomics -- into one of LP_ES_ACC_SLOTS slots of the accumulator set of its iteration
// (d.es_partials: LP_ES_ACC_SETS sets x LP_ES_ACC_SLOTS slots x 8 doubles; slot = block index mod LP_ES_ACC_SLOTS, so an
// address sees grid / 64 adds, not grid).  Whoever needs the totals of an iteration -- the next launch of a gated loop,
// lp_es_decide_kernel of a watched one -- sums the 64 slots in a fixed order, one slot per lane, with DPP moves: no
// per-block table to re-read (which limited the folded verdict to grids of <= 512 blocks) and no one-block kernel between
// two launches of a replayed loop at any size (C5: 21.5 -> see profiles/r03_microbench_es.log).  The sets rotate: launch i
// adds into set i % 3, reads set (i - 1) % 3 and clears set (i + 1) % 3, which nobody touches before launch i + 1.
// A double add of fp32 block sums is exact to ~1e-16 whatever order the atomics land in; the fp32 quotients the stop rule
// forms from the totals (earlystop.py:55) therefore agree between runs except when a total sits within 1e-16 of an fp32
// rounding boundary.
constexpr int kEsSlots = LP_ES_ACC_SLOTS, kEsSets = LP_ES_ACC_SETS;
static_assert(kEsSlots == kWave, "one accumulator slot per lane of the reducing wave");

__device__ __forceinline__ double* es_acc_set(const lp_step_desc& d, int iteration) {
    return d.es_partials + static_cast<size_t>(((iteration % kEsSets) + kEsSets) % kEsSets) * kEsSlots * 8;
}

// lane l holds slot l's six doubles; afterwards EVERY lane holds the six totals as floats (same bits in every wave of
// every block: fixed DPP order over the slots).  The 64 slot values are rounded to fp32 first and totalled in fp32 -- the
// precision the reference's own sums have (earlystop.py:52-55), and half the registers of a double tree inside a launch
// whose operand loads are all in flight at this point.
__device__ __forceinline__ void es_slot_total(const double (&v)[kEsSums], float (&tot)[kEsSums]) {
    float f[kEsSums];
#pragma unroll
    for (int k = 0; k < kEsSums; ++k) f[k] = static_cast<float>(v[k]);
    wave_sum_dpp(f);                                              // total valid in lane 63
#pragma unroll
    for (int k = 0; k < kEsSums; ++k)
        tot[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f[k]), kWave - 1));
}

__device__ __forceinline__ double clamp01(double v) { return v <= 0.0 ? 0.0 : (v >= 1.0 ? 1.0 : v); }

// New sigma call: patience counter, anchor and buffer rotation start over; the abt-scaled threshold
// (earlystop.py:21-29,105-113) is formed here from the same per-row abt the coefficient table is built from.
// (inlined, and only into the kernels that can carry a reset: a real call would make every instantiation pay the
// callee's register budget -- measured: 20 -> 108 VGPRs on the hot VEC = 1 kernel)
__device__ __forceinline__ void es_reset_state(const lp_step_desc& d, bool fold) {
    lp_es_state* es = d.es;
    const int n = fold ? (d.t_abt_stride ? d.rows : 1) : d.rows;
    const bool av = (d.flags & LP_FL_AV) != 0;       // two table rows per batch row; the mean of the BLENDED abt tensor weighs
    float sum = 0.0f;                                // them by the share of audio elements (lanpaint.py:70, earlystop.py:105-113)
    for (int r = 0; r < n; ++r) {
        if (av) {
            const float v = d.coef[static_cast<int64_t>(2 * r) * LP_COEF_STRIDE + LP_C_ABT];
            const float a = d.coef[static_cast<int64_t>(2 * r + 1) * LP_COEF_STRIDE + LP_C_ABT];
            sum = sum + (v * (1.0f - d.av_frac) + a * d.av_frac);
        } else {
            sum = sum + (fold ? d.t_abt[static_cast<int64_t>(r) * d.t_abt_stride]
                              : d.coef[static_cast<int64_t>(r) * LP_COEF_STRIDE + LP_C_ABT]);
        }
    }
    const double abt_val = static_cast<double>(sum / static_cast<float>(n));        // float(torch.mean(abt).item())
    const double a = clamp01(abt_val);
    const double thr_eff = d.es_threshold * clamp01(4.0 * a * (1.0 - a));
    // total_ran is never reset; a folded loop leaves it current in one slot of the pair only (LP_FL_ES_CLOSE)
    const int64_t total = es[0].total_ran > es[1].total_ran ? es[0].total_ran : es[1].total_ran;
    es->total_ran = total;
    es->stopped = 0; es->counter = 0; es->n_ran = 0;
    es->cur_slot = -1; es->anchor_slot = -1; es->write_slot = 0;
    es->enabled = thr_eff > 0.0 ? 1 : 0;
    es->seq_base = d.es_seq_base;
    es->threshold_eff = thr_eff;
    es->abt_val = abt_val;
    es[1] = es[0];                     // second slot of the ping-pong pair a folded loop alternates between
}

__device__ __forceinline__ void es_post_seq(double* host, int64_t seq) {
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<int64_t*>(host), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The stop rule of one iteration (earlystop.py:279-313) from the six sums; one thread.  `st` is the state as ONE
// block load at the top of the deciding kernel (field-by-field reads through the pointer cost a memory round trip
// each: 9 -> ~5 us per iteration at SDXL size), updated in registers and written back by the caller.
__device__ __forceinline__ void es_decide(const lp_step_desc& d, lp_es_state& st, const float (&sf)[kEsSums], bool have_prev,
                                          bool have_anchor, int i, bool post) {
    double* host = post ? d.es_host : nullptr;
    const bool has_ring = d.es_ring != nullptr;
    const float* s = sf;
    const double nan = __builtin_nan("");
    // fp32 quotients like the reference's (earlystop.py:55); only the threshold compare is in double
    double dist_in = static_cast<double>(s[0] / (s[1] + 1e-12f)), dist_ring = nan, dist_drift = nan;
    double dist = dist_in;
    if (have_prev && has_ring) {
        dist_ring = static_cast<double>(s[2] / (s[3] + 1e-12f));
        dist = dist_in > dist_ring ? dist_in : dist_ring;
    }
    const bool enabled = st.enabled != 0 && s[1] >= 1e-6f;          // earlystop.py:111-117
    int counter = st.counter, anchor = st.anchor_slot, stopped = 0;
    const int cur = st.write_slot;                                   // this iteration's x0s
    if (enabled) {
        const double thr = st.threshold_eff;
        if (dist <= thr) {                                           // drift guard, :295-306
            if (anchor < 0) {
                anchor = cur;
            } else if (have_anchor) {
                const double di = static_cast<double>(s[4] / (s[1] + 1e-12f));
                const double dr = has_ring ? static_cast<double>(s[5] / (s[3] + 1e-12f)) : nan;
                dist_drift = has_ring ? (di > dr ? di : dr) : di;
                dist = dist > dist_drift ? dist : dist_drift;
            }
        } else {
            anchor = -1;
        }
        if (dist <= thr) {
            counter += 1;
        } else {
            counter = 0;
            anchor = -1;
        }
        stopped = counter >= d.es_patience_eff ? 1 : 0;
    }
    st.counter = counter;
    st.anchor_slot = anchor;
    st.cur_slot = cur;
    int w = 0;
    while (w == cur || w == anchor) ++w;                             // the buffer that is neither
    st.write_slot = w;
    st.n_ran = i + 1;
    st.total_ran += 1;
    st.stopped = stopped;
    if (host) {
        double* rec = host + LP_ES_TRACE0 + 8 * i;
        rec[0] = dist; rec[1] = dist_in; rec[2] = dist_ring; rec[3] = dist_drift;
        rec[4] = static_cast<double>(counter); rec[5] = static_cast<double>(stopped); rec[6] = 0.0; rec[7] = 0.0;
        host[1] = static_cast<double>(i + 1); host[2] = static_cast<double>(stopped);
        host[3] = enabled ? 1.0 : 0.0; host[4] = st.threshold_eff; host[5] = st.abt_val;
        host[6] = static_cast<double>(st.total_ran);
        // a watched loop (eager) hears about every iteration; a gated one only when its last launch has run -- the
        // system-scope fence in front of the sequence word waits for the PCIe writes above (~3 us per launch)
        const bool gated = (d.flags & LP_FL_ES_GATED) != 0;
        if (!gated) es_post_seq(host, st.seq_base + i + 1);
        else if (i + 1 == d.es_n_steps) es_post_seq(host, st.seq_base + LP_ES_SEQ_DONE);
    }
}

// the fields a decision changes (the rest are per-call constants the reset wrote into both slots of the ping-pong pair)
__device__ __forceinline__ void es_store_dynamic(lp_es_state* dst, const lp_es_state& st) {
    dst->stopped = st.stopped; dst->counter = st.counter; dst->n_ran = st.n_ran; dst->cur_slot = st.cur_slot;
    dst->anchor_slot = st.anchor_slot; dst->write_slot = st.write_slot; dst->total_ran = st.total_ran;
}

// ---- profiling build (-DLP_SHADER_CLOCK; scripts/shader_clock.py): where a launch spends its time on the chip ---------
// Stamps of the shader clock (s_memtime) at the points LP_CLK_STAMPS names, by thread 0 of the first and the last
// block.  Stamp 3 waits for every outstanding load first, so it perturbs what follows a little; the release build
// compiles all of it away.
#ifdef LP_SHADER_CLOCK
__device__ __forceinline__ uint64_t clk_now() {
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t clk_after_loads() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t clk_real() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define LP_CLK_DECL                                                                                                  \
    uint64_t clk_t[LP_CLK_STAMPS] = {}, clk_r0 = 0;                                                                  \
    const bool clk_on = d_arg.clk_out && threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1); \
    if (clk_on) { clk_t[0] = clk_now(); clk_r0 = clk_real(); }
#define LP_CLK(k) if (clk_on) clk_t[k] = clk_now();
#define LP_CLK_LOADS(k) if (clk_on) clk_t[k] = clk_after_loads();
#define LP_CLK_FLUSH                                                                                                 \
    if (clk_on) {                                                                                                    \
        double* clk_o = d_arg.clk_out + (blockIdx.x == 0 ? 0 : 16);                                                  \
        for (int k = 0; k < LP_CLK_STAMPS; ++k) clk_o[k] = clk_t[k] ? static_cast<double>(clk_t[k] - clk_t[0]) : 0.0; \
        clk_o[15] = static_cast<double>(clk_real() - clk_r0);                                                        \
        clk_o[14] = static_cast<double>(clk_r0 & 0xffffffffffffull);      /* absolute 100 MHz time of the block's entry */ \
    }
#else
#define LP_CLK_DECL
#define LP_CLK(k)
#define LP_CLK_LOADS(k)
#define LP_CLK_FLUSH
#endif

// Threads per block, fixed: blockDim.x read at run time is one more scalar-load round trip (a hidden kernel argument) in
// front of the first operand load, and the early-stop reductions are written for four waves.
constexpr int kBlock = 256;

constexpr uint32_t kPost = LP_PH_POST_FIRST | LP_PH_POST_STEADY;
constexpr uint32_t kTouchXt = LP_PH_REPLACE | kPost | LP_PH_PRE_HALF;

// MODE: 0 = per-row coefficient table, any mask (soft values take the per-element branch);
//       1 = per-element times (AV packs);  2 = per-row table + mask known to be hard 0/1 (bit-packed):
//       the general branch -- divisions, expm1, the soft-mask blend -- is compiled out of the hot kernel.
enum : int { MODE_ROW = 0, MODE_PER_EL = 1, MODE_HARD = 2 };

// X0W: storage width of x0 / x0_big known at compile time (4 = fp32, 2 = bf16/fp16) or 0 = read the flag at
// run time.  With a run-time dtype branch around the loads the two sides share registers and the compiler
// has to put an s_waitcnt vmcnt(0) at their join -- ahead of the remaining loads -- so the hot phase
// combinations are instantiated per width; likewise they only take fp32 (MODE_ROW) or bit-packed (MODE_HARD)
// masks, a uint8 mask goes through the PH = 0 kernel.
// RNG: in-kernel generator fixed at compile time in the hot variants (0 = Philox2x32 pair, 1 = torch's randn
// stream reproduced exactly) or 2 = read d.rng_kind at run time.
// ST (with VEC = 4, RNG = 1, n_el > bg): the launch is laid out like ATen's random kernel -- a block is 256 of its threads, each
// evaluating ONE Philox4x32 block and its two Box-Muller pairs per draw for the four elements it serves in torch (idx, idx + bg,
// idx + 2 bg, idx + 3 bg of one round of its grid-stride loop) -- and the values are transposed through LDS to the lanes that
// stream those elements 16 bytes at a time (wave w of the block: the 256 consecutive elements of slot w).  LP_RNG_TORCH on video
// latents: 26 us with one block per element, 14 -> 11.5 us with four 4-byte streams per lane (rounds 2-4), now the Philox2x32
// kernels' access pattern.  Batches and tensors of several rounds: one blockIdx.y per (round, batch row crossing it), each
// block working on the elements that belong to both with that row's coefficients.
// ES: the POST phase also evaluates the inner early-stop rule (LP_FL_ES): 1 = per-block sums for the decision kernel
// that follows, 2 = the launch first applies the verdict of the iteration before itself (gated loops, small grids).
template <int VEC, int MODE, uint32_t PH, int X0W, int RNG, bool ST = false, int ES = 0>
// (the folded early-stop kernels at 16 B per lane hold ~106 SGPRs -- state, two coefficient regions, verdict -- and gfx950 grants a
// wave at most 96 when eight share a SIMD: asking for 8 waves per EU makes the compiler keep the overflow in lanes of a spare
// VGPR instead; at 7 waves the video latent's 2048 blocks, 8 per CU, need a second round)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((ES == 2 && VEC == 4 && PH != 0) ? 8 : 1))) void lp_step_kernel(void* a0, void* a1, const void* a2, const void* a3, const void* a4,
                                                       const void* a5, int32_t a_el_per_row, uint32_t a_flags,
                                                       const lp_step_desc d_arg) {
    constexpr bool PER_EL = MODE == MODE_PER_EL;
    constexpr bool HARD = MODE == MODE_HARD;
    // Kernel-argument preload (gfx940+; build flag -amdgpu-kernarg-preload-count): the command processor puts the
    // first 14 dwords of the argument segment into SGPRs before the wave starts.  The argument segment is written per
    // launch, so reading the descriptor from it is a cold scalar load -- one DRAM round trip, ~0.3 us of the ~1 us a
    // think step spends on the chip at SDXL size (scripts/shader_clock.py) -- and whatever is needed to ISSUE the
    // first-touch loads of the step should not sit behind it.  Fourteen dwords hold six pointers, the row length and
    // the flags (LP_STEP_ARGS below): x_t, C and the two heads always; on the latency-bound sizes (VEC = 1) the
    // coefficient table and the replayed graph's generator state, whose loads start the two longest dependent chains
    // (state -> Philox rounds; table -> arithmetic), on the streaming sizes y and the mask (the region-aware decision
    // waits for the mask).  The rest of the descriptor arrives while those loads fly.  (Older firmware runs the
    // compiler's compatibility prologue, which loads the same SGPRs itself.)
    constexpr bool SMALL = VEC == 1;
    // half-width backbone outputs (and a half-width x_in) of a streaming launch travel 16 bytes per lane PAIR (lp_common.h)
    constexpr bool PAIR = X0W == 2 && VEC == 4 && !ST;
    static_assert(!PAIR || ES == 0, "pair accesses: every lane that has not left the kernel is active");
    LP_CLK_DECL
    lp_step_desc d = d_arg;
    d.x_t = static_cast<float*>(a0); d.C = static_cast<float*>(a1);
    if constexpr ((PH & LP_PH_REPLACE) != 0) {       // replace launch: its four input streams
        d.x = static_cast<const float*>(a2);
        d.known = d.noise = static_cast<const float*>(a3);
        d.y = static_cast<const float*>(a4); d.mask = a5;
    } else {
        d.x0 = a2; d.x0_big = a3;
        if constexpr (SMALL) {
            d.coef = static_cast<const float*>(a4); d.rng_offset_ptr = static_cast<const uint64_t*>(a5);
        } else {
            d.y = static_cast<const float*>(a4); d.mask = a5;
        }
    }
    d.el_per_row = a_el_per_row; d.flags = a_flags;
    static_assert(!ES || (!ST && !PER_EL && (PH & LP_PH_REPLACE) == 0), "early stop: think-step launches of the row-table modes");
    constexpr bool es_fold = ES == 2;               // the verdict of iteration i - 1 rides in launch i (small grids)
    int row = blockIdx.y;
    // ST: blockIdx.y enumerates (round of ATen's grid-stride loop, batch row crossing that round); the block works on the
    // part [st_lo, st_hi) of the flat tensor that belongs to both, with that row's coefficients
    int64_t st_base = 0, st_lo = 0, st_hi = 0;
    uint32_t st_round = 0;
    if constexpr (ST) {
        const int64_t span = 4 * static_cast<int64_t>(d.rng_bg), epr = d.el_per_row;
        const uint32_t rounds = d.rng_inc >> 2, per_round = gridDim.y / rounds;
        st_round = blockIdx.y / per_round;
        st_base = static_cast<int64_t>(st_round) * span;
        row = static_cast<int>(st_base / epr) + static_cast<int>(blockIdx.y - st_round * per_round);
        const int64_t row_lo = static_cast<int64_t>(row) * epr, row_hi = row_lo + epr;
        st_lo = row_lo > st_base ? row_lo : st_base;
        st_hi = row_hi < st_base + span ? row_hi : st_base + span;
        if (row >= d.rows || st_lo >= st_hi) return;
        // a row that only clips a corner of the round touches a few lanes' slots: blocks without any leave at once
        const int64_t i0 = st_base + static_cast<int64_t>(blockIdx.x) * kBlock, i1 = i0 + kBlock;
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t s0 = i0 + k * static_cast<int64_t>(d.rng_bg), s1 = i1 + k * static_cast<int64_t>(d.rng_bg);
            any = any || (s0 < st_hi && s1 > st_lo);
        }
        if (!any) return;
    }
    const uint32_t fl = d.flags;
    bool es_gated = false, es_idle = false;
    int es_prev = -1, es_anchor = -1, es_write = 0;
    bool es_keeper = false;
    // folded decision: what its prologue loads (state fields, the first wave's slot of the previous iteration's accumulator set)
    lp_es_state es_lite;
    uint2 es_words[8];
    double es_fv[kEsSums];           // (written and read by the block's first wave only)
    if constexpr (ES) {
        es_gated = (fl & LP_FL_ES_GATED) != 0;
        if constexpr (es_fold) {
            // Folded decision: launch i first applies the stop rule of iteration i - 1 -- every block totals the
            // previous iteration's accumulator set itself (64 slots, same fixed order -> same bits in every block) and
            // steps the state in registers; block 0 stores it.  The state ping-pongs between two slots and the sums
            // rotate through three sets, so nobody reads what a neighbour block of the same launch writes.
            // This replaces a decision kernel between every two launches: ~5 us per iteration at SDXL size.
            // Everything the verdict needs is only LOADED here; it is formed after the operand loads of the step have
            // been issued as well (below, behind the Philox rounds), so the launch pays one memory round trip where a
            // verdict-first order pays three (state -> sums -> the x0s buffers the verdict selects).
            const int rd = (d.es_index + 1) & 1;
            const lp_es_state* sp = d.es + rd;
            // The state's 64 leading bytes through VECTOR loads of one address (every lane the same words, made
            // wave-uniform again with readfirstlane where the verdict is formed): a scalar load would be waited for by
            // the next s_waitcnt lgkmcnt(0) -- the kernarg reads in front of the operand loads -- i.e. put its whole
            // round trip back in front of them; vector loads retire in order behind nothing.
            // (Streaming sizes, VEC = 4: registers decide -- at 98 VGPRs the launch ran 4 waves per SIMD and the video latent's
            // 2048 blocks took two rounds, 17.9 us against 9.5 for the plain launch.  There the state comes through SCALAR loads
            // (the launch waits for it anyway, below) and the accumulator slot is fetched where the verdict is formed.)
            if constexpr (VEC == 4) {
                const uint2* sv = reinterpret_cast<const uint2*>(sp);    // wave-uniform address: s_load
#pragma unroll
                for (int k = 0; k < 8; ++k) es_words[k] = sv[k];
            } else {
                uint32_t lane_zero;
                asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));          // opaque: keeps the address in a VGPR
                const uint2* sv = reinterpret_cast<const uint2*>(sp) + lane_zero;
#pragma unroll
                for (int k = 0; k < 8; ++k) es_words[k] = sv[k];
            }
            es_keeper = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
            // lane l of the block's FIRST wave: slot l of the previous iteration's accumulator set (48 B); the other three
            // waves get the totals through LDS (a launch of 4 096 waves each reading the 3 KB set moved more bytes out of
            // L2 than the SDXL-batch operands themselves)
            if constexpr (VEC != 4) {
                if (threadIdx.x < kWave) {
                    const double* src = es_acc_set(d, d.es_index - 1) + static_cast<size_t>(threadIdx.x) * 8;
#pragma unroll
                    for (int k = 0; k < kEsSums; ++k) es_fv[k] = src[k];
                }
            }
        } else {                                     // wave-uniform scalar loads of the device-side stop state
            es_prev = d.es->cur_slot; es_anchor = d.es->anchor_slot; es_write = d.es->write_slot;
            if (es_gated && d.es->stopped != 0) {    // the loop has stopped: only re-emit x_in from the committed x_t
                const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
                if (g >= d.el_per_row / VEC) return;
                const int64_t i = static_cast<int64_t>(row) * d.el_per_row + g * VEC;
                float xt[VEC], xo[VEC];
                load_f32<VEC>(d.x_t, i, xt);
                if constexpr (PH == 0 && MODE == MODE_ROW && !ST) {
                    if (fl & LP_FL_AV) {             // AV pack: every element on its own stream's table row (2 r: video, 2 r + 1: audio)
                        const uint32_t word = static_cast<const uint32_t*>(d.av_bits)[i >> 5];
                        const uint32_t nib = word >> (static_cast<uint32_t>(i) & 31u);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            const float sc = d.coef[static_cast<int64_t>(2 * row + static_cast<int>((nib >> k) & 1u)) * LP_COEF_STRIDE + LP_C_SCALE];
                            xo[k] = (fl & LP_FL_FLOW) ? xt[k] / sc : xt[k] * sc;
                        }
                        store_any<VEC>(d.x_in, xin_dtype(fl), i, xo);
                        return;
                    }
                }
                const float sc = load_row(d.coef, row).scale;
#pragma unroll
                for (int k = 0; k < VEC; ++k) xo[k] = (fl & LP_FL_FLOW) ? xt[k] / sc : xt[k] * sc;
                store_any<VEC>(d.x_in, xin_dtype(fl), i, xo);
                return;
            }
        }
    }
    const uint32_t ph = PH ? PH : d.phases;          // compile-time for the hot combinations
    const bool flow = fl & LP_FL_FLOW;
    const bool post = ph & kPost;
    // The phase-specialised kernels (PH != 0: the launches a think loop repeats) never see the two rare forms -- x0s handed in
    // (LP_FL_X0S_GIVEN: the public langevin_dynamics) and host-supplied noise (xi_post / xi_pre: recorded streams, explicit
    // torch.randn_like draws) -- lp_step routes those to the run-time-phase kernels (launch_phase), so the four selects on
    // `given` and the eight between in-kernel and host noise leave the hot code (round 5: 12 VALU instructions per lane of
    // a launch that is VALU-bound at streaming sizes)
    // (Philox kernels without early stop only: the early-stop kernels sit at their register limits -- the changed schedule sent one of
    // them to scratch memory -- and the torch-stream kernels, whose two Philox4x32 blocks per lane the scheduler then interleaves
    // further, went from 52-64 to 65-74 VGPRs, i.e. below 8 waves per SIMD)
    constexpr bool HOT = PH != 0 && ES == 0 && RNG == 0;
    const bool given = HOT ? false : (fl & LP_FL_X0S_GIVEN) != 0;
    const int x0dt = X0W == 4 ? static_cast<int>(DT_F32) : x0_dtype(fl), xindt = xin_dtype(fl);
    const int64_t groups = ST ? static_cast<int64_t>(d.rng_bg) : d.el_per_row / VEC;
    const int64_t row_base = static_cast<int64_t>(row) * d.el_per_row;
    const float lam = d.lambda, opl = d.one_plus_lambda;

    RowCoef rc;
    bool av_mixed = false;          // LP_FL_AV: the wave straddles the video / audio seam of its row
    uint32_t av_nib = 0u;           // LP_FL_AV: this lane's indicator bits (VEC of them)
    const bool fold_coeffs = (ph & LP_PH_COEFFS) != 0;       // compile-time for PH != 0
    if (fold_coeffs) {
        // lp_coeffs folded into the replace launch: every block derives the two row scalars its own work needs
        // (same expressions as the table's), lanes 0..3 of the row's first block build the table for the
        // launches that follow
        float abt_f, ve_f, rs_f, tm_sig = 0.0f;
        constexpr bool fold_sigma = (PH & LP_PH_SIGMA) != 0;
        if constexpr (fold_sigma) {
            // LP_PH_SIGMA: the times of this row straight from sigma (what lp_sigma_times would have written to t_ve / t_abt /
            // t_model; the replace sigma IS sigma); the first wave of the first block also does that kernel's part --
            // the rule's two scalars, the rule against the speculated count, the mailbox
            float ft;
            rs_f = d.sg_sigma[row];
            sigma_to_times(rs_f, flow, ve_f, abt_f, ft);
            tm_sig = flow ? ft : ve_f;
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kWave) {      // (the block's first wave, all 64 lanes alive here)
                const SigmaRule rule{d.sg_n_steps, d.sg_early_stop, d.sg_total_steps, d.sg_guess, d.sg_min_step_frac, d.sg_valid_out};
                sigma_rows_and_rule(d.sg_sigma, d.rows, d.sg_schedule, d.sg_schedule_len, flow, d.sg_times_out, d.sg_scalars_out,
                                    d.sg_seq_out, d.sg_seq, rule);
            }
        } else {
            abt_f = d.t_abt[static_cast<int64_t>(row) * d.t_abt_stride];
            ve_f = d.t_ve ? d.t_ve[static_cast<int64_t>(row) * d.t_ve_stride] : 0.0f;
            rs_f = d.t_rsig ? d.t_rsig[static_cast<int64_t>(row) * d.t_rsig_stride] : 0.0f;
        }
        rc.scale = row_scale(flow, abt_f, ve_f);
        rc.rscale = 1.0f / rc.scale;        // (what the table's LP_C_RSCALE holds: the emit of this launch divides by it, see EMIT)
        rc.rsigma = rs_f;
        if (blockIdx.x == 0 && threadIdx.x < 4) {
            lp_hyper h;
            h.lambda = d.lambda; h.beta = d.beta; h.step_size = d.step_size; h.min_step_frac = d.min_step_frac;
            h.is_flow = flow ? 1 : 0; h.one_plus_lambda = d.one_plus_lambda;
            const float tm_f = fold_sigma ? tm_sig : (d.t_model ? d.t_model[static_cast<int64_t>(row) * d.t_model_stride] : 0.0f);
            const float step = d.step_size * fmaxf(1.0f - abt_f, d.min_step_frac);                 // lanpaint.py:81
            coeffs_lane(h, abt_f, ve_f, rs_f, tm_f, step, (threadIdx.x >> 1) & 1, threadIdx.x & 1,
                        d.coef_out + static_cast<int64_t>(row) * LP_COEF_STRIDE);
            if (row == 0 && threadIdx.x == 0 && d.rng_state_out) {     // generator state for the replayed launches
                d.rng_state_out[0] = d.rng_state_val[0];
                d.rng_state_out[1] = d.rng_state_val[1];
            }
        }
    } else {
        if constexpr (PH == 0) {
            // a replace launch that does not build the table (AV packs: lp_coeffs built both rows) still publishes the state
            if ((ph & LP_PH_REPLACE) && d.rng_state_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
                d.rng_state_out[0] = d.rng_state_val[0];
                d.rng_state_out[1] = d.rng_state_val[1];
            }
        }
        if constexpr (!PER_EL) {
            if constexpr (SMALL && PH != 0) {
                static_assert(VEC == 1, "load_row_early leaves rc.rscale poisoned: the div_shared paths (VEC == 4) must use load_row");
                rc = load_row_early(d.coef, row);
            } else {
                // AV packs (LP_FL_AV, see below): which of the batch row's two table rows this wave runs on is decided HERE, so that
                // `rc` has ONE load site (a RowCoef assigned on several paths and merged ends up in scratch memory)
                int crow = row;
                if constexpr (PH == 0 && MODE == MODE_ROW && !ST) {
                    if (fl & LP_FL_AV) {
                        const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
                        const bool in_row = g0 < groups;
                        const int64_t i0 = row_base + (in_row ? g0 : groups - 1) * VEC;
                        const uint32_t word = static_cast<const uint32_t*>(d.av_bits)[i0 >> 5];
                        av_nib = (word >> (static_cast<uint32_t>(i0) & 31u)) & ((1u << VEC) - 1u);
                        const bool any_audio = __ballot(in_row && av_nib != 0u) != 0ull;
                        const bool any_video = __ballot(in_row && av_nib != ((1u << VEC) - 1u)) != 0ull;
                        av_mixed = any_audio && any_video;
                        crow = 2 * row + ((any_audio && !any_video) ? 1 : 0);
                    }
                }
                rc = load_row(d.coef, crow);
            }
        }
    }

    // device-side generator state of a replayed graph: with the pointer preloaded its load starts here
    uint64_t rng_w0 = 0, rng_w1 = 0;
    bool rng_have = false;
    if constexpr (SMALL) {
        if (d.rng_offset_ptr) {
            rng_w0 = d.rng_offset_ptr[0];
            rng_w1 = d.rng_offset_ptr[1];
            rng_have = true;
        }
    }

    // one group per lane and no grid-stride loop: the launch covers the row (see launch()), which keeps every
    // address in this straight-line body a kernarg pointer + one offset and lets the scalar loads (coefficient
    // row, replayed-graph RNG counter) fly together with the vector loads instead of ahead of a loop
    const int64_t g_raw = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    // ST (round 5): the block is 256 ATen threads -- [256 b, 256 b + 256) of the generator's launch -- whose values fall on FOUR
    // runs of 256 consecutive elements, bg apart (slot k of thread j: element base + k bg + 256 b + j).  Wave w of the block
    // takes run w, 16 bytes per lane like every other streaming launch; the values reach it through LDS (below).
    int64_t st_i = 0;
    if constexpr (ST) st_i = st_base + static_cast<int64_t>(threadIdx.x >> 6) * static_cast<int64_t>(d.rng_bg) +
                             static_cast<int64_t>(blockIdx.x) * kBlock + static_cast<int64_t>(threadIdx.x & 63) * VEC;
    // ES: every lane stays for the block reduction; a lane past the end recomputes the last group and stores nothing
    // ST: every lane stays for the noise exchange; a lane outside the row's part of the round works on the part's first
    // group and stores nothing (segment bounds are multiples of four elements: a lane's group never straddles one)
    const bool active = ST ? (st_i >= st_lo && st_i < st_hi) : (g_raw < groups);
    if constexpr (!ES && !ST) {
        if (!active) return;
    }
    const int64_t g = active ? g_raw : groups - 1;
    float es_p[kEsSums] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool es_zero_wave = false;           // the wave added nothing to its sums (mask-uniform, all known): no reduction tree needed
    {
        const int64_t i = ST ? (active ? st_i : st_lo) : row_base + g * VEC;

        // ---- AV packs (LP_FL_AV; MiniMax-H3 flat audio / video packs, lanpaint.py:60-74) ---------------------------------------
        // The reference blends per-stream times with a full-size 0/1 indicator (VE * (1 - ai) + VE_a * ai, ...): every element
        // sits on exactly one of TWO per-row time sets.  The coefficient table then holds two rows per batch row (2 r: video,
        // 2 r + 1: audio) and the indicator travels as bits; a wave whose elements all belong to one stream -- all but the
        // wave that straddles the video / audio seam of a row -- simply takes that stream's row and runs the ordinary table
        // path.  The straddling wave uses the reference's per-element formulas on the two rows' fp32 fields (elem_from_row).
        // Run-time-phase row-table kernels only; the phase-specialised kernels never see the flag (lp_step routes it).
        constexpr bool AVK = PH == 0 && MODE == MODE_ROW && !ST;
        // What an element of a STRADDLING wave takes from its own row: read per lane from the table (one wave per batch row does
        // this; a second RowCoef held next to `rc` and selected per lane would put both structs into scratch memory).
        // The table words are read as relaxed wavefront-scope ATOMIC loads -- plain global_load instructions -- because a plain C++
        // load here gives the optimiser `x = mixed ? table[f] : rc.f`, which it rewrites into ONE load through a pointer chosen
        // between the table and &rc.f: `rc` then has its address taken and the whole row lives in scratch memory (132 B, found in the
        // ISA of the three run-time row kernels; atomic loads are never merged that way).
        auto av_row = [&](int k) __attribute__((always_inline)) -> const float* {
            return d.coef + static_cast<int64_t>(2 * row + static_cast<int>((av_nib >> k) & 1u)) * LP_COEF_STRIDE;
        };
        auto av_ld = [&](const float* c, int f) __attribute__((always_inline)) -> float {
            return __hip_atomic_load(c + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        };
        auto row_scale_of = [&](int k) __attribute__((always_inline)) -> float {
            if constexpr (AVK) {
                if (av_mixed) return av_ld(av_row(k), LP_C_SCALE);
            }
            return rc.scale;
        };
        auto row_rsigma_of = [&](int k) __attribute__((always_inline)) -> float {
            if constexpr (AVK) {
                if (av_mixed) return av_ld(av_row(k), LP_C_RSIGMA);
            }
            return rc.rsigma;
        };
        auto elem_of = [&](int k, float mk) __attribute__((always_inline)) -> ElemCoef {
            // elem_from_row on the element's row; a straddling wave reads the row's fp32 fields per lane from the table (scalars
            // merged field by field -- two ElemCoef structs built on two paths and merged would again live in scratch memory)
            float ax = rc.ax, ay = rc.ay, dx = rc.dx, dy = rc.dy, dtx = rc.dtx, dty = rc.dty, sq = rc.sqrt_abt, oma = rc.oma,
                  sc = rc.scale, valid = rc.valid;
            if constexpr (AVK) {
                if (av_mixed) {
                    const float* c = av_row(k);
                    ax = av_ld(c, LP_C_AX); ay = av_ld(c, LP_C_AY); dx = av_ld(c, LP_C_DX); dy = av_ld(c, LP_C_DY);
                    dtx = av_ld(c, LP_C_DTX); dty = av_ld(c, LP_C_DTY); sq = av_ld(c, LP_C_SQRT_ABT); oma = av_ld(c, LP_C_OMA);
                    sc = av_ld(c, LP_C_SCALE); valid = av_ld(c, LP_C_VALID);
                }
            }
            const float om = 1.0f - mk;
            ElemCoef e;
            e.a = ax * om + ay * mk;
            e.d = dx * om + dy * mk;
            e.dt = dtx * om + dty * mk;
            e.sqrt_abt = sq;
            e.oma = oma;
            e.scale = sc;
            e.valid = valid != 0.0f;
            return e;
        };

        // (every lambda of this body is always_inline: one that stays a call takes the arrays it captures -- xt, cv, x0s -- by address,
        // which sends them to scratch memory; that happened to the first-iteration kernels and cost 24-42 %, found by the
        // same-box A/B against the round-3 library, scripts/r04_ab_more.sh)
        // The verdict of iteration i - 1, formed inside launch i (folded loops): state -> totals of the accumulator slots ->
        // stop rule.  Latency-bound sizes (VEC = 1) call it BEHIND the operand loads and the Philox rounds (one memory round
        // trip instead of three); streaming sizes (VEC = 4) call it FIRST, before any operand load is issued: its twenty-odd
        // registers then never overlap the operands', the launch stays at 8 waves per SIMD, and the history loads know which
        // buffers they need (round 3: verdict behind the loads, 98 VGPRs, 4 waves per SIMD -- the video latent's 2048 blocks ran
        // in two rounds, 17.9 us against 9.5 for the plain launch).
        auto form_verdict = [&]() __attribute__((always_inline)) {
                const int it = d.es_index;
                {
                    const auto u = [&](int w) { return __builtin_amdgcn_readfirstlane(static_cast<int>((w & 1) ? es_words[w >> 1].y : es_words[w >> 1].x)); };
                    const auto u64 = [&](int w) { return (static_cast<uint64_t>(static_cast<uint32_t>(u(w + 1))) << 32) | static_cast<uint32_t>(u(w)); };
                    static_assert(offsetof(lp_es_state, enabled) == 28 && offsetof(lp_es_state, seq_base) == 32 &&
                                  offsetof(lp_es_state, abt_val) == 56, "lp_es_state layout");
                    es_lite.stopped = u(0); es_lite.counter = u(1); es_lite.n_ran = u(2); es_lite.cur_slot = u(3);
                    es_lite.anchor_slot = u(4); es_lite.write_slot = u(5); es_lite.enabled = u(7);
                    es_lite.seq_base = static_cast<int64_t>(u64(8)); es_lite.total_ran = static_cast<int64_t>(u64(10));
                    es_lite.threshold_eff = __longlong_as_double(static_cast<long long>(u64(12)));
                    es_lite.abt_val = __longlong_as_double(static_cast<long long>(u64(14)));
                }
                int v_stopped = 0, v_cur = -1, v_anchor = -1, v_write = 0;
                bool v_have = false;
                if (it > 0 && es_lite.stopped == 0) {        // (block-uniform: every wave takes the barrier)
                    // The block's FIRST wave totals the accumulator slots and applies the rule; the other three only take the
                    // outcome -- stop flag and the three buffer roles -- from LDS.  (Round 3 had every wave redo the rule from
                    // the totals: ~100 VALU instructions per wave of a launch that is VALU-bound at streaming sizes.)
                    __shared__ int fold_out[4];
                    if (threadIdx.x < kWave) {
                        float tot[kEsSums];
                        if constexpr (VEC == 4) {        // (fetched here, not at the top: twelve registers less across the loads)
                            const double* src = es_acc_set(d, d.es_index - 1) + static_cast<size_t>(threadIdx.x) * 8;
#pragma unroll
                            for (int k = 0; k < kEsSums; ++k) es_fv[k] = src[k];
                        }
                        if constexpr (VEC != 4) {
                            // (the slot was fetched at the top of the kernel; the empty asm pins its first USE here, behind the operand
                            // loads and the Philox rounds.  Without it the optimiser folds the double -> float conversion of
                            // es_slot_total into the block that loads the slot, and the first wave of every block waits a whole memory
                            // round trip for its accumulator slot before it issues a single operand load: 0.5 us per launch)
#pragma unroll
                            for (int k = 0; k < kEsSums; ++k) asm volatile("" : "+v"(es_fv[k]));
                        }
                        es_slot_total(es_fv, tot);
                        const bool hp = es_lite.cur_slot >= 0, ha = es_lite.anchor_slot >= 0;
                        es_decide(d, es_lite, tot, hp, ha, it - 1, es_keeper);
                        if (threadIdx.x == 0) {
                            fold_out[0] = es_lite.stopped; fold_out[1] = es_lite.cur_slot;
                            fold_out[2] = es_lite.anchor_slot; fold_out[3] = es_lite.write_slot;
                        }
                    }
                    __syncthreads();
                    // (every wave, the first included, takes the outcome from LDS into plain scalars: writing it back into `es_lite`
                    // on one path only would merge the struct across paths and send it to scratch memory)
                    v_stopped = fold_out[0]; v_cur = fold_out[1]; v_anchor = fold_out[2]; v_write = fold_out[3];
                    v_have = true;
                }
                if ((fl & LP_FL_ES_CLOSE) && it + 1 == d.es_n_steps) {
                    // last launch of the loop and no closing decision kernel (LP_FL_ES_CLOSE): unless the loop has
                    // stopped, this launch commits iteration `it` -- account it now and tell the host the call is done
                    if (es_lite.stopped == 0) {
                        es_lite.n_ran = it + 1;
                        es_lite.total_ran += 1;
                    }
                    if (es_keeper) {
                        // (this launch's own slot only: other blocks may still be reading the other one; the next
                        // reset takes the running count from whichever slot is ahead)
                        es_store_dynamic(d.es + (it & 1), es_lite);
                        if (double* host = d.es_host) {
                            host[1] = static_cast<double>(es_lite.n_ran); host[2] = static_cast<double>(es_lite.stopped);
                            host[3] = es_lite.enabled ? 1.0 : 0.0; host[4] = es_lite.threshold_eff; host[5] = es_lite.abt_val;
                            host[6] = static_cast<double>(es_lite.total_ran);
                            es_post_seq(host, es_lite.seq_base + LP_ES_SEQ_DONE);
                        }
                    }
                } else if (es_keeper) {
                    es_store_dynamic(d.es + (it & 1), es_lite);
                }
                es_prev = v_have ? v_cur : es_lite.cur_slot; es_anchor = v_have ? v_anchor : es_lite.anchor_slot;
                es_write = v_have ? v_write : es_lite.write_slot;
                es_idle = (v_have ? v_stopped : es_lite.stopped) != 0;      // stopped: only re-emit x_in from the committed x_t
        };
        if constexpr (ES == 2 && VEC == 4) form_verdict();
        // ---- issue every load of this launch before any arithmetic ---------------------
        float m[VEC], xt[VEC], yv[VEC], cv[VEC], x0[VEC], x0b[VEC], xi_a[VEC], xi_b[VEC], corr[VEC];
        float xv[VEC], kn[VEC], nv[VEC], rs[VEC], abt_e[VEC], ve_e[VEC];
        Raw<VEC> m_raw, x0_raw, x0b_raw;
        // HARD: bit-packed by dispatch; hot MODE_ROW variants: fp32 (uint8 masks are routed to PH = 0)
        const uint32_t mfl = HARD ? static_cast<uint32_t>(LP_FL_MASK_BITS) : (PH != 0 ? (fl & ~LP_FL_MASK_U8) : fl);
        if constexpr (!SMALL) load_mask_raw<VEC>(d.mask, mfl, i, m_raw);    // (SMALL: its pointer is not preloaded, below)
        // Region-aware streams (bit-packed mask, streaming sizes): an inpaint element needs head 0 only, a known one
        // needs head 1 and y only (lanpaint.py:182-184 with m in {0,1}).  Skipping a stream per LANE saves nothing (the
        // wave still touches the cache lines); skipping it for the whole WAVE does: 256 consecutive elements whose
        // mask bits are all 0 drop x0_BIG and y (36 -> 28 B/element), all 1 drop x0 (-> 32 B).  The decision is a
        // ballot over the mask word, i.e. wave-uniform (scalar branches); the mask load is the oldest in flight, so
        // only it is waited for -- x_t and C are issued before the decision, the conditional streams right after it.
        // In a mixed wave (mask edges, fine-grained masks) each LANE still leaves out the stream none of its four elements
        // reads: the loads run under the lanes' predicate, and a 128-byte line no active lane touches is not fetched
        // (eight lanes = 32 consecutive elements of one region; 50 % box on 1.2 GB: -7 %, disc -4 %).
        constexpr bool RA = HARD && VEC == 4 && (PH & kPost) != 0;
        // Wave-uniform ARITHMETIC (round 4; the streaming kernel turned out VALU-bound, not bandwidth-bound: 538 VALU
        // instructions per wave keep the SIMDs > 80 % busy, profiles/r04_sq_*.md): a wave whose 256 mask bits are all 0 or
        // all 1 takes its region's coefficients from SGPRs -- no mask decode, no per-element selects -- through the very
        // expressions of the per-element path (same operations on the same values: bit-identical results).
        constexpr bool UNI = HARD && VEC == 4 && (PH & (kPost | LP_PH_PRE_HALF)) != 0 && (PH & LP_PH_REPLACE) == 0;
        constexpr bool MIXSEL = UNI && ES == 0;       // mixed waves of these kernels: both regions computed, results selected (below)
        int uni = -1;                                // 0: every element inpaint, 1: every element known, -1: mixed
        bool need_x0 = true, need_known = true;
        bool lane_x0 = true, lane_known = true;
        bool pair_x0 = true, pair_known = true;       // PAIR: either lane of the pair reads the stream
        if constexpr (PER_EL) {
            load_f32<VEC>(d.abt_el, i, abt_e);
            if (!flow) load_f32<VEC>(d.ve_el, i, ve_e);
        }
        if (ph & LP_PH_REPLACE) {
            load_f32<VEC>(d.x, i, xv);
            // region-aware like the POST streams below: a wave of 256 inpaint elements (mask bits all 0) keeps its x and reads
            // neither the noise nor the known latent -- 12 -> 4 B / element read there (streaming kernels, bit-packed mask)
            bool need_kn = true;
            if constexpr (HARD && VEC == 4 && (PH & LP_PH_REPLACE) != 0) {
                if (!(fl & LP_FL_NO_REGION_SKIP)) {
                    const uint32_t nib = (m_raw.w[0] >> (static_cast<uint32_t>(i) & 31u)) & 0xFu;
                    need_kn = __ballot(nib != 0u) != 0ull;
                }
            }
            if (!need_kn) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) kn[k] = 0.0f, nv[k] = 0.0f, yv[k] = 0.0f;      // (times m = 0 below)
            } else if (d.replace_kind == LP_REPLACE_KNOWN) {
                load_f32<VEC>(d.known, i, kn);
            } else {
                load_f32<VEC>(d.noise, i, nv);
                load_f32<VEC>(d.y, i, yv);
                if constexpr (PER_EL) load_f32<VEC>(d.rsig_el, i, rs);
            }
        } else {
            // a gated early-stop loop (below) starts iteration i >= 1 from the TENTATIVE state the launch before it left in
            // es_xte -- x_t holds the committed post-iteration state, which only a stopped loop reads again
            load_f32<VEC>((ES && es_gated && (ph & LP_PH_POST_STEADY)) ? d.es_xte : d.x_t, i, xt);
        }
        if ((ph & LP_PH_POST_STEADY) || ((ph & LP_PH_PRE_HALF) && !post)) load_f32<VEC>(d.C, i, cv);
        if constexpr (UNI) {
            const uint32_t nib = (m_raw.w[0] >> (static_cast<uint32_t>(i) & 31u)) & 0xFu;       // this lane's 4 mask bits
            const bool any_known = __ballot(nib != 0u) != 0ull, any_inpaint = __ballot(nib != 0xFu) != 0ull;
            uni = !any_known ? 0 : (!any_inpaint ? 1 : -1);
        }
        if constexpr (RA) {
            if (!given && d.x0_big != d.x0 && !(fl & (LP_FL_CFG_FUSED | LP_FL_NO_REGION_SKIP))) {   // (HARD: no corr_el)
                const uint32_t nib = (m_raw.w[0] >> (static_cast<uint32_t>(i) & 31u)) & 0xFu;   // this lane's 4 mask bits
                need_known = uni != 0;
                need_x0 = uni != 1;
                lane_known = nib != 0u;
                lane_x0 = nib != 0xFu;
                if constexpr (PAIR) {              // the pair's eight mask bits sit in the same word (its first element is 8 k)
                    const uint32_t nib8 = (m_raw.w[0] >> (static_cast<uint32_t>(i) & 24u)) & 0xFFu;
                    pair_known = nib8 != 0u;
                    pair_x0 = nib8 != 0xFFu;
                }
            }
        }
        const int64_t i_x0 = i, i_kn = i;
        if (post) {
            if constexpr (PAIR) {
                if (need_x0) load_raw_pair(d.x0, i, pair_x0, x0_raw);
                if (!(d.x0_big == d.x0 || given) && need_known) load_raw_pair(d.x0_big, i, pair_known, x0b_raw);
            } else {
                if (need_x0 && lane_x0) load_raw_w<VEC, X0W>(d.x0, x0dt, i_x0, x0_raw);
                if (!(d.x0_big == d.x0 || given) && need_known && lane_known) load_raw_w<VEC, X0W>(d.x0_big, x0dt, i_kn, x0b_raw);
            }
        }
        // ---- from here on the descriptor proper is needed (the first wait for the argument segment) ----
        if constexpr (SMALL) load_mask_raw<VEC>(d.mask, mfl, i, m_raw);
        const bool has_corr = d.corr_el != nullptr && !given;
        const bool host_post = HOT ? false : d.xi_post != nullptr, host_pre = HOT ? false : d.xi_pre != nullptr;
        const bool need_rng = (post && !host_post) || ((ph & LP_PH_PRE_HALF) && !host_pre);
        if (post) {
            if (!given && need_known && lane_known) load_f32<VEC>(d.y, i_kn, yv);
            if (host_post) load_f32<VEC>(d.xi_post, i, xi_a);
            if (has_corr) load_f32<VEC>(d.corr_el, i, corr);
        }
        if ((ph & LP_PH_PRE_HALF) && host_pre) load_f32<VEC>(d.xi_pre, i, xi_b);
        // early stop: the previous x0s, the drift anchor and the ring weight the metric compares against
        float x0p[VEC], anc[VEC], rg[VEC], es_b0[VEC], es_b1[VEC], es_b2[VEC];
        // The mask-edge ring weight (earlystop.py:32-49) of a HARD mask is 0 or 1 -- ring pixels are inpaint pixels, whose
        // weight 1 - m is 1 -- so it travels as bits like the mask (LP_FL_ES_RING_BITS: 0.125 B instead of 4 B per element and
        // one register instead of four; the phase-specialised hard-mask kernels take no other form, lp_step routes an fp32
        // ring to the run-time kernels)
        constexpr bool RING_BITS_ONLY = ES != 0 && HARD && PH != 0;
        const bool ring_bits = RING_BITS_ONLY || (fl & LP_FL_ES_RING_BITS) != 0;
        uint32_t rg_raw = 0u;
        auto load_ring = [&]() __attribute__((always_inline)) {
            if (!d.es_ring) return;
            if (ring_bits) {
                if constexpr (!ST) rg_raw = reinterpret_cast<const uint32_t*>(d.es_ring)[i >> 5];
            } else {
                if constexpr (!RING_BITS_ONLY) load_f32<VEC>(d.es_ring, i, rg);
            }
        };
        auto ring_weight = [&](int k) __attribute__((always_inline)) -> float {
            if (!d.es_ring) return 0.0f;
            if (ring_bits) {
                if constexpr (!ST) return static_cast<float>((rg_raw >> ((static_cast<uint32_t>(i) & 31u) + k)) & 1u);
                else return 0.0f;
            }
            if constexpr (!RING_BITS_ONLY) return rg[k];
            else return 0.0f;
        };
        if constexpr (ES) {
            if (post) {
                if constexpr (es_fold && VEC == 4) {
                    // (streaming sizes: see below -- the verdict is formed right here, behind the operand loads, and the history
                    // loads follow it)
                } else if constexpr (es_fold) {      // latency-bound sizes: which two of the three buffers the metric compares
                                                     // against is part of the pending verdict -- read all three, select later
                    load_f32<VEC>(d.es_x0s[0], i, es_b0);
                    load_f32<VEC>(d.es_x0s[1], i, es_b1);
                    load_f32<VEC>(d.es_x0s[2], i, es_b2);
                } else {
                    if (es_prev >= 0) load_f32<VEC>(es_prev == 0 ? d.es_x0s[0] : es_prev == 1 ? d.es_x0s[1] : d.es_x0s[2], i, x0p);
                    if (es_anchor >= 0) load_f32<VEC>(es_anchor == 0 ? d.es_x0s[0] : es_anchor == 1 ? d.es_x0s[1] : d.es_x0s[2], i, anc);
                }
                if constexpr (!(es_fold && VEC == 4)) {
                    load_ring();
                }
            }
        }

        // Streaming kernels: the device-side generator state of a replayed graph, read HERE -- behind the operand loads, ahead of
        // the first store of the kernel.  Ahead of every store the compiler can prove the words unclobbered and reads them with a
        // SCALAR load (its own counter); round 4 read them inside the noise block, behind the I/O-table store below: a VECTOR
        // load, whose `s_waitcnt vmcnt(0)` made the Philox rounds wait for every operand load issued before it -- in a replayed
        // graph (the engine's launches; the micro-benchmarks pass no state pointer) the noise was generated AFTER the memory
        // latency instead of under it (rocprofv3, C5 steady: 9.05 us in the bench's replays against 8.57 in the micro-benchmark).
        if constexpr (!SMALL) {
            if (need_rng && d.rng_offset_ptr) {
                rng_w0 = d.rng_offset_ptr[0];
                rng_w1 = d.rng_offset_ptr[1];
                rng_have = true;
            }
        }
        // per-call I/O pointers for the launches of this sigma call that live in a captured graph (lp_finalize)
        if (d.io_table_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            d.io_table_out[0] = d.io_table_val[0];
            d.io_table_out[1] = d.io_table_val[1];
            if (d.io_valid) d.io_table_out[2] = 1ull;      // (not on a speculated call: there the sigma rule owns the word)
        }
        if constexpr (PH == 0 || (PH & LP_PH_REPLACE) != 0) {
            if (d.es_reset && d.es && blockIdx.x == 0 && blockIdx.y == 0) {
                if (d.es_partials) {                      // every accumulator set starts the call at zero
                    for (int k = threadIdx.x; k < kEsSets * kEsSlots * 8; k += kBlock) d.es_partials[k] = 0.0;
                }
                if (threadIdx.x == 0) es_reset_state(d, fold_coeffs);
            }
        }

        LP_CLK(1)
        // ---- Philox + Box-Muller while the loads are in flight ----------------------------
        if (need_rng) {
            const bool torch_kind = RNG == 2 ? (d.rng_kind == LP_RNG_TORCH) : (RNG == 1);
            uint64_t seq = d.rng_offset, seed = d.rng_seed;
            if (rng_have) {
                seq += rng_w0;
                if (torch_kind) seed = rng_w1;
            }
            if (torch_kind) {
                // torch.randn_like(x_t) twice, in the reference's order: the POST draw, then the PRE draw
                const bool draw_post = post && !host_post, draw_pre = (ph & LP_PH_PRE_HALF) && !host_pre;
                const uint64_t off_pre = seq + (draw_post ? d.rng_inc : 0u);
                if constexpr (ST) {
                    static_assert(!ST || VEC == 4, "the ATen layout is four elements per lane");
                    // ATen thread j = 256 b + threadIdx.x evaluates ONE Philox4x32 block per draw (round q of its grid-stride loop:
                    // counter offset / 4 + q) and gets the values of its four elements -- which lie bg apart.  Round 4 gave the LANE
                    // those four elements (four 4-byte streams per tensor: 55 memory instructions per lane, 585 VALU instructions per
                    // wave with their per-slot predicates and addresses).  Now the values are TRANSPOSED through LDS -- written
                    // [slot][thread], read back by wave `slot` as 16 bytes per lane -- so that every tensor is streamed 16 bytes per
                    // lane in memory order like the Philox2x32 kernels' (8 KB of LDS per block, one barrier): the generator keeps
                    // its order, memory keeps its own, LDS is where they meet.
                    __shared__ __attribute__((aligned(16))) float st_xi[2][4][kBlock];
                    const uint32_t aten_thread = static_cast<uint32_t>(g_raw);
                    if (draw_post) {
                        float z[4];
                        torch_normal4(aten_thread, seed, seq + 4ull * st_round, z);
#pragma unroll
                        for (int k = 0; k < 4; ++k) st_xi[0][k][threadIdx.x] = z[k];
                    }
                    if (draw_pre) {
                        float z[4];
                        torch_normal4(aten_thread, seed, off_pre + 4ull * st_round, z);
#pragma unroll
                        for (int k = 0; k < 4; ++k) st_xi[1][k][threadIdx.x] = z[k];
                    }
                    __syncthreads();
                    const int w = threadIdx.x >> 6, l4 = (threadIdx.x & 63) * 4;
                    if (draw_post) {
                        const float4 q = *reinterpret_cast<const float4*>(&st_xi[0][w][l4]);
                        xi_a[0] = q.x; xi_a[1] = q.y; xi_a[2] = q.z; xi_a[3] = q.w;
                    }
                    if (draw_pre) {
                        const float4 q = *reinterpret_cast<const float4*>(&st_xi[1][w][l4]);
                        xi_b[0] = q.x; xi_b[1] = q.y; xi_b[2] = q.z; xi_b[3] = q.w;
                    }
                } else {
                    const bool small = d.n_el <= static_cast<int64_t>(d.rng_bg);     // one ATen thread per element
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const uint64_t li = static_cast<uint64_t>(elem_index(i, k));
                        // (phase-specialised kernels: one v_mad_u64_u32 per product of a Philox4x32 round instead of a mul_hi / mul_lo
                        // pair -- 40 quarter-rate multiplies per draw instead of 80 in a launch that has one wave per SIMD at image sizes)
                        if (draw_post) xi_a[k] = torch_normal<PH != 0>(li, seed, seq, d.rng_bg, small);
                        if (draw_pre) xi_b[k] = torch_normal<PH != 0>(li, seed, off_pre, d.rng_bg, small);
                    }
                }
            } else {
                // every element index below 2^32 (any real latent): the Philox key does not depend on the element and its ten
                // round keys are scalar -- one v_add less per round and element (lp_common.h, normal_pair_key)
                // (streaming kernels only: at one element per lane the launch is latency-bound and the per-element key of round 3
                // is the shorter chain -- same-box A/B against the round-3 library, scripts/r04_ab_c2.sh)
                const bool key_uniform = !SMALL && d.n_el <= 0xffffffffll;
                const uint32_t key0 = philox_key(seed, seq, 0);
                if constexpr (VEC == 4 && HOT) {
                    if (key_uniform) {
                        // the lane's four Philox blocks, then Box-Muller on 2-vectors (packed fp32, lp_common.h::normal_pairs4_key)
                        uint32_t e4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) e4[k] = static_cast<uint32_t>(elem_index(i, k));
                        normal_pairs4_key<true>(e4, static_cast<uint32_t>(seq), key0, xi_a, xi_b);
                    } else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) normal_pair(static_cast<uint64_t>(elem_index(i, k)), seq, seed, xi_a[k], xi_b[k]);
                    }
                } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float za, zb;
                    const uint64_t e = static_cast<uint64_t>(elem_index(i, k));
                    // (one v_mad_u64_u32 per Philox round in the phase-specialised kernels; the run-time-phase ones sit at the SGPR
                    // limit and the instruction's carry pair tips them into a private segment)
                    if (key_uniform) normal_pair_key<PH != 0>(static_cast<uint32_t>(e), static_cast<uint32_t>(seq), key0, za, zb);
                    else normal_pair(e, seq, seed, za, zb);
                    if (!host_post) xi_a[k] = za;
                    if (!host_pre) xi_b[k] = zb;
                }
                }
            }
        }

        if constexpr (ES == 2 && VEC == 4) {
            // exactly the buffers the metric compares against -- the previous x0s, the drift anchor while one is held, the ring
            // bits of a 4-D latent -- now that the verdict (formed behind the operand loads, above) says which they are
            if (post) {
                if (es_prev >= 0) load_f32<VEC>(es_prev == 0 ? d.es_x0s[0] : es_prev == 1 ? d.es_x0s[1] : d.es_x0s[2], i, x0p);
                load_ring();
            }
        }
        LP_CLK(2)
        LP_CLK_LOADS(3)
        // ---- folded early stop: the verdict of the previous iteration, now that its inputs have had time to arrive --
        if constexpr (ES) {
            if constexpr (es_fold) {
                if constexpr (VEC != 4) form_verdict();
                if constexpr (VEC != 4) {
                    if (post) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            x0p[k] = es_prev == 0 ? es_b0[k] : es_prev == 1 ? es_b1[k] : es_b2[k];
                            anc[k] = es_anchor == 0 ? es_b0[k] : es_anchor == 1 ? es_b1[k] : es_b2[k];
                        }
                    }
                }
            }
        }
        LP_CLK(4)
        const bool live = active && !es_idle;         // lanes that commit results (a stopped folded loop only emits)

        // ---- decode what was loaded in a storage format --------------------------------------------
        if (UNI && uni >= 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) m[k] = static_cast<float>(uni);
        } else {
            cvt_mask<VEC>(mfl, i, m_raw, m);
        }
        if (post) {
            if constexpr (PAIR) {
                cvt_raw_pair(x0dt, x0_raw, x0);
                if (!(d.x0_big == d.x0 || given)) cvt_raw_pair(x0dt, x0b_raw, x0b);
            } else {
                cvt_raw<VEC>(x0dt, x0_raw, x0, i);
                if (!(d.x0_big == d.x0 || given)) cvt_raw<VEC>(x0dt, x0b_raw, x0b, i);
            }
        }

        // ---- REPLACE: x = x(1-m) + known*m ; x_t = VP(x) -----------------------------------
        if (ph & LP_PH_REPLACE) {
            if (d.replace_kind != LP_REPLACE_KNOWN) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float r = PER_EL ? rs[k] : row_rsigma_of(k);
                    kn[k] = (d.replace_kind == LP_REPLACE_VE) ? fmaf(nv[k], r, yv[k])
                                                              : (r * (d.noise_scale * nv[k]) + (1.0f - r) * yv[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float xr = fmaf(kn[k], m[k], xv[k] * (1.0f - m[k]));
                float sc;
                if constexpr (PER_EL) {
                    sc = flow ? (sqrtf(abt_e[k]) + sqrtf(1.0f - abt_e[k])) : sqrtf(1.0f + ve_e[k] * ve_e[k]);
                } else {
                    sc = row_scale_of(k);
                }
                if constexpr (VEC == 4 && !PER_EL && PH != 0) {
                    // streaming replace launch of a VE model: the row's scale divides all four elements of the lane -- the reciprocal
                    // formed once per lane above (rc.rscale) and one residual correction per element give the IEEE quotient
                    // (div_shared: 11 + 4 x 3 instructions instead of 4 x 11; round 5, the last of round 4's listed candidates)
                    xt[k] = flow ? xr * sc : div_shared(xr, sc, rc.rscale);
                } else {
                    xt[k] = flow ? xr * sc : xr / sc;
                }
            }
        }

        // OU(x, dt/2, C) of lanpaint.py:280 for element k (table path or the reference's own formulas)
        // (table form with the region's coefficients given: `q` is an SGPR set in a mask-uniform wave)
        auto half_valid = [&](float x, float c, float xi, const RegionCoef& q) __attribute__((always_inline)) -> float {
            return fmaf(q.e_half, x, fmaf(q.k_half, c, q.std_half * xi));
        };
        auto half_table = [&](float x, float c, float xi, const RegionCoef& q) __attribute__((always_inline)) -> float {
            return rc.valid != 0.0f ? half_valid(x, c, xi, q) : x;
        };
        auto half_step = [&](float x, float c, float xi, int k) __attribute__((always_inline)) -> float {
            const float mk = m[k];
            const bool table = HARD || (!PER_EL && !av_mixed && ((mk == 0.0f) || (mk == 1.0f)));
            if (table) {
                // streaming kernels: a branch, not a select between the two coefficient sets -- each side then takes its set
                // straight from SGPRs (a v_cndmask needs one of the two in VGPRs, ten registers pinned for the whole kernel).
                // One-element-per-lane kernels hold the row in VGPRs anyway (load_row_early) and are latency-bound: there the
                // select is the shorter dependent chain (same-box A/B at C2: 2.09 us with selects, 2.18 with branches)
                if constexpr (SMALL) return half_table(x, c, xi, rc.reg[mk == 1.0f ? 1 : 0]);
                if (mk == 1.0f) return half_table(x, c, xi, rc.reg[1]);
                return half_table(x, c, xi, rc.reg[0]);
            }
            if constexpr (!HARD) {
                ElemCoef e;
                if constexpr (PER_EL) {
                    e = elem_from_times(abt_e[k], flow ? 0.0f : ve_e[k], mk, flow, opl, d.beta, d.step_size, d.min_step_frac);
                } else {
                    e = elem_of(k, mk);
                }
                if (e.valid) return ou_general(x, e.dt / 2.0f, e.a, c, e.d, xi);
            }
            return x;
        };

        float x0s[VEC], xb[VEC];
        if constexpr (ES) {
            if (ph & LP_PH_POST_FIRST) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xb[k] = xt[k];     // x_t_before of iteration 0 (earlystop.py:288)
            }
        }

        // ---- POST: score split -> x0s, C' ; drift correction ; OU ----------------------------
        if (post) {
            if (d.x0_big == d.x0 || given) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) x0b[k] = x0[k];
            } else if (fl & LP_FL_CFG_FUSED) {      // x0 = cond, x0b = uncond -> both CFG heads
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float c = x0[k], u = x0b[k], diff = c - u;
                    x0[k] = fmaf(diff, d.cfg_scale, u);
                    x0b[k] = fmaf(diff, d.cfg_scale_big, u);
                }
            }
            if (given) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) yv[k] = 0.0f;
            }
            // (iteration 0 has no previous C: a row whose step is not positive stores C = 0, below -- assigned where the element is
            // handled, not pre-zeroed here: an array defined on two paths and merged kept the compiler from promoting it to
            // registers in the first-iteration kernels, which then went through scratch memory)
            if constexpr (!HARD) {
                if (has_corr) {       // lanpaint.py:173-180: both heads pulled towards the model-space input, before the split
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float sc;
                        if constexpr (PER_EL) sc = flow ? (sqrtf(abt_e[k]) + sqrtf(1.0f - abt_e[k])) : sqrtf(1.0f + ve_e[k] * ve_e[k]);
                        else sc = row_scale_of(k);
                        const float xm = flow ? xt[k] / sc : xt[k] * sc;
                        x0[k] = xm + corr[k] * (x0[k] - xm);
                        x0b[k] = xm + corr[k] * (x0b[k] - xm);
                    }
                }
            }
            // table path of element k: two regions per row, no transcendental per element (`q`: the region's coefficients)
            // (post_vals: the three results of element k in region `q` -- x0s, the new C, the new x_t -- as VALUES, so that a mixed wave
            // can form both regions' and select; post_valid: the same stored)
            auto post_vals = [&](int k, const RegionCoef& q, bool known_el, float& s0, float& cn, float& xn) __attribute__((always_inline)) {
                s0 = (given || !known_el) ? x0[k] : fmaf(-lam, x0b[k], opl * yv[k]);
                cn = fmaf(q.cx0, s0, q.cxt * xt[k]);
                if (ph & LP_PH_POST_FIRST) {
                    xn = fmaf(q.e_full, xt[k], fmaf(q.k_full, cn, q.std_full * xi_a[k]));
                } else {
                    const float xd = fmaf(cn - cv[k], q.dt, xt[k]);
                    xn = fmaf(q.e_half, xd, fmaf(q.k_half, cv[k], q.std_half * xi_a[k]));
                }
            };
            auto post_valid = [&](int k, const RegionCoef& q, bool known_el) __attribute__((always_inline)) {
                float s0, cn, xn;
                post_vals(k, q, known_el, s0, cn, xn);
                x0s[k] = s0;
                xt[k] = xn;
                cv[k] = cn;
            };
            auto post_skipped = [&](int k) __attribute__((always_inline)) {      // a row whose step is not positive (lanpaint.py:205)
                x0s[k] = x0[k];
                if (ph & LP_PH_POST_FIRST) cv[k] = 0.0f;
            };
            auto post_table = [&](int k, const RegionCoef& q, bool known_el) __attribute__((always_inline)) {
                if (rc.valid != 0.0f) post_valid(k, q, known_el);
                else post_skipped(k);
            };
            if (UNI && uni >= 0) {               // mask-uniform wave: the region's coefficient set stays in SGPRs
                // (the row's `valid` test OUTSIDE the element loop: four elements of straight-line arithmetic on scalar coefficients,
                // which the compiler pairs into packed fp32 instructions -- v_pk_fma_f32, two IEEE FMAs per instruction; with the
                // test inside, every element sat behind its own scalar branch)
                if (rc.valid == 0.0f) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) post_skipped(k);
                } else if (uni == 0) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) post_valid(k, rc.reg[0], false);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) post_valid(k, rc.reg[1], true);
                }
            } else if (MIXSEL && rc.valid != 0.0f) {
                // A MIXED wave of a hard mask at 16 bytes per lane (mask edges; a spatial mask on a video latent is mixed waves
                // only): both regions' arithmetic for every element, on the two scalar coefficient sets -- straight-line code the
                // compiler pairs into packed fp32 like the uniform waves' -- and one select per result, instead of a branch per
                // element whose two sides a mixed wave walks one after the other anyway (round 5).  Same expressions on the same
                // values for the side an element takes: bit-identical; what the other side computes -- possibly from a stream the
                // lane did not load -- is discarded.
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const bool kn = m[k] == 1.0f;
                    float s0a, cna, xna, s0b, cnb, xnb;
                    post_vals(k, rc.reg[0], false, s0a, cna, xna);
                    post_vals(k, rc.reg[1], true, s0b, cnb, xnb);
                    x0s[k] = kn ? s0b : s0a;
                    xt[k] = kn ? xnb : xna;
                    cv[k] = kn ? cnb : cna;
                }
            } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float mk = m[k];
                const bool table = HARD || (!PER_EL && !av_mixed && ((mk == 0.0f) || (mk == 1.0f)));
                if (table) {
                    if constexpr (SMALL) {
                        post_table(k, rc.reg[mk == 1.0f ? 1 : 0], mk == 1.0f);
                    } else {
                        if (mk == 1.0f) post_table(k, rc.reg[1], true);
                        else post_table(k, rc.reg[0], false);
                    }
                } else if constexpr (!HARD) {
                    ElemCoef e;
                    if constexpr (PER_EL) {
                        e = elem_from_times(abt_e[k], flow ? 0.0f : ve_e[k], mk, flow, opl, d.beta, d.step_size,
                                            d.min_step_frac);
                    } else {
                        e = elem_of(k, mk);
                    }
                    const float h0 = x0[k], h1 = x0b[k];          // (the audio correction went into the heads above)
                    if (e.valid) {
                        float s0 = h0;
                        if (!given) {
                            const float score_x = -(xt[k] - h0);
                            const float score_y = -opl * (xt[k] - yv[k]) + lam * (xt[k] - h1);
                            s0 = xt[k] + (score_x * (1.0f - mk) + score_y * mk);
                        }
                        const float cn = (e.sqrt_abt * s0 - xt[k]) / e.oma + e.a * xt[k];
                        x0s[k] = s0;
                        if (ph & LP_PH_POST_FIRST) {
                            xt[k] = ou_general(xt[k], e.dt, e.a, cn, e.d, xi_a[k]);
                        } else {
                            const float xd = xt[k] + (cn - cv[k]) * e.dt;
                            xt[k] = ou_general(xd, e.dt / 2.0f, e.a, cv[k], e.d, xi_a[k]);
                        }
                        cv[k] = cn;
                    } else {
                        x0s[k] = h0;
                        if (ph & LP_PH_POST_FIRST) cv[k] = 0.0f;
                    }
                }
            }
            }
            if ((fl & LP_FL_WRITE_X0S) && live) store_f32<VEC>(d.x0s, i, x0s);
            if constexpr (ES) {
                if (live) {
                    store_f32<VEC>(es_write == 0 ? d.es_x0s[0] : es_write == 1 ? d.es_x0s[1] : d.es_x0s[2], i, x0s);
                    // weighted squared differences (earlystop.py:52-55): w1 = 1 - mask, w2 = ring
                    if constexpr (ES == 2 && VEC == 4) {
                        // streaming sizes: the drift anchor (held only while the distance sits under the threshold) is fetched
                        // HERE, after the arithmetic, not with the operands: four registers less at the kernel's peak -- the ones
                        // that decide between 7 and 8 waves per SIMD; its latency is exposed only in iterations that hold an anchor
                        if (es_anchor >= 0 && !(UNI && uni == 1))
                            load_f32<VEC>(es_anchor == 0 ? d.es_x0s[0] : es_anchor == 1 ? d.es_x0s[1] : d.es_x0s[2], i, anc);
                    }
                    if (UNI && uni == 1) {
                        // every element of the wave is known: w1 = 1 - m = 0 and the ring (inpaint pixels next to known ones)
                        // has none of them -- the wave adds exact zeros to all six sums, i.e. nothing
                        es_zero_wave = true;
                    } else if (UNI && uni == 0 && !d.es_ring) {
                        // every element inpaint, no ring (5-D latents): w1 = 1 -- the products by 1.0f and the sums of the
                        // weights are the same numbers without the multiplies
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            const float da = es_prev >= 0 ? x0s[k] - x0p[k] : xt[k] - xb[k];
                            es_p[0] += da * da;
                            es_p[1] += 1.0f;
                            if (es_anchor >= 0) {
                                const float db = x0s[k] - anc[k];
                                es_p[4] += db * db;
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            const float w1 = 1.0f - m[k];
                            const float w2 = ring_weight(k);
                            const float da = es_prev >= 0 ? x0s[k] - x0p[k] : xt[k] - xb[k];
                            const float da2 = da * da;
                            es_p[0] += da2 * w1;
                            es_p[1] += w1;
                            es_p[2] += da2 * w2;
                            es_p[3] += w2;
                            if (es_anchor >= 0) {
                                const float db = x0s[k] - anc[k];
                                const float db2 = db * db;
                                es_p[4] += db2 * w1;
                                es_p[5] += db2 * w2;
                            }
                        }
                    }
                }
            }
        }

        // ---- PRE_HALF: first half-step of the next iteration (uses the new C) ------------------
        // (gated early-stop loop: tentative -- it only feeds the emit, x_t keeps the post-iteration state)
        float xe[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) xe[k] = xt[k];
        if (ph & LP_PH_PRE_HALF) {
            if (UNI && uni >= 0 && rc.valid == 0.0f) {
                // (a skipped row keeps x: xe = xt already)
            } else if (UNI && uni == 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xe[k] = half_valid(xt[k], cv[k], xi_b[k], rc.reg[0]);
            } else if (UNI && uni == 1) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xe[k] = half_valid(xt[k], cv[k], xi_b[k], rc.reg[1]);
            } else if (MIXSEL && rc.valid != 0.0f) {       // mixed wave: both regions' half-step, selected (see POST above)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float ha = half_valid(xt[k], cv[k], xi_b[k], rc.reg[0]), hb = half_valid(xt[k], cv[k], xi_b[k], rc.reg[1]);
                    xe[k] = m[k] == 1.0f ? hb : ha;
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xe[k] = half_step(xt[k], cv[k], xi_b[k], k);
            }
            if (!(ES && es_gated)) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xt[k] = xe[k];
            }
        }

        LP_CLK(5)
        if (post && live) store_f32<VEC>(d.C, i, cv);
        if ((ph & kTouchXt) && live) store_f32<VEC>(d.x_t, i, xt);
        if constexpr (ES) {
            // gated loop: the state after the (tentative) half-step goes to es_xte, where the next launch picks it up; x_t
            // keeps the post-iteration state.  (Round 3 had the next launch REDO the half-step from the same noise instead:
            // a second Philox block + Box-Muller per element, ~200 VALU instructions per wave of a VALU-bound launch, to
            // save these 4 bytes per element.)
            if (es_gated && (ph & LP_PH_PRE_HALF) && live) store_f32<VEC>(d.es_xte, i, xe);
            if (es_fold && es_idle) load_f32<VEC>(d.x_t, i, xe);       // stopped: re-emit x_in from the COMMITTED state
        }

        // ---- EMIT: model-space latent for the next backbone call --------------------------------
        if (ph & LP_PH_EMIT) {
            float xo[VEC];
            if constexpr (VEC == 4 && !PER_EL && PH != 0) {
                // streaming row-table kernels: the row's scale divides all four elements of the lane -- one reciprocal + three
                // instructions per quotient instead of four IEEE division sequences, same values (div_shared, lp_common.h).
                const float sc = rc.scale;
                if (flow) {
                    // (round 5: the correctly rounded reciprocal of the row's scale comes from the coefficient table, LP_C_RSCALE --
                    // the IEEE sequence for 1.0f / sc was 11 instructions per lane)
                    // (the early-stop launches sit at their SGPR limit -- one more live scalar sent the torch-stream one to scratch
                    // memory -- and keep forming the reciprocal themselves; the empty asm keeps it behind the branch)
                    float y;
                    if constexpr (ES != 0 || (RNG == 1 && !ST)) {     // (likewise the non-strided torch-stream kernels: one of them at 101 SGPRs)
                        y = 1.0f / sc;
                        asm volatile("" : "+v"(y));
                    } else {
                        y = rc.rscale;
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) xo[k] = div_shared(xe[k], sc, y);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) xo[k] = xe[k] * sc;
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float sc;
                    if constexpr (PER_EL) {
                        sc = flow ? (sqrtf(abt_e[k]) + sqrtf(1.0f - abt_e[k])) : sqrtf(1.0f + ve_e[k] * ve_e[k]);
                    } else {
                        sc = row_scale_of(k);
                    }
                    xo[k] = flow ? xe[k] / sc : xe[k] * sc;
                }
            }
            if constexpr (PAIR) {
                if (xindt != DT_F32) store_half_pair(d.x_in, xindt, i, xo);
                else store_any<VEC>(d.x_in, xindt, i, xo);
            } else {
                if (active) store_any<VEC>(d.x_in, xindt, i, xo);
            }
        }
    }

    LP_CLK(6)
    if constexpr (!ES) {
        LP_CLK_FLUSH
    }
    // ---- early stop: this block's sums go into the iteration's accumulator set (device-scope atomics, no fence); the
    // next launch of a gated loop / lp_es_decide_kernel of a watched one totals the 64 slots and applies the rule.
    // (One kernel with a "last block done" ticket needs a device-scope fence per block, i.e. an L2 write-back on
    // every XCD: measured 30 us per launch at 65 536 elements.  A kernel boundary gives the same visibility.)
    if constexpr (ES) {
        if (es_idle || !post) return;
        // fp32 throughout, like the reference's own sums (earlystop.py:52-55), in a fixed order
        __shared__ float es_part[4][kEsSums];
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        if (!es_zero_wave) wave_sum_dpp(es_p);          // (a wave of known elements holds six exact zeros in every lane)
        if (lane == kWave - 1) {
#pragma unroll
            for (int k = 0; k < kEsSums; ++k) es_part[wave][k] = es_p[k];
        }
        __syncthreads();
        const unsigned blk = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x < kEsSums && !(d.tune & LP_TUNE_ES_NO_ATOMICS)) {
            const float p0 = es_part[0][threadIdx.x], p1 = es_part[1][threadIdx.x], p2 = es_part[2][threadIdx.x],
                        p3 = es_part[3][threadIdx.x];
            const float v = ((p0 + p1) + p2) + p3;                  // the block's sum, fp32, fixed order
            double* acc = es_acc_set(d, d.es_index) + static_cast<size_t>(blk % kEsSlots) * 8 + threadIdx.x;
            // unsafeAtomicAdd: the hardware global_atomic_add_f64 for THIS add only (no compare-and-swap loop), instead of
            // building the whole library with -munsafe-fp-atomics; well defined on torch's coarse-grained device allocations
            (void)unsafeAtomicAdd(acc, static_cast<double>(v));
        }
        // the set of the iteration after this one starts from zero (nobody reads or adds to it during this launch)
        if (blk == 0 && threadIdx.x >= kWave) {
            double* nxt = es_acc_set(d, d.es_index + 1);
            for (int k = threadIdx.x - kWave; k < kEsSlots * 8; k += kBlock - kWave) nxt[k] = 0.0;
        }
        LP_CLK(7)
        LP_CLK_FLUSH
    }
}

// One wave: fixed-order total of the accumulator slots of iteration d.es_index, then the stop rule.
// `slot`: which of the two state slots (0 unless it closes a folded loop).
__global__ __launch_bounds__(64) void lp_es_decide_kernel(const lp_step_desc d, int slot) {
    // Both loads of this kernel are issued up front -- the lane's accumulator slot, and (lane 0) the whole state as one
    // block load -- so the launch pays ONE memory round trip.  A stopped loop simply discards what it loaded.
    const double* src = es_acc_set(d, d.es_index) + static_cast<size_t>(threadIdx.x) * 8;
    double v[kEsSums];
#pragma unroll
    for (int k = 0; k < kEsSums; ++k) v[k] = __builtin_nontemporal_load(src + k);
    lp_es_state st;
    if (threadIdx.x == 0) st = d.es[slot];
    const bool gated = (d.flags & LP_FL_ES_GATED) != 0;
    float tot[kEsSums];
    es_slot_total(v, tot);                                                  // fixed order: the DPP tree over the 64 slots
    if (threadIdx.x != 0) return;
    if (gated && st.stopped != 0) {      // stopped loop: its last launch tells the host the call is done
        if (d.es_index + 1 == d.es_n_steps) {
            d.es[0] = st;
            d.es[1] = st;
            if (d.es_host) es_post_seq(d.es_host, st.seq_base + LP_ES_SEQ_DONE);
        }
        return;
    }
    es_decide(d, st, tot, st.cur_slot >= 0, st.anchor_slot >= 0, d.es_index, true);
    d.es[0] = st;                        // both slots: the next reset / the next watched launch start from slot 0
    d.es[1] = st;
}

// ---- launch geometry ------------------------------------------------------------------------
struct Timer {
    hipEvent_t start, stop;
};

// Developer switches of a launch (lp_step_desc.tune, LP_TUNE_*): the micro-benchmarks A/B launch geometries and the two
// early-stop layouts through the descriptor.  The library reads nothing from the environment and keeps no process-wide state.
// ST launches: blockIdx.y = round * per_round + j, j-th batch row crossing that round of 4 bg elements (at most
// ceil(4 bg / el_per_row) + 1 of them, never more than there are rows); 0 = not representable
static unsigned st_segments(const lp_step_desc& d) {
    const int64_t span = 4 * static_cast<int64_t>(d.rng_bg);
    const int64_t rounds = (d.n_el + span - 1) / span;
    int64_t per_round = (span + d.el_per_row - 1) / d.el_per_row + 1;
    if (per_round > d.rows) per_round = d.rows;
    const int64_t gy = rounds * per_round;
    return (gy > 0 && gy <= 65535 && static_cast<int64_t>(d.rng_inc) == 4 * rounds) ? static_cast<unsigned>(gy) : 0u;
}

// ---- coverage build (-DLP_TRACE_INSTANTIATIONS; build/liblanpaint_hip_trace.so, never the product library) ------------
// Which lp_step_kernel<...> instantiations does a process really launch?  Every launch notes its template arguments; the
// set is appended to the file LANPAINT_AMD_TRACE_FILE names when the process exits.  scripts/instantiation_coverage.py
// compares it with the instantiations the product library contains (tests/test_cabi_exports.py keeps the two in step).
#ifdef LP_TRACE_INSTANTIATIONS
}  // namespace lp
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
namespace lp {
struct TraceSet {
    std::mutex mu;
    std::set<std::string> seen;
    ~TraceSet() {
        const char* path = std::getenv("LANPAINT_AMD_TRACE_FILE");
        if (!path || seen.empty()) return;
        if (FILE* f = std::fopen(path, "a")) {
            for (const auto& s : seen) std::fprintf(f, "%s\n", s.c_str());
            std::fclose(f);
        }
    }
};
static TraceSet& trace_set() {
    static TraceSet t;
    return t;
}
static void trace_note(int vec, int mode, unsigned ph, int x0w, int rng, bool st, int es) {
    char buf[96];
    std::snprintf(buf, sizeof buf, "%d, %d, %uu, %d, %d, %s, %d", vec, mode, ph, x0w, rng, st ? "true" : "false", es);
    TraceSet& t = trace_set();
    std::lock_guard<std::mutex> lock(t.mu);
    t.seen.insert(buf);
}
#define LP_TRACE(es) trace_note(VEC, MODE, PH, X0W, RNG, ST, es)
#else
#define LP_TRACE(es)
#endif

// the preloaded leading arguments of lp_step_kernel<VEC, ...> (see its head) followed by the descriptor
#define LP_STEP_ARGS(d)                                                                                          \
    static_cast<void*>((d).x_t), static_cast<void*>((d).C),                                                      \
        ((PH & LP_PH_REPLACE) ? static_cast<const void*>((d).x) : (d).x0),                                        \
        ((PH & LP_PH_REPLACE) ? static_cast<const void*>((d).replace_kind == LP_REPLACE_KNOWN ? (d).known : (d).noise) \
                              : (d).x0_big),                                                                      \
        ((PH & LP_PH_REPLACE) || VEC != 1 ? static_cast<const void*>((d).y) : static_cast<const void*>((d).coef)), \
        ((PH & LP_PH_REPLACE) || VEC != 1 ? (d).mask : static_cast<const void*>((d).rng_offset_ptr)),             \
        static_cast<int32_t>((d).el_per_row), (d).flags, (d)

template <int VEC, int MODE, uint32_t PH, int X0W = 0, int RNG = 2, bool ST = false, int ES = 0>
static hipError_t launch(const lp_step_desc& d, hipStream_t stream, Timer* timer) {
    const int64_t groups = ST ? static_cast<int64_t>(d.rng_bg) : d.el_per_row / VEC;
    constexpr int block = kBlock;
    int64_t bx = (groups + block - 1) / block;
    // One group per lane, no grid-stride loop: capping the grid at 2048 blocks cost 30 % on a 33 M-element
    // batch (220 -> 170 us; profiles/r01_microbench_kernel_variants.log); the BASELINE shapes all fit in
    // <= 2048 blocks anyway.
    if (bx < 1) bx = 1;
    if (bx > 0x7fffffff) return hipErrorInvalidValue;
    unsigned gy = static_cast<unsigned>(d.rows);
    if constexpr (ST) gy = st_segments(d);            // (round, row) pairs, see the kernel head
    const dim3 grid(static_cast<unsigned>(bx), gy);
    if constexpr (ES != 0) {
        // gated (replayed) loop: the stop rule of iteration i - 1 rides in launch i -- every wave totals the 64 accumulator
        // slots itself -- and one closing lp_es_decide_kernel follows the last launch unless the loop closes itself
        // (LP_FL_ES_CLOSE).  Watched (eager) loop: the host waits for every verdict, so a one-wave kernel forms it right
        // after the launch.  (The folded kernel is its own instantiation: its extra live state would cost the plain
        // early-stop launch registers it does not need.)
        const bool fold = (d.flags & LP_FL_ES_GATED) && !(d.tune & LP_TUNE_ES_NO_FOLD);
        if constexpr (PH != 0) {
            // the phase-specialised early-stop kernels exist in their folded form only (the launches a replayed loop
            // repeats); a fused-phase launch that is not folded (a tuning switch) takes the run-time-phase kernel
            if (!fold) return launch<VEC, MODE, 0, 0, 2, false, 1>(d, stream, timer);
            LP_TRACE(2);
            hipLaunchKernelGGL((lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, 2>), grid, dim3(block), 0, stream, LP_STEP_ARGS(d));
        } else {
            LP_TRACE(fold ? 2 : 1);
            if (fold) hipLaunchKernelGGL((lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, 2>), grid, dim3(block), 0, stream, LP_STEP_ARGS(d));
            else hipLaunchKernelGGL((lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, 1>), grid, dim3(block), 0, stream, LP_STEP_ARGS(d));
        }
        const bool close = fold && (d.flags & LP_FL_ES_CLOSE) && d.es_index + 1 == d.es_n_steps;
        if ((d.phases & kPost) && !(d.tune & LP_TUNE_ES_NO_DECIDE) && !close && (!fold || d.es_index + 1 == d.es_n_steps)) {
            if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
            hipLaunchKernelGGL(lp_es_decide_kernel, dim3(1), dim3(kWave), 0, stream, d, fold ? (d.es_index & 1) : 0);
        }
        return hipGetLastError();
    } else {
        // (inside `else`: after an `if constexpr` that returns, the statements below would still be instantiated for the
        // early-stop launches -- eight lp_step_kernel<..., ES = 1> code objects nothing ever launched, round 3)
        LP_TRACE(ES);
        if (timer) {
            hipExtLaunchKernelGGL((lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, ES>), grid, dim3(block), 0, stream, timer->start,
                                  timer->stop, 0, LP_STEP_ARGS(d));
        } else {
            hipLaunchKernelGGL((lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, ES>), grid, dim3(block), 0, stream, LP_STEP_ARGS(d));
        }
        return hipGetLastError();
    }
}

static bool aligned(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <int VEC>
static hipError_t launch_phase(const lp_step_desc& d, hipStream_t stream, Timer* timer) {
    if (d.flags & LP_FL_PER_ELEMENT) return launch<VEC, MODE_PER_EL, 0>(d, stream, timer);
    constexpr uint32_t R = LP_PH_REPLACE, F = LP_PH_POST_FIRST, S = LP_PH_POST_STEADY, P = LP_PH_PRE_HALF,
                       E = LP_PH_EMIT;
    // a bit-packed mask is hard by construction; the audio correction needs the general branch
    const bool hard = (d.flags & LP_FL_MASK_BITS) && d.corr_el == nullptr;
    if (d.flags & LP_FL_AV)             // two time sets per row (AV packs): the run-time row-table kernels know the flag
        return (d.flags & LP_FL_ES) ? launch<VEC, MODE_ROW, 0, 0, 2, false, 1>(d, stream, timer) : launch<VEC, MODE_ROW, 0>(d, stream, timer);
    // x0s handed in (the public langevin_dynamics) and host-supplied noise (recorded streams, explicit torch.randn_like draws)
    // only exist in the run-time-phase kernels (round 5: their selects left the phase-specialised ones)
    const bool rare = (d.flags & LP_FL_X0S_GIVEN) || d.xi_post || d.xi_pre;
    const bool x0_half = x0_dtype(d.flags) != DT_F32;
    // half-width heads of a streaming launch go 16 bytes per lane pair (lp_common.h): 16-byte aligned streams, rows of 8 k
    // elements; anything else takes the run-time kernel with its 8-byte accesses
    const bool pair_ok = VEC != 4 || !x0_half ||
                         (aligned(d.x0, 16) && aligned(d.x0_big, 16) && (xin_dtype(d.flags) == DT_F32 || aligned(d.x_in, 16)) &&
                          d.el_per_row % 8 == 0);
    if (d.flags & LP_FL_ES) {                                // inner early stop evaluated on the device
        // the two launches a loop repeats, specialised like the plain hot kernels (bit-packed mask, fp32 heads): the
        // run-time phase kernel spends 1.0 us of shader clock before its first operand load, these 0.6
        if (hard && !x0_half && !rare && (!d.es_ring || (d.flags & LP_FL_ES_RING_BITS))) {
            const bool rt = d.rng_kind == LP_RNG_TORCH;
            if (d.phases == (S | P | E))
                return rt ? launch<VEC, MODE_HARD, S | P | E, 4, 1, false, 1>(d, stream, timer)
                          : launch<VEC, MODE_HARD, S | P | E, 4, 0, false, 1>(d, stream, timer);
            if (d.phases == (F | P | E))
                return rt ? launch<VEC, MODE_HARD, F | P | E, 4, 1, false, 1>(d, stream, timer)
                          : launch<VEC, MODE_HARD, F | P | E, 4, 0, false, 1>(d, stream, timer);
        }
        return hard ? launch<VEC, MODE_HARD, 0, 0, 2, false, 1>(d, stream, timer)
                    : launch<VEC, MODE_ROW, 0, 0, 2, false, 1>(d, stream, timer);
    }
    if (d.flags & LP_FL_MASK_U8) return launch<VEC, MODE_ROW, 0>(d, stream, timer);   // legacy format: run-time everything
    // (a replace launch WITHOUT the folded table -- a caller that ran lp_coeffs itself -- and a loop of ONE iteration,
    // F | E, run through the run-time-phase kernel: rare launches that do not earn instantiations of their own)
    if (d.phases == (R | E | LP_PH_COEFFS))                                          // replace step + the coefficient table
        return hard ? launch<VEC, MODE_HARD, R | E | LP_PH_COEFFS>(d, stream, timer)
                    : launch<VEC, MODE_ROW, R | E | LP_PH_COEFFS>(d, stream, timer);
    if (d.phases == (R | E | LP_PH_COEFFS | LP_PH_SIGMA))                            // ... + the sigma algebra of the call (bit mask only)
        return launch<VEC, MODE_HARD, R | E | LP_PH_COEFFS | LP_PH_SIGMA>(d, stream, timer);
    const bool rng_torch = d.rng_kind == LP_RNG_TORCH;
    // ATen's element-to-thread layout pays off when a Philox block really serves several elements of this tensor
    // (batch rows: as long as a row covers at least half a round most lanes still use two or more values of their block)
    const bool strided = VEC == 4 && rng_torch && !d.xi_post && !d.xi_pre && d.n_el > static_cast<int64_t>(d.rng_bg) &&
                         (d.rng_bg % kBlock) == 0 &&           /* a block is 256 whole ATen threads */
                         (d.rows == 1 || d.el_per_row >= 2 * static_cast<int64_t>(d.rng_bg)) && st_segments(d) != 0;
#define LP_HOT(MODE_, PH_)                                                                                   \
    (strided ? (x0_half ? launch<4, MODE_, PH_, 2, 1, true>(d, stream, timer) : launch<4, MODE_, PH_, 4, 1, true>(d, stream, timer)) \
     : x0_half ? (rng_torch ? launch<VEC, MODE_, PH_, 2, 1>(d, stream, timer) : launch<VEC, MODE_, PH_, 2, 0>(d, stream, timer)) \
               : (rng_torch ? launch<VEC, MODE_, PH_, 4, 1>(d, stream, timer) : launch<VEC, MODE_, PH_, 4, 0>(d, stream, timer)))
    if (!pair_ok || rare) return hard ? launch<VEC, MODE_HARD, 0>(d, stream, timer) : launch<VEC, MODE_ROW, 0>(d, stream, timer);
    if (hard) {
        switch (d.phases) {
            case S | P | E: return LP_HOT(MODE_HARD, S | P | E);   // steady state
            case F | P | E: return LP_HOT(MODE_HARD, F | P | E);   // iteration 0
            case S | E: return LP_HOT(MODE_HARD, S | E);           // last iteration
            default: return launch<VEC, MODE_HARD, 0>(d, stream, timer);   // unfused (early stop), n_steps == 1, etc.
        }
    }
    switch (d.phases) {
        case S | P | E: return LP_HOT(MODE_ROW, S | P | E);
        case F | P | E: return LP_HOT(MODE_ROW, F | P | E);
        case S | E: return LP_HOT(MODE_ROW, S | E);
        default: return launch<VEC, MODE_ROW, 0>(d, stream, timer);
    }
#undef LP_HOT
}

int step_dispatch(const lp_step_desc* dp, hipStream_t stream, void* timer_handle) {
    Timer* timer = static_cast<Timer*>(timer_handle);
    if (!dp) return LP_E_INVALID;
    const lp_step_desc& d = *dp;
    if (d.n_el <= 0 || d.rows <= 0 || d.el_per_row <= 0 || d.n_el != d.el_per_row * d.rows) return LP_E_INVALID;
    if (d.rows > 65535 || d.el_per_row > 0x7fffffff) return LP_E_UNSUPPORTED;
    if (!d.mask || !d.x_t) return LP_E_INVALID;
    if ((d.flags & LP_FL_MASK_BITS) && ((d.flags & (LP_FL_MASK_U8 | LP_FL_MASK_DENOISE)) || !aligned(d.mask, 4)))
        return LP_E_INVALID;
    const uint32_t ph = d.phases;
    if (ph == 0 || (ph & ~0x7fu)) return LP_E_INVALID;
    if (ph & LP_PH_SIGMA) {        // only on the fused replace launch of a row-table call with a bit-packed mask
        if (ph != (LP_PH_REPLACE | LP_PH_EMIT | LP_PH_COEFFS | LP_PH_SIGMA) || !(d.flags & LP_FL_MASK_BITS) || d.corr_el ||
            d.replace_kind == LP_REPLACE_KNOWN)
            return LP_E_INVALID;
        if (!d.sg_sigma || !d.sg_schedule || d.sg_schedule_len <= 0 || !d.sg_scalars_out) return LP_E_INVALID;
        if (d.es_reset) return LP_E_UNSUPPORTED;      // (the early-stop reset reads abt from t_abt, which this launch only writes)
    }
    if (d.flags & LP_FL_AV) {      // coefficient table with two rows per batch row, built by lp_coeffs (not folded), indicator as bits
        if (!d.av_bits || !aligned(d.av_bits, 4) || !d.coef || (d.flags & LP_FL_PER_ELEMENT) || (ph & (LP_PH_COEFFS | LP_PH_SIGMA)) ||
            !(d.av_frac >= 0.0f && d.av_frac <= 1.0f))
            return LP_E_INVALID;
    }
    if (d.rng_kind != LP_RNG_PHILOX && d.rng_kind != LP_RNG_TORCH) return LP_E_INVALID;
    if (d.rng_kind == LP_RNG_TORCH && (d.rng_bg == 0 || (d.rng_inc & 3u) || d.rng_inc == 0)) return LP_E_INVALID;
    if (ph & LP_PH_COEFFS) {       // only as the fused replace launch of a row-table call
        if ((ph & ~LP_PH_SIGMA) != (LP_PH_REPLACE | LP_PH_EMIT | LP_PH_COEFFS) || (d.flags & LP_FL_PER_ELEMENT)) return LP_E_INVALID;
        if (!d.coef_out) return LP_E_INVALID;
        if (!(ph & LP_PH_SIGMA)) {
            if (!d.t_abt || (!(d.flags & LP_FL_FLOW) && !d.t_ve)) return LP_E_INVALID;
            if (d.replace_kind != LP_REPLACE_KNOWN && !d.t_rsig) return LP_E_INVALID;
        }
    }
    if ((ph & LP_PH_POST_FIRST) && (ph & LP_PH_POST_STEADY)) return LP_E_INVALID;
    if ((ph & LP_PH_REPLACE) && (ph & (kPost | LP_PH_PRE_HALF))) return LP_E_INVALID;
    const bool per_el = d.flags & LP_FL_PER_ELEMENT;
    if (!per_el && !d.coef && !(ph & LP_PH_COEFFS)) return LP_E_INVALID;
    if (per_el && (!d.abt_el || (!(d.flags & LP_FL_FLOW) && !d.ve_el))) return LP_E_INVALID;
    if (ph & LP_PH_REPLACE) {
        if (!d.x) return LP_E_INVALID;
        if (d.replace_kind == LP_REPLACE_KNOWN) {
            if (!d.known) return LP_E_INVALID;
        } else if (d.replace_kind == LP_REPLACE_VE || d.replace_kind == LP_REPLACE_FLOW) {
            if (!d.noise || !d.y) return LP_E_INVALID;
            if (per_el && !d.rsig_el) return LP_E_INVALID;
        } else {
            return LP_E_INVALID;
        }
    }
    if (ph & kPost) {
        const bool given = d.flags & LP_FL_X0S_GIVEN;
        if (!d.x0 || !d.C || (!given && (!d.x0_big || !d.y))) return LP_E_INVALID;
        if ((d.flags & LP_FL_WRITE_X0S) && !d.x0s) return LP_E_INVALID;
    }
    if ((ph & LP_PH_PRE_HALF) && !d.C) return LP_E_INVALID;
    if ((ph & LP_PH_EMIT) && !d.x_in) return LP_E_INVALID;
    if (d.es_reset && !d.es) return LP_E_INVALID;
    if (d.flags & LP_FL_ES) {
        if (timer) return LP_E_UNSUPPORTED;      // an early-stop launch may be two kernels: no single event pair describes it
        if (per_el || !d.es || !d.es_partials || d.es_index < 0 || d.es_n_steps <= d.es_index) return LP_E_INVALID;
        if (!d.es_x0s[0] || !d.es_x0s[1] || !d.es_x0s[2]) return LP_E_INVALID;
        if ((d.flags & LP_FL_ES_GATED) && !d.es_xte) return LP_E_INVALID;                 // the tentative-state buffer of a gated loop
        if ((d.flags & LP_FL_ES_CLOSE) && !(d.flags & LP_FL_ES_GATED)) return LP_E_INVALID;
        if ((d.flags & LP_FL_ES_RING_BITS) && (!(d.flags & LP_FL_MASK_BITS) || !aligned(d.es_ring, 4))) return LP_E_INVALID;
    } else if (d.flags & (LP_FL_ES_GATED | LP_FL_ES_CLOSE | LP_FL_ES_RING_BITS)) {
        return LP_E_INVALID;
    }

    const size_t half_al = 8, f_al = 16;
    const bool x0_half = x0_dtype(d.flags) != DT_F32, xin_half = xin_dtype(d.flags) != DT_F32;
    const bool can_vec4 = (d.el_per_row % 4 == 0) && aligned(d.x, f_al) && aligned(d.known, f_al) &&
                          aligned(d.noise, f_al) && aligned(d.y, f_al) &&
                          aligned(d.mask, (d.flags & (LP_FL_MASK_U8 | LP_FL_MASK_BITS)) ? 4 : f_al) && aligned(d.x_t, f_al) &&
                          aligned(d.C, f_al) && aligned(d.x0s, f_al) && aligned(d.x0, x0_half ? half_al : f_al) &&
                          aligned(d.x0_big, x0_half ? half_al : f_al) && aligned(d.x_in, xin_half ? half_al : f_al) &&
                          aligned(d.xi_post, f_al) && aligned(d.xi_pre, f_al) && aligned(d.abt_el, f_al) &&
                          aligned(d.ve_el, f_al) && aligned(d.rsig_el, f_al) && aligned(d.corr_el, f_al);
    // small latents are latency bound: one element per lane puts 4x more waves on the chip
    bool vec4 = can_vec4 && d.n_el > 512 * 1024;
    if (d.tune & LP_TUNE_VEC4) vec4 = can_vec4;
    if (d.tune & LP_TUNE_VEC1) vec4 = false;
    const hipError_t err = vec4 ? launch_phase<4>(d, stream, timer) : launch_phase<1>(d, stream, timer);
    return err == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

// The replace launch as node 0 of a replayed graph: write this call's descriptor into the node's arguments.  The
// argument list is LP_STEP_ARGS of a replace launch (PH & LP_PH_REPLACE): x_t, C, x, known | noise, y, mask, row length,
// flags, the descriptor by value.  Which instantiation / grid the node runs was fixed at capture; the caller keeps
// shape, flags, phases and pointer alignment what they were (the engine's identity pre-check).
// What step_dispatch looks at when it picks the kernel and the grid of a replace launch.  A captured node keeps its kernel;
// only its arguments are rewritten per call (replace_node_update), so a rewritten descriptor has to be one the dispatcher
// would have sent to the SAME instantiation -- otherwise the captured kernel would read fields it was not compiled for
// (a phase-specialised kernel ignores host noise tensors and treats x0 as backbone heads, for one).
uint32_t replace_fingerprint(const lp_step_desc& d) {
    const uint32_t sel = LP_FL_FLOW | LP_FL_MASK_BITS | LP_FL_MASK_U8 | LP_FL_MASK_DENOISE | LP_FL_PER_ELEMENT | LP_FL_AV |
                         LP_FL_XIN_BF16 | LP_FL_XIN_F16 | LP_FL_X0S_GIVEN | LP_FL_NO_REGION_SKIP;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    const uint64_t parts[] = {d.phases, d.flags & sel, static_cast<uint64_t>(static_cast<uint32_t>(d.replace_kind)),
                              static_cast<uint64_t>(static_cast<uint32_t>(d.rng_kind)), static_cast<uint64_t>(d.n_el),
                              static_cast<uint64_t>(d.rows), static_cast<uint64_t>(d.tune), d.es ? 1ull : 0ull,
                              d.corr_el ? 1ull : 0ull};
    for (uint64_t v : parts) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); }
    return static_cast<uint32_t>(h ^ (h >> 32)) | 1u;          // never 0: 0 = "not recorded" (a binding filled in by hand)
}

int replace_node_update(const lp_step_desc* dp, hipGraphExec_t exec, const lp_graph_binding* b) {
    if (!dp || !exec || !b || !b->node || !b->func) return LP_E_INVALID;
    lp_step_desc d = *dp;
    if (!(d.phases & LP_PH_REPLACE) || !d.x || !d.x_t) return LP_E_INVALID;
    // a replace launch carries neither host noise tensors nor given x0s; and it must still be the launch that was captured
    if (d.xi_post || d.xi_pre || (d.flags & LP_FL_X0S_GIVEN)) return LP_E_INVALID;
    if (b->fingerprint != 0u && b->fingerprint != replace_fingerprint(d)) return LP_E_INVALID;
    void* a0 = d.x_t;
    void* a1 = d.C;
    const void* a2 = d.x;
    const void* a3 = d.replace_kind == LP_REPLACE_KNOWN ? static_cast<const void*>(d.known) : static_cast<const void*>(d.noise);
    const void* a4 = d.y;
    const void* a5 = d.mask;
    int32_t epr = static_cast<int32_t>(d.el_per_row);
    uint32_t fl = d.flags;
    void* args[9] = {&a0, &a1, &a2, &a3, &a4, &a5, &epr, &fl, &d};
    hipKernelNodeParams p{};
    p.func = b->func;
    p.gridDim = dim3(b->grid[0], b->grid[1], b->grid[2]);
    p.blockDim = dim3(b->block[0], b->block[1], b->block[2]);
    p.sharedMemBytes = b->shared_bytes;
    p.kernelParams = args;
    p.extra = nullptr;
    return hipGraphExecKernelNodeSetParams(exec, static_cast<hipGraphNode_t>(b->node), &p) == hipSuccess ? LP_OK : LP_E_LAUNCH;
}

int timer_create(void** out) {
    if (!out) return LP_E_INVALID;
    Timer* t = new Timer();
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
        delete t;
        return LP_E_LAUNCH;
    }
    *out = t;
    return LP_OK;
}

int timer_destroy(void* h) {
    if (!h) return LP_E_INVALID;
    Timer* t = static_cast<Timer*>(h);
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return LP_OK;
}

int timer_elapsed_ns(void* h, double* ns) {
    if (!h || !ns) return LP_E_INVALID;
    Timer* t = static_cast<Timer*>(h);
    if (hipEventSynchronize(t->stop) != hipSuccess) return LP_E_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->start, t->stop) != hipSuccess) return LP_E_LAUNCH;
    *ns = static_cast<double>(ms) * 1e6;
    return LP_OK;
}

}  // namespace lp
