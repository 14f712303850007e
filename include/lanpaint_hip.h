/* lanpaint_hip.h -- C ABI of liblanpaint_hip.so (MI355X / gfx950).
 *
 * The reference (scraed/LanPaint v2.1.0) has NO native code and NO FFI: its hot
 * path is ~330 lines of eager PyTorch (src/LanPaint/lanpaint.py).  Every entry
 * point below therefore REPLACES a span of eager ATen ops in the reference; the
 * span is cited as file:line (relative to the reference root) on each function.
 * The binding a reference maintainer would add is a ctypes stub -- see
 * INTEGRATION.md and lanpaint_amd/_cabi.py (the one this repo ships).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / ATen types.
 *   - every function returns int: 0 = ok, <0 = LP_E_* (lp_strerror() names it).
 *     No exception crosses the boundary.
 *   - all tensor pointers are DEVICE pointers owned by the caller (torch
 *     allocations).  The library allocates no device memory, keeps no global
 *     state (it reads nothing from the environment; developer switches travel
 *     in lp_step_desc.tune) and is re-entrant.
 *   - every entry point ENQUEUES on the caller's HIP stream and returns without
 *     synchronising, so it may be called while that stream is being captured
 *     into a hipGraph -- with these exceptions, which block the calling thread
 *     and must NOT be called on a capturing stream:
 *       lp_node_call          polls a pinned-host mailbox for the device's answer
 *                             and falls back to hipStreamSynchronize after
 *                             `spin_limit` polls; it also launches hipGraphExec_t
 *                             handles (hipGraphLaunch), which a capture refuses;
 *       lp_replay_call        launches a hipGraphExec_t / updates a graph node's
 *                             arguments (host-side, not capturable);
 *       lp_timer_elapsed_ns   waits for the timed launch (hipEventSynchronize);
 *       lp_graph_*            host-side graph surgery, no stream involved.
 *     lp_timer_create / lp_graph_clone_tail allocate HOST-side runtime handles
 *     (events, a graph clone) that the caller releases with lp_timer_destroy /
 *     lp_graph_release.
 *   - tensors are dense row-major fp32 in the latent's own layout
 *     [B, C, (F), H, W] flattened: n_el = B * el_per_row.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 */
#ifndef LANPAINT_HIP_H
#define LANPAINT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LP_ABI_VERSION 20

/* The library is built with -fvisibility=hidden: the entry points declared LP_API below are its ONLY dynamic symbols (the
 * dispatch functions, kernel handles and device stubs of the C++ side stay internal; tests/test_cabi_exports.py checks
 * `nm -D --defined-only` lists nothing but lp_*). */
#define LP_API __attribute__((visibility("default")))

/* ---- error codes ------------------------------------------------------- */
#define LP_OK             0
#define LP_E_INVALID     -1   /* null / inconsistent argument                  */
#define LP_E_UNSUPPORTED -2   /* layout or flag combination not implemented   */
#define LP_E_LAUNCH      -3   /* hipLaunchKernel failed (hipGetLastError)      */
#define LP_E_ALIGN       -4   /* pointer not aligned for the requested layout */

/* ---- per-row coefficient table ------------------------------------------ */
/* One row of LP_COEF_STRIDE floats per batch row, produced on the device by
 * lp_coeffs() so the think loop needs no host<->device sync (the reference
 * recomputes these ~40 tiny ops + 1 host sync EVERY iteration:
 * lanpaint.py:205,295-328).  Region r: 0 = inpaint ("x" branch, mask==0),
 * 1 = known ("y" branch, mask==1).                                          */
#define LP_COEF_STRIDE   36
#define LP_C_SCALE        0   /* flow: sqrt(abt)+sqrt(1-abt); VE: sqrt(1+sigma^2)  (lanpaint.py:96-99) */
#define LP_C_SQRT_ABT     1
#define LP_C_OMA          2   /* 1 - abt                                         */
#define LP_C_ABT          3
#define LP_C_RSIGMA       4   /* sigma used by the replace step (lanpaint.py:94) */
#define LP_C_DTX          5   /* dt of the inpaint branch  = step                */
#define LP_C_DTY          6   /* dt of the known branch    = beta*step           */
#define LP_C_AX           7   /* 1/(1-abt)          (lanpaint.py:315,319)        */
#define LP_C_AY           8   /* (1+lambda)/(1-abt) (lanpaint.py:316,320)        */
#define LP_C_DX           9   /* sqrt(2)            (lanpaint.py:326)            */
#define LP_C_DY          10
#define LP_C_VALID       11   /* 1.0 if step > 0 else 0.0 (lanpaint.py:205)      */
#define LP_C_REGION0     12   /* 10 floats per region, see LP_R_*                */
#define LP_C_REGION1     22
#define LP_R_E_FULL       0   /* exp(-A dt)                  (lanpaint.py:242)   */
#define LP_R_K_FULL       1   /* (1-exp(-A dt))/A            (lanpaint.py:247)   */
#define LP_R_STD_FULL     2   /* sqrt(D^2 (1-exp(-2A dt))/(2A)) (lanpaint.py:249-252) */
#define LP_R_E_HALF       3   /* same three at dt/2                              */
#define LP_R_K_HALF       4
#define LP_R_STD_HALF     5
#define LP_R_DT           6
#define LP_R_A            7
#define LP_R_CX0          8   /* sqrt(abt)/(1-abt): C = CX0*x0s + CXT*x_t (lanpaint.py:219) */
#define LP_R_CXT          9   /* A - 1/(1-abt)                                   */
#define LP_C_TMODEL      32   /* the time the backbone is called with inside the loop (flow_t or VE sigma,
                                 lanpaint.py:165,170): lets a replayed hipGraph hand `table[:, 32]` to the model */
#define LP_C_RSCALE      33   /* RN(1 / scale), formed in double and rounded once (= the IEEE fp32 quotient 1.0f / scale: a
                                 double has 2 * 24 + 2 <= 53 bits, so the second rounding is innocuous).  The streaming kernels'
                                 flow-model emit x_t / scale divides four elements per lane by this row scalar: with the
                                 correctly rounded reciprocal one residual correction per element gives the IEEE quotient
                                 (lp_common.h::div_shared) -- the reciprocal itself, 11 VALU instructions per lane, now comes
                                 from the table (round 5).                                                             */

typedef struct lp_hyper {
    float    lambda;          /* LanPaint_Lambda   (lanpaint.py:11)             */
    float    beta;            /* LanPaint_Beta     (lanpaint.py:17,188-190)     */
    float    step_size;       /* LanPaint_StepSize (lanpaint.py:14,81)          */
    float    min_step_frac;   /* MinStepFrac       (lanpaint.py:18,81)          */
    int32_t  is_flow;         /* IS_FLUX or IS_FLOW (lanpaint.py:96,144,161)    */
    float    one_plus_lambda; /* fp32(1 + Lambda) with the sum taken in double,
                                 as Python evaluates `(1 + lamb)` (lanpaint.py:183,316) */
} lp_hyper;

/* ---- phases of the fused step kernel -------------------------------------- */
/* The model call is the only unavoidable cut in the loop, so one launch does
 * [everything after model call i] + [everything before model call i+1].      */
#define LP_PH_REPLACE      (1u << 0) /* x = x(1-m)+known*m; x_t = VP(x)   lanpaint.py:94-99           */
#define LP_PH_POST_FIRST   (1u << 1) /* C = coefC; x_t = OU(x_t,dt,C)     lanpaint.py:275-277         */
#define LP_PH_POST_STEADY  (1u << 2) /* C'=coefC; x_t+=(C'-C)dt; x_t=OU(x_t,dt/2,C_old); C=C'  :281-284 */
#define LP_PH_PRE_HALF     (1u << 3) /* x_t = OU(x_t,dt/2,C) -- first half of the NEXT iteration  :280  */
#define LP_PH_EMIT         (1u << 4) /* x_in = model-space(x_t)           lanpaint.py:144-147,163,168 */
#define LP_PH_COEFFS       (1u << 5) /* with LP_PH_REPLACE: this launch ALSO writes the call's coefficient table
                                        (lp_coeffs folded in: one launch less per sigma call) from the raw per-row
                                        times `t_*`; its own replace / emit use the row's scale and replace sigma
                                        computed from the same inputs.  Not with LP_FL_PER_ELEMENT.             */
#define LP_PH_SIGMA        (1u << 6) /* with LP_PH_REPLACE | LP_PH_COEFFS (bit-packed mask): the launch ALSO does what
                                        lp_sigma_times does -- sigma -> (VE sigma, abt, flow t) per row, the two scalars
                                        of the inner-step rule, the rule itself against a speculated count, the mailbox
                                        (lp_step_desc.sg_*) -- and takes its per-row times from sigma instead of t_ve /
                                        t_abt / t_rsig / t_model.  One launch less on a speculated sigma call, where the
                                        later arrival of the answer costs nothing (lp_node_call).                       */

/* ---- flags ---------------------------------------------------------------- */
#define LP_FL_FLOW          (1u << 0)  /* flow/flux VP scaling, else VE                        */
#define LP_FL_MASK_DENOISE  (1u << 1)  /* mask buffer is ComfyUI's denoise_mask: kernel applies
                                          m = 1 - (dm > 0.5)              (nodes.py:281-283)  */
#define LP_FL_MASK_U8       (1u << 2)  /* mask buffer is uint8 0/1 (1 = known)                 */
#define LP_FL_WRITE_X0S     (1u << 3)  /* also store x0s (LangevinState.x0; early stop)        */
#define LP_FL_X0_BF16       (1u << 4)  /* x0 / x0_big are bf16                                 */
#define LP_FL_X0_F16        (1u << 5)  /* x0 / x0_big are fp16                                 */
#define LP_FL_XIN_BF16      (1u << 6)  /* x_in is written as bf16                              */
#define LP_FL_XIN_F16       (1u << 7)  /* x_in is written as fp16                              */
#define LP_FL_PER_ELEMENT   (1u << 8)  /* abt_el/ve_el/... per-element times (AV packs,
                                          lanpaint.py:60-74): general path                     */
#define LP_FL_CFG_FUSED     (1u << 10) /* `x0` = cond prediction, `x0_big` = uncond prediction of ONE batched
                                          backbone pass; the kernel forms both CFG heads itself,
                                          head = uncond + (cond - uncond) * scale  (nodes.py:161-175 +
                                          ComfyUI cfg_function) instead of 2 x 3 eager elementwise passes */
#define LP_FL_MASK_BITS     (1u << 11) /* mask buffer is bit-packed latent_mask (1 = known): element i is
                                          bit (i & 31) of 32-bit word (i >> 5), as lp_pack_mask writes it;
                                          0.125 B/element instead of 4.  Not combinable with MASK_DENOISE /
                                          MASK_U8; binary masks only (SURVEY 8b `mask_kind`)            */
#define LP_FL_NO_REGION_SKIP (1u << 12) /* stream x0, x0_big and y for every element even where the bit-packed mask makes
                                          one of them unused for a whole wave (measurement / A-B switch)       */
#define LP_FL_ES            (1u << 13) /* the POST phase of this launch also evaluates the inner early-stop rule ON THE DEVICE
                                          (earlystop.py:238-336): every block forms the sums of the weighted MSEs (x0s against the
                                          previous x0s and against the drift anchor; iteration 0: x_t after against x_t before)
                                          and adds them into the accumulator set of the iteration (es_partials);
                                          lp_step then enqueues a one-wave kernel that totals the set in a fixed order, applies
                                          the threshold / patience / drift-anchor logic, updates lp_es_state and posts the
                                          trace record to `es_host` (a gated loop instead applies the rule at the top of its
                                          NEXT launch, at any grid size, and enqueues that kernel after the last launch
                                          only).  Row-table launches only (not LP_FL_PER_ELEMENT).                         */
#define LP_FL_ES_GATED      (1u << 14) /* with LP_FL_ES, a launch of a loop the host does not watch (hipGraph replay): once
                                          lp_es_state.stopped is set the launch only re-emits x_in from the committed x_t;
                                          otherwise PRE_HALF is TENTATIVE -- x_t is stored in its post-iteration state, the
                                          state after the half-step goes to es_xte and feeds x_in -- and a POST_STEADY
                                          launch starts from es_xte, so stopping after iteration i leaves exactly the
                                          reference's state after i iterations (round 3 redid the half-step from the same
                                          noise instead of storing it: a second Philox block per element).               */
#define LP_FL_ES_CLOSE      (1u << 15) /* with LP_FL_ES_GATED on the LAST launch of a loop (es_index + 1 == es_n_steps) whose
                                          verdict is folded into the launches (small grids): no closing decision kernel
                                          follows.  The launch, having applied the verdict of the iteration before, accounts
                                          its own iteration (n_ran, total_ran) and posts the call's "done" word itself; the
                                          verdict of the last iteration is never formed -- stopping after the last iteration
                                          changes nothing (earlystop.py:313 only breaks a loop that is over) -- and its trace
                                          record is not written, so a caller that wants the full trace leaves the flag off.  */
#define LP_FL_ES_RING_BITS  (1u << 16) /* with LP_FL_ES and LP_FL_MASK_BITS: es_ring is the BIT-PACKED ring (lp_pack_mask of lp_boundary_ring's
                                          output; bit = ring pixel).  With a hard mask the ring weight is 0 or 1 (ring pixels are
                                          inpaint pixels: 1 - m = 1), so the launch reads 0.125 B / element for it and derives the
                                          weight in registers.  The phase-specialised hard-mask kernels take only this form; an fp32
                                          ring next to a bit-packed mask runs through the run-time kernels.                          */
#define LP_FL_AV            (1u << 17) /* AV packs (MiniMax-H3 flat audio / video latents; lanpaint.py:60-74): every element runs on one of
                                          TWO per-row time sets, chosen by a 0/1 indicator.  `coef` then holds two rows per batch row
                                          (2 r: video times, 2 r + 1: audio times; lp_coeffs with 2 * rows inputs), `av_bits` the
                                          bit-packed indicator (1 = audio; lp_pack_mask layout) and `av_frac` the share of audio
                                          elements.  A wave whose elements all sit on one stream runs the ordinary table path on
                                          that stream's row -- no per-element transcendental, where the reference-shaped
                                          LP_FL_PER_ELEMENT form needs three full-size time tensors and evaluates exp / expm1 per
                                          element; only the wave straddling the seam of a row uses the per-element formulas.
                                          Not with LP_PH_COEFFS / LP_FL_PER_ELEMENT.  (With LP_FL_ES_GATED a stopped launch
                                          re-emits every element with its own stream's scale.)                                 */
#define LP_FL_X0S_GIVEN     (1u << 9)  /* `x0` already holds x0s = x_t + score(x_t) (public
                                          langevin_dynamics(x_t, score, ...) entry, lanpaint.py:192,218) */

/* Device-side state of the inner early stop (one per engine and device; lp_step_desc.es).  Reset by the launch
 * that carries `es_reset` (the replace launch of a sigma call), updated by the deciding block of every
 * LP_FL_ES launch.  x0s_buf: three rotating buffers for LangevinState.x0 -- the stopper compares the current one
 * with the previous one and with the drift anchor (earlystop.py:283-306), so the buffer being written is always
 * the one that is neither.                                                                                      */
typedef struct lp_es_state {
    int32_t  stopped;         /* 1 once patience_counter >= patience_eff (earlystop.py:313)              */
    int32_t  counter;         /* patience_counter                                                       */
    int32_t  n_ran;           /* iterations whose result is committed in x_t / C                        */
    int32_t  cur_slot;        /* x0s_buf index of the last committed iteration's x0s, -1 = none yet     */
    int32_t  anchor_slot;     /* x0s_buf index of the drift anchor, -1 = none                           */
    int32_t  write_slot;      /* x0s_buf index the next POST writes                                     */
    uint32_t reserved0;
    int32_t  enabled;         /* threshold_eff > 0 (earlystop.py:111-113); the zero-inpaint-weight test
                                 (:115-117) is applied when the sums are known                          */
    int64_t  seq_base;        /* mailbox sequence base of the call in flight (set at reset)             */
    int64_t  total_ran;       /* iterations committed since the state was created (never reset): lets a
                                 host that does not wait after every call account them later             */
    double   threshold_eff;   /* threshold * clamp01(4 abt (1 - abt)), abt = mean over rows             */
    double   abt_val;
    float*   x0s_buf[3];
} lp_es_state;

/* Mailbox (`es_host`, pinned host or device memory, doubles): [0] sequence word (int64 bits, stored last with a
 * system-scope release): seq_base + i + 1 after the decision of iteration i, seq_base + LP_ES_SEQ_DONE after the
 * last launch of a gated loop; [1] n_ran, [2] stopped, [3] enabled, [4] threshold_eff, [5] abt_val, [6] total_ran;
 * [LP_ES_TRACE0 + 8 i ..]: record of iteration i = { dist, dist_inpaint, dist_ring, dist_drift (NaN = not
 * evaluated), patience_counter, stopped, 0, 0 }.                                                       */
#define LP_ES_ACC_SLOTS  64
#define LP_ES_ACC_SETS   3
#define LP_ES_ACC_DOUBLES (LP_ES_ACC_SETS * LP_ES_ACC_SLOTS * 8)
#define LP_ES_SEQ_DONE   0x10000
#define LP_ES_TRACE0     8
#define LP_ES_MAILBOX_DOUBLES(n_steps) (LP_ES_TRACE0 + 8 * (n_steps))

/* in-kernel noise generators */
#define LP_RNG_PHILOX 0   /* Philox2x32-10 per element, one Box-Muller pair per launch (independent stream)  */
#define LP_RNG_TORCH  1   /* the device's torch.randn stream, reproduced exactly                             */

/* replace-step source (lanpaint.py:84-94) */
#define LP_REPLACE_KNOWN    0   /* `known` = model_sampling.noise_scaling(...) computed by the caller */
#define LP_REPLACE_VE       1   /* y + n*sigma                                                        */
#define LP_REPLACE_FLOW     2   /* sigma*(noise_scale*n) + (1-sigma)*y                                */

typedef struct lp_step_desc {
    int64_t   n_el;            /* total latent elements                                   */
    int64_t   el_per_row;      /* elements per batch row (C*(F)*H*W), < 2^31               */
    int32_t   rows;            /* B                                                       */
    uint32_t  phases;          /* LP_PH_*                                                 */
    uint32_t  flags;           /* LP_FL_*                                                 */
    int32_t   replace_kind;    /* LP_REPLACE_*                                            */
    float     lambda;          /* duplicated from lp_hyper for the per-element path       */
    float     one_plus_lambda;
    float     beta;
    float     step_size;
    float     min_step_frac;
    float     noise_scale;     /* model_sampling.noise_scale (lanpaint.py:91)             */
    float     cfg_scale;       /* LP_FL_CFG_FUSED: cond_scale      (head 0, x0)           */
    float     cfg_scale_big;   /*                  cond_scale_BIG  (head 1, x0_BIG)       */
    const float* coef;         /* [rows][LP_COEF_STRIDE] from lp_coeffs()                 */
    const float* x;            /* REPLACE: sampler latent (model space)                   */
    const float* known;        /* REPLACE, LP_REPLACE_KNOWN                               */
    const float* noise;        /* REPLACE, VE/FLOW kinds                                  */
    const float* y;            /* latent_image (clean known latent)                       */
    const void*  mask;         /* fp32 (default), u8 or bit-packed (LP_FL_MASK_*), full latent shape */
    float*       x_t;          /* VP-space state, read+written                            */
    float*       C;            /* LangevinState.C, read+written                           */
    float*       x0s;          /* LangevinState.x0 out (LP_FL_WRITE_X0S) or NULL          */
    const void*  x0;           /* model output head 0                                     */
    const void*  x0_big;       /* model output head 1 (may alias x0)                      */
    void*        x_in;         /* EMIT: model-space latent for the next model call        */
    const float* xi_post;      /* host-supplied N(0,1) for POST_* (NULL => generated in-kernel, rng_kind) */
    const float* xi_pre;       /* host-supplied N(0,1) for PRE_HALF (NULL => generated in-kernel)         */
    uint64_t     rng_seed;     /* generator seed / key                                    */
    uint64_t     rng_offset;   /* LP_RNG_PHILOX: launch sequence number (unique per launch);
                                  LP_RNG_TORCH: torch philox offset of this launch's first draw */
    const uint64_t* rng_offset_ptr; /* optional device u64[2] (graph replay): [0] is added to rng_offset;
                                  LP_RNG_TORCH also takes the seed from [1]              */
    const float* abt_el;       /* LP_FL_PER_ELEMENT: per-element abt                      */
    const float* ve_el;        /*                    per-element VE sigma                 */
    const float* rsig_el;      /*                    per-element replace sigma            */
    const float* corr_el;      /* audio_correction (lanpaint.py:173-180) or NULL          */
    /* LP_PH_COEFFS: the inputs of lp_coeffs (same meaning, same strides) and the table to write */
    const float* t_ve;         /* [rows] VE sigma (NULL for flow)                         */
    const float* t_abt;        /* [rows] abt                                              */
    const float* t_rsig;       /* [rows] replace sigma                                    */
    const float* t_model;      /* [rows] backbone time argument -> slot LP_C_TMODEL       */
    float*       coef_out;     /* [rows][LP_COEF_STRIDE]                                  */
    int32_t      t_ve_stride, t_abt_stride, t_rsig_stride, t_model_stride;   /* 0 = broadcast row 0 */
    /* In-kernel noise generator (xi_post / xi_pre NULL).  LP_RNG_TORCH reproduces, bit for bit, the values
     * `torch.randn_like(x_t)` would return on this device for generator state (rng_seed, offset): ATen's
     * Philox4x32-10 thread / offset mapping (DistributionTemplates.h) over rocRAND's own normal4.  Draw order
     * inside one launch: the POST draw, then the PRE draw (offset + rng_inc) -- the reference's order
     * (lanpaint.py:277,280,283).  rng_bg = block * grid of ATen's launch for n_el elements, rng_inc = the
     * generator-offset increment of one such call.                                                        */
    int32_t      rng_kind;     /* LP_RNG_PHILOX (0) | LP_RNG_TORCH (1)                     */
    uint32_t     rng_bg;
    uint32_t     rng_inc;
    /* LP_PH_COEFFS launches of a replayed graph publish the caller's generator state for the captured launches:
     * rng_state_out[0] = rng_state_val[0] (offset base), [1] = rng_state_val[1] (seed); NULL = nothing.      */
    uint64_t*    rng_state_out;
    uint64_t     rng_state_val[2];
    /* Per-call I/O pointers for launches CAPTURED in a hipGraph (their kernargs are frozen at capture): the first
     * lane of the launch stores io_table_val[0..1] to io_table_out[0..1] (NULL = nothing).  The engine's replace
     * launch -- the one launch of a sigma call that stays outside the graph -- publishes { address of the sampler
     * latent x (lanpaint.py:156 writes it in place), address of this call's `out` } for the captured lp_finalize
     * (lp_final_desc.io_table), so that the caller's tensors never have to be staged through static buffers.
     * io_table_out[2] is the "this sigma call is valid" word, see io_valid below.                              */
    uint64_t*    io_table_out;
    uint64_t     io_table_val[2];
    /* Inner early stop on the device (LP_FL_ES; earlystop.py:58-336 with the default metric).                 */
    lp_es_state* es;             /* device state, TWO consecutive lp_es_state (a gated loop on a small grid alternates
                                    between them); also given to the launch that resets it                     */
    float*       es_x0s[3];      /* the three rotating x0s buffers (== es->x0s_buf, as launch arguments so the kernel
                                    selects one by slot index instead of chasing a pointer through the state)     */
    const float* es_ring;        /* mask-edge ring weight (lp_boundary_ring; 4-D latents) as fp32, or bit-packed with
                                    LP_FL_ES_RING_BITS, or NULL                                              */
    double*      es_partials;    /* device scratch, LP_ES_ACC_DOUBLES doubles: the accumulator sets the blocks of an
                                    LP_FL_ES launch add their six sums into (set = es_index mod LP_ES_ACC_SETS, slot =
                                    block mod LP_ES_ACC_SLOTS).  The es_reset launch clears all of it; launch i clears
                                    the set of iteration i + 1.  Ordinary (coarse-grained) device memory: the adds are
                                    hardware fp64 atomics, which fine-grained / host-coherent allocations do not honour */
    double*      es_host;        /* mailbox, LP_ES_MAILBOX_DOUBLES(es_n_steps) doubles                       */
    double       es_threshold;   /* threshold before the abt scaling (earlystop.py:78-81)                    */
    int64_t      es_seq_base;    /* es_reset: sequence base of this call                                     */
    int32_t      es_patience_eff;/* max(1, patience) + 1 (earlystop.py:103)                                  */
    int32_t      es_index;       /* iteration i this launch's POST belongs to                                */
    int32_t      es_n_steps;     /* iterations of the loop (the last launch of a gated loop posts "done")    */
    int32_t      es_reset;       /* 1: this launch (a replace launch) initialises *es for a new call         */
    /* Profiling builds only (library compiled with -DLP_SHADER_CLOCK, scripts/shader_clock.py): thread 0 of the
     * first and of the last block store LP_CLK_STAMPS shader-clock stamps each (s_memtime ticks since its own kernel
     * entry) at clk_out[0..] and clk_out[16..], then the 100 MHz s_memrealtime span of the kernel at [15] / [31].
     * A release build ignores the field.                                                                      */
    /* LP_PH_SIGMA (see there): the arguments of lp_sigma_times_mailbox and of the inner-step rule                */
    const float* sg_sigma;       /* device [rows]                                                              */
    const float* sg_schedule;    /* device [sg_schedule_len]                                                   */
    float*       sg_times_out;   /* device [3][rows]                                                           */
    float*       sg_scalars_out; /* pinned host float[4]                                                       */
    int32_t*     sg_seq_out;     /* pinned host                                                                */
    uint64_t*    sg_valid_out;   /* device word lp_finalize checks                                             */
    double       sg_min_step_frac;
    int32_t      sg_schedule_len, sg_seq, sg_n_steps, sg_early_stop, sg_total_steps, sg_guess;
    double*      clk_out;
    const void*  av_bits;        /* LP_FL_AV: bit-packed stream indicator, 1 = audio element (LP_MASK_BITS_BYTES(n_el) bytes)          */
    float        av_frac;        /* LP_FL_AV: audio elements / all elements (the early-stop threshold uses the mean of the blended abt) */
    uint32_t     reserved2;
    float*       es_xte;         /* LP_FL_ES_GATED: n_el floats, the state after the TENTATIVE first half-step of the next
                                    iteration (lanpaint.py:280).  A gated launch stores it next to the committed x_t; the next
                                    launch of the loop starts from it, a stopped loop never looks at it again.             */
    uint32_t     tune;           /* LP_TUNE_*: developer switches of this launch (micro-benchmarks, A/B runs); 0 in production */
    uint32_t     io_valid;       /* with io_table_out: 1 = this launch also stores 1 into io_table_out[2], the word whose 0 voids a
                                    captured lp_finalize.  Every replace launch that is not part of a speculated lp_node_call sets
                                    it, so that a speculated run which voided itself and was then abandoned (an error return) cannot
                                    leave later replays voided; lp_node_call clears it on the launches it queues itself -- there
                                    the sigma rule owns the word.                                                              */
} lp_step_desc;
#define LP_TUNE_VEC1          (1u << 0)   /* one element per lane whatever the size                                          */
#define LP_TUNE_VEC4          (1u << 1)   /* four elements per lane whenever layout and alignment allow                      */
#define LP_TUNE_ES_NO_DECIDE  (1u << 2)   /* LP_FL_ES: do not enqueue the one-wave decision kernel behind the launch         */
#define LP_TUNE_ES_NO_FOLD    (1u << 3)   /* LP_FL_ES_GATED: decision kernel after every launch instead of the folded verdict */
#define LP_TUNE_ES_NO_ATOMICS (1u << 4)   /* LP_FL_ES: the blocks do not add their sums to the accumulator set (WRONG verdicts: a
                                             measurement switch that prices the atomics)                                    */
#define LP_CLK_STAMPS 8  /* 0 entry, 1 operand loads issued, 2 noise generated, 3 operands arrived, 4 stop verdict
                            formed, 5 arithmetic done, 6 stores issued, 7 per-block sums written                */

typedef struct lp_final_desc {
    int64_t   n_el;
    uint32_t  flags;           /* LP_FL_MASK_*, LP_FL_X0_BF16/F16 (dtype of model_out), LP_FL_CFG_FUSED */
    float     cfg_scale;       /* LP_FL_CFG_FUSED: head 0 = uncond + (model_out - uncond)*cfg_scale */
    const void*  model_out;    /* final denoise, head 0 (lanpaint.py:151-153); cond prediction when fused */
    const void*  uncond;       /* LP_FL_CFG_FUSED: uncond prediction, else NULL           */
    const float* y;
    const void*  mask;
    const float* x_src;        /* final model-space x (the last EMIT)                     */
    float*       x_dst;        /* sampler latent, overwritten IN PLACE (lanpaint.py:156); NULL = skip */
    float*       out;          /* out*(1-m) + y*m (lanpaint.py:154)                       */
    uint64_t*    rng_bump_ptr; /* optional: *ptr += rng_bump after the launch (graph replay) */
    uint64_t     rng_bump;
    const uint64_t* io_table;  /* optional device u64[3] (a finalize captured in a hipGraph; word 2 = 0 voids the launch --
                                  nothing written, rng_bump not applied -- and must be 1 otherwise; BOTH published addresses must
                                  be 16-byte aligned -- the launch is laid out for 16 B per lane before it can see
                                  them): x_dst and out are
                                  read from io_table[0] / io_table[1] on the device, as the replace launch of the
                                  same sigma call published them (lp_step_desc.io_table_out); the x_dst / out
                                  fields are then ignored.  A zero address in slot 0 skips the write-back.      */
} lp_final_desc;

/* ---- entry points --------------------------------------------------------- */
LP_API int lp_abi_version(void);
LP_API const char* lp_strerror(int code);

/* K1  per-row coefficient table on the device.
 * Replaces: KSamplerX0Inpaint scalars feeding LanPaint.LanPaint (lanpaint.py:81-82),
 *           prepare_step_size (lanpaint.py:295-328) and the closed-form OU
 *           factors of advance_time_overdamped (lanpaint.py:241-252).
 * ve_sigma / abt / replace_sigma: device fp32, `rows` entries each, or 1 entry
 * broadcast when the matching *_stride is 0.  step_override (nullable): explicit
 * per-row step size (the `step_size` argument of the public langevin_dynamics,
 * lanpaint.py:192) instead of StepSize*max(1-abt, MinStepFrac).  t_model (nullable):
 * copied into slot LP_C_TMODEL.                                                */
LP_API int lp_coeffs(const lp_hyper* hyper, const float* ve_sigma, int ve_stride, const float* abt, int abt_stride,
              const float* replace_sigma, int rs_stride, const float* step_override, int step_stride,
              const float* t_model, int t_stride, int rows, float* coef_table, void* stream);

/* One sigma call's whole enqueue sequence in ONE host call, for callers that replay the think loop as a
 * hipGraph (the loop between the replace step and the finalise, captured by the caller):
 *     [lp_coeffs(...)] ; lp_step(replace) ; hipGraphLaunch(graph_exec, stream) ; lp_finalize(final).
 * `hyper` NULL skips the separate lp_coeffs launch (the replace descriptor then carries LP_PH_COEFFS);
 * `final` NULL skips the lp_finalize launch (it is a node of the captured graph, lp_final_desc.io_table);
 * `replace` NULL skips the replace launch (the caller enqueued it earlier: KSamplerX0Inpaint starts a sigma call
 * before it knows the inner-step count and picks the graph afterwards).
 * Host-side launch cost matters at SDXL-latent sizes (the whole call is ~40 us of GPU time): four trips through
 * an FFI cost more than the kernels they start.  `graph_exec` is a hipGraphExec_t (NULL: skip the graph launch).
 * Stops at the first failing step and returns its code.                                              */
typedef struct lp_call_desc {
    const lp_hyper*      hyper;
    const float*         ve_sigma;      int32_t ve_stride;
    const float*         abt;           int32_t abt_stride;
    const float*         replace_sigma; int32_t rs_stride;
    const float*         t_model;       int32_t t_stride;
    int32_t              rows;
    float*               coef_table;
    const lp_step_desc*  replace;       /* LP_PH_REPLACE | LP_PH_EMIT launch, or NULL           */
    void*                graph_exec;    /* hipGraphExec_t of the captured think loop, or NULL   */
    const lp_final_desc* final;         /* lp_finalize descriptor, or NULL                      */
    const struct lp_graph_binding* replace_binding;
                                        /* non-NULL: the replace launch is the FIRST NODE of `graph_exec`
                                           (lp_graph_bind_replace): `replace` is not launched, its pointers and
                                           scalars are written into that node's arguments
                                           (hipGraphExecKernelNodeSetParams) before the graph launch -- the whole
                                           sigma call is then ONE hipGraphLaunch with nothing eager in front       */
} lp_call_desc;
LP_API int lp_replay_call(const lp_call_desc* call, void* stream);

/* One hipGraphLaunch per sigma call (round 3).  A sigma call captured WITH its replace launch (lanpaint.py:81-99,
 * the first node of the graph) needs that node's per-call arguments -- the sampler's x, the noise, sigma / times,
 * this call's `out`, generator state -- refreshed before every replay.  lp_graph_bind_replace finds the node in the
 * captured hipGraph_t (its single root; checked against the captured descriptor) and records what
 * hipGraphExecKernelNodeSetParams needs; lp_replay_call then patches instead of launching (replace_binding).
 * hipGraphExecKernelNodeSetParams only affects launches enqueued AFTER it (checked on the MI355X with the GPU held
 * busy: scripts/experiments/graph_setparams.hip), so calls may be queued back to back.
 * lp_graph_clone_tail: the same graph WITHOUT that first node, instantiated -- for a caller that enqueues the replace
 * launch early (KSamplerX0Inpaint does not know the inner-step count yet, nodes.py:286-299) and the rest afterwards.
 * lp_graph_release destroys what lp_graph_clone_tail returned.                                                  */
typedef struct lp_graph_binding {
    void*    node;                      /* hipGraphNode_t of the captured replace launch                           */
    void*    func;                      /* its kernel                                                              */
    uint32_t grid[3], block[3];
    uint32_t shared_bytes;
    uint32_t fingerprint;               /* of everything in the captured descriptor that selected this kernel instantiation
                                           and its grid (phases, the flag bits the dispatcher reads, replace_kind, rng_kind,
                                           sizes): a later argument rewrite with a descriptor that would have dispatched to
                                           ANOTHER kernel is refused (LP_E_INVALID) instead of running the captured one on
                                           arguments it does not understand                                              */
} lp_graph_binding;
LP_API int lp_graph_bind_replace(void* graph, const lp_step_desc* captured_replace, lp_graph_binding* out);
LP_API int lp_graph_clone_tail(void* graph, void** tail_graph_out, void** tail_exec_out);
LP_API int lp_graph_release(void* tail_graph, void* tail_exec);
/* For the sampler callable (KSamplerX0Inpaint, nodes.py:229-315), which needs the device's word on the inner-step count inside
 * every sigma call: a copy of the captured call `graph` (root = the replace launch, LP_PH_REPLACE | LP_PH_EMIT | LP_PH_COEFFS)
 * whose root is the SAME launch with the sigma algebra folded in (`with_sigma`: the captured replace descriptor with
 * LP_PH_SIGMA and its sg_* fields set), instantiated.  `binding_out` names the new root for lp_node_call's per-call argument
 * refresh.  Briefly captures on a private stream to learn the launch geometry: not capture-safe.  Release with
 * lp_graph_release.                                                                                                   */
LP_API int lp_graph_clone_sigma_root(void* graph, const lp_step_desc* with_sigma, void** graph_out, void** exec_out,
                                     lp_graph_binding* binding_out);

/* K1a  sigma -> (VE_sigma, abt, flow_t) per batch row plus the two scalars the inner-step rule needs,
 * in ONE launch.  Replaces the ~15 eager scalar ops + 2 host syncs of KSamplerX0Inpaint.__call__
 * (nodes.py:242-252 times, :286 argmin over the schedule, :299 mean(1 - abt)); every operation is a
 * separately rounded fp32 op in the reference's order (no FMA contraction), so n_eff decisions match.
 * times_out: [3][rows] = VE_sigma, abt, flow_t.  scalars_out: [2] = { index of the schedule entry
 * closest to mean(sigma) (first minimum), mean(1 - abt) }.                                        */
LP_API int lp_sigma_times(const float* sigma, int32_t rows, const float* schedule, int32_t schedule_len, int32_t is_flow,
                   float* times_out, float* scalars_out, void* stream);
/* Same launch with a mailbox: `scalars_out` may be device-visible PINNED HOST memory; after the two scalars the
 * kernel stores `seq` to `seq_out` with a system-scope release, so a host thread polling *seq_out learns the two
 * numbers ~2 us after the kernel ran instead of through a blocking device->host copy (the one host dependency of
 * the per-sigma inner-step rule, nodes.py:286-299: sigma exists only on the device, in stream order).       */
LP_API int lp_sigma_times_mailbox(const float* sigma, int32_t rows, const float* schedule, int32_t schedule_len, int32_t is_flow,
                           float* times_out, float* scalars_out, int32_t* seq_out, int32_t seq, void* stream);

/* The sampler-facing callable's steady state in ONE host call (round 3).  KSamplerX0Inpaint.__call__ (nodes.py:229-315)
 * has to learn from the device where sigma sits in the schedule before it can fix the inner-step count (nodes.py:286-299).
 * lp_node_call enqueues lp_sigma_times_mailbox and the replace launch of the call, polls the pinned mailbox for the two
 * scalars, applies the reference's rule
 *     n_eff = 0                                   if total_steps - step <= early_stop
 *           = n_steps                             if min_step_frac <= 0 or frac >= min_step_frac or n_steps <= 0
 *           = max(0, round_half_even(n_steps * frac / min_step_frac))        otherwise   (Python's round())
 * and launches the graph the caller captured for that count (`exec_by_count[n_eff]`: everything of the sigma call after
 * the replace launch).  Three FFI trips and the Python between them become one; the rule is evaluated in double on the
 * float32 scalars exactly as the Python expression does.  `launched` = 0 when no graph is known for the count (the
 * caller finishes the call itself; the replace launch is already enqueued).  `scalars_out` is float[4]: word 2 is the
 * sequence number (`seq_out`), word 3 the count the device computed.                                              */
typedef struct lp_node_call_desc {
    const float*        sigma;          /* device [rows]                                                          */
    int32_t             rows;
    int32_t             schedule_len;
    const float*        schedule;       /* device [schedule_len]: the sampler's sigmas                            */
    int32_t             is_flow;
    int32_t             seq;            /* sequence word this call posts                                          */
    float*              times_out;      /* device [3][rows]                                                       */
    float*              scalars_out;    /* PINNED HOST float[4]: { step index, mean(1 - abt), (seq), device n_eff } */
    int32_t*            seq_out;        /* PINNED HOST                                                            */
    const lp_step_desc* replace;        /* the call's replace launch (LP_PH_REPLACE | ...), NULL = none           */
    int32_t             n_steps;        /* PaintMethod.n_steps (LanPaint_NumSteps)                                */
    int32_t             early_stop;     /* LanPaint_EarlyStop                                                     */
    int32_t             total_steps;    /* len(sigmas) - 1                                                        */
    int32_t             n_counts;       /* entries of exec_by_count                                               */
    double              min_step_frac;  /* LanPaint_MinStepFrac                                                   */
    void* const*        exec_by_count;  /* hipGraphExec_t per inner-step count, NULL entries allowed              */
    int32_t             spin_limit;     /* polls before falling back to hipStreamSynchronize                      */
    int32_t             guess;          /* >= 0: SPECULATE -- queue exec_by_count[guess] before the device has answered;
                                           the sigma kernel evaluates the same rule, and when the true count differs it
                                           zeroes *valid_word, which voids the queued run (its lp_finalize writes
                                           nothing, lp_final_desc.io_table word 2); lp_node_call then queues the call
                                           again for the true count.  < 0: wait for the answer first              */
    uint64_t*           valid_word;     /* device word the captured lp_finalize checks (io_table + 2), or NULL: never
                                           speculate                                                              */
    int32_t             fold_sigma;     /* 1: a speculated call carries the sigma algebra inside its replace launch
                                           (LP_PH_SIGMA) instead of a kernel of its own, when the descriptor allows */
    int32_t             n_eff;          /* out                                                                    */
    int32_t             launched;       /* out: 1 = exec_by_count[n_eff] was launched                             */
    int32_t             speculated;     /* out: 1 = a run was queued for `guess`                                  */
    int32_t             hit;            /* out: 1 = ... and the guess was right                                   */
    float               step_f, frac;   /* out: the two scalars as read from the mailbox                          */
    void* const*        full_exec_by_count;
                                        /* optional, per inner-step count like exec_by_count: hipGraphExec_t of the WHOLE sigma
                                           call whose first node is the replace launch with the sigma algebra folded in
                                           (lp_graph_clone_sigma_root).  A speculated call then is ONE hipGraphLaunch: the
                                           node's arguments are refreshed from `replace` + the sigma fields of this
                                           descriptor (hipGraphExecKernelNodeSetParams), nothing is launched in front of
                                           the graph.  NULL / NULL entry: the replace launch goes eagerly in front of
                                           exec_by_count[guess] as before                                           */
    const struct lp_graph_binding* const* full_binding_by_count;   /* the root node of each full_exec_by_count entry */
    int32_t             one_launch;     /* out: 1 = the speculated call went out as one graph launch              */
    int32_t             reserved0;
} lp_node_call_desc;
LP_API int lp_node_call(lp_node_call_desc* call, void* stream);
/* the rule alone (host arithmetic; tests pin it against the reference's min_step_frac_effective_steps table)   */
LP_API int32_t lp_effective_inner_steps(int32_t n_steps, double step_f, double frac, int32_t total_steps, int32_t early_stop,
                                 double min_step_frac);

/* K0 / K_first / K2  the fused step (phases select the work).
 * Replaces: lanpaint.py:94-99 (REPLACE), :159-184 + :212-220 (score split + Coef_C),
 *           :232-254 (exact OU + noise injection), :274-286 (the scheme),
 *           :144-147 / :163 / :168 (EMIT).                                     */
LP_API int lp_step(const lp_step_desc* desc, void* stream);

/* Measurement hooks (bench.py roofline leg): lp_step_timed launches exactly like
 * lp_step but through hipExtLaunchKernelGGL with a start/stop event pair bound to the
 * dispatch itself, so lp_timer_elapsed_ns returns the kernel's own begin->end time
 * (what rocprofv3 --kernel-trace reports), not a host-side interval.  The timer is a
 * caller-owned handle; lp_timer_elapsed_ns blocks until that launch has finished.     */
LP_API int lp_timer_create(void** timer);
LP_API int lp_timer_destroy(void* timer);
LP_API int lp_step_timed(const lp_step_desc* desc, void* stream, void* timer);   /* LP_E_UNSUPPORTED for LP_FL_ES launches */
/* n timed launches of the same descriptor from one host call (rng_offset + i per launch), so the GPU stays
 * busy between them: launched one by one through an FFI the host paces a ~10 us kernel and every dispatch
 * starts on an idle chip (measured 13.0 us instead of the 10.5 us rocprofv3 reports for the same kernel). */
LP_API int lp_step_timed_burst(const lp_step_desc* desc, void* stream, void* const* timers, int32_t n);
LP_API int lp_timer_elapsed_ns(void* timer, double* ns);

/* Measurement utility, like lp_step_timed_burst (bench.py's `launch_floor`; no reference counterpart -- the reference has no
 * launch structure to measure): ONE host call enqueues `repeats` x [ for i in 0 .. n-1: `before` (an lp_step launch, NULL =
 * none), hipGraphLaunch(graph_execs[i]) (NULL entry = none), `after` (an lp_step launch, NULL = none) ] on `stream`.  With the
 * hipGraphExec_t handles of a job's captured sigma calls and an elementwise launch standing for the sampler's update between
 * them, the wall time of the burst is what the schedule costs with NO host code between the launches: the floor a
 * host-driven loop over the same graphs can reach.  Not capture-safe (drives graph handles). */
LP_API int lp_replay_burst(void* const* graph_execs, int32_t n, const lp_step_desc* before, const lp_step_desc* after,
                           int32_t repeats, void* stream);

/* K3  finalise: known-region reprojection + in-place write-back.
 * Replaces: lanpaint.py:154,156.                                               */
LP_API int lp_finalize(const lp_final_desc* desc, void* stream);

/* Standalone N(0,1) fill with the generator the fused kernel uses: Philox2x32-10, one
 * block per latent element keyed on (seed, element, launch offset), Box-Muller; slot 0 =
 * cosine branch (POST stream), 1 = sine branch (PRE stream).  Lets tests reproduce the
 * in-kernel noise exactly.  Replaces torch.randn_like (lanpaint.py:252).               */
LP_API int lp_philox_normal(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t slot, void* stream);
/* Fill `out` with what torch.randn(n_el, device=...) returns for generator state (seed, offset) on this device
 * (test hook for LP_RNG_TORCH; bg as in lp_step_desc.rng_bg).                                            */
LP_API int lp_torch_normal(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t bg, void* stream);

/* K4  inner early-stop metric (earlystop.py:32-55).
 * lp_boundary_ring: ring[i] = (mask<=0.5) & any 4-neighbour(H,W) known, as fp32
 *                   (planes = B*C images of H x W).
 * lp_wmse_pair:     acc[0..3] = { sum(w1 d^2), sum(w1), sum(w2 d^2), sum(w2) }
 *                   with d = a - b, w1 = 1 - mask (inpaint weight), w2 = ring
 *                   (NULL => acc[2..3] = 0).  acc is a device double[4]; the
 *                   reduction order is fixed, so results are deterministic.
 *                   block_scratch: device double[4 * scratch_blocks].          */
LP_API int lp_boundary_ring(const float* mask, float* ring, int64_t planes, int32_t height, int32_t width, void* stream);
LP_API int lp_wmse_pair(const float* a, const float* b, const float* mask, const float* ring, int64_t n_el,
                 double* acc, double* block_scratch, int32_t scratch_blocks, void* stream);

/* Bytes of the bit-packed form of an n_el-element mask (whole 64-bit ballot words). */
#define LP_MASK_BITS_BYTES(n_el) ((((n_el) + 63) / 64) * 8)

/* Pack a binary fp32 mask into the LP_FL_MASK_BITS layout: one wave64 ballot per 64 elements.
 * flags = 0: `mask` is latent_mask, bit = (v > 0.5); flags = LP_FL_MASK_DENOISE: `mask` is ComfyUI's
 * denoise_mask, bit = !(v > 0.5) (nodes.py:281-283 folded in).  `bits` holds LP_MASK_BITS_BYTES(n_el)
 * bytes, 8-byte aligned; tail bits are 0.  `nonbinary` (nullable, device int32) is set to 1 when an
 * input value is neither 0 nor 1 (soft mask: the packed form would not be equivalent).             */
LP_API int lp_pack_mask(const float* mask, int64_t n_el, uint32_t flags, void* bits, int32_t* nonbinary, void* stream);
/* Same launch, which also writes the fp32 latent_mask (1 = known; nodes.py:281-283 with LP_FL_MASK_DENOISE) to `latent_out`
 * (n_el floats; may be the same buffer as `mask` when flags == 0 -- the kernel declares neither pointer restrict).  One launch re-derives BOTH forms of a mask whose tensor may have been
 * rewritten in place -- what KSamplerX0Inpaint does on every sigma call for tensors that carry no version counter
 * (torch.inference_mode), where the reference recomputes the mask on every call anyway (nodes.py:277-283).             */
LP_API int lp_pack_mask_latent(const float* mask, int64_t n_el, uint32_t flags, void* bits, float* latent_out, void* stream);

/* ATen's nearest-exact source-index rules (F.interpolate(mode="nearest-exact"): nodes.py:78,88,110,125-127,1079,1278-1287).
 * "Bit-exact mask index math" means the rule of the kernel torch runs on the device the REFERENCE holds the mask on -- the
 * reference resamples before `.to(device)` (nodes.py:159-160), i.e. on the CPU tensor ComfyUI hands it -- and ATen has three
 * (every (in, out) <= 512 checked against torch 2.10 on the CPU: tests/test_oracle_properties.py); scale = float(in)/float(out):
 *   SCALAR           min(int(floorf((i + 0.5f) * scale)), in-1): torch's GPU kernels; CPU 2-D kernel when out_h + out_w <= 128;
 *                    CPU channels-last kernels with > 3 channels
 *   CPU_GENERIC_FMA  s = max(fmaf(scale, i + 0.5f, -0.5f), 0); min(int(floorf(float(double(s) + 0.5))), in-1): the CPU's
 *                    TensorIterator kernel (1-D, 3-D, 2-D with out_h + out_w > 128) as its AVX2 / AVX512 builds contract it
 *   CPU_GENERIC      the same with product and subtraction rounded separately (a CPU without FMA: ATEN_CPU_CAPABILITY=default)
 * The three agree on every down-sampling pair (pixel mask -> latent grid); up-sampling they differ on ties (2 -> 41, i = 20). */
#define LP_NN_ATEN_SCALAR          0
#define LP_NN_ATEN_CPU_GENERIC_FMA 1
#define LP_NN_ATEN_CPU_GENERIC     2
#define LP_RESHAPE_BINARIZE        1      /* lp_reshape_mask flags bit 0: also apply 1 - (v > 0.5) (nodes.py:281-283) */
#define LP_RESHAPE_RULE_SHIFT      8      /* lp_reshape_mask flags bits 8..9: one of LP_NN_ATEN_*                      */

/* K5  mask preparation (nodes.py:59-133); index math bit-for-bit with torch's nearest-exact.
 * dst[b][c][f][h][w] = max over the temporal window (video: 5 taps, -inf pad;
 * else 1 tap) of src[f_src(f+k)][h_src(h)][w_src(w)], with idx_src(i) by the LP_NN_ATEN_* rule in `flags`
 * (0 = the scalar rule: every caller of ABI <= 17 passed 0 or 1 here),
 * with dst batch b reading src batch b % src_b and dst channel c reading src
 * channel c % src_c (the reference's repeat + slice).  `flags`: LP_RESHAPE_BINARIZE | (rule << LP_RESHAPE_RULE_SHIFT). */
LP_API int lp_reshape_mask(const float* src, int32_t src_b, int32_t src_c, int32_t src_f, int32_t src_h, int32_t src_w,
                    float* dst, int32_t batch, int32_t channels, int32_t dst_f, int32_t dst_h, int32_t dst_w,
                    int32_t temporal_taps, int32_t flags, void* stream);

/* Post-decode mask blend (pixel space, once per job):
 *   m = conv2d(max_pool2d(mask, k, stride 1, pad k/2), gaussian_kernel_2d(k), pad k/2)
 *   out = image1 * (1 - m) + image2 * m
 * Replaces MaskBlend.blend_images (nodes.py:610-638) and merge_video_with_mask
 * (nodes.py:1060-1088, incl. its nearest-exact resample of a lower-resolution mask).
 * One fused launch: the mask tile (+ 2*(k/2) halo) is staged in LDS, dilated and blurred
 * separably there (the 2-D Gaussian is the outer product of its normalised 1-D profile),
 * then the NHWC images are blended.  k odd, 1..51.                                       */
typedef struct lp_blend_desc {
    int32_t batch, height, width, channels;   /* images are [batch, height, width, channels] fp32 */
    int32_t k;                                /* blend_overlap                                  */
    int32_t mask_batch;                       /* 1 = one mask frame for every image, else == batch */
    int32_t mask_h, mask_w;                   /* mask resolution (resampled nearest-exact when != image) */
    const float* mask;                        /* [mask_batch, mask_h, mask_w]                   */
    const float* image1;
    const float* image2;
    float*       out;                         /* [batch, height, width, channels]               */
    float*       smooth_out;                  /* optional [batch, height, width] smoothed mask  */
    int32_t nn_rule;                          /* LP_NN_ATEN_* rule of the mask resample (ABI 18) */
    int32_t reserved0;
} lp_blend_desc;
LP_API int lp_mask_blend(const lp_blend_desc* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LANPAINT_HIP_H */
