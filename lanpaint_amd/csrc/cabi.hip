// cabi.hip -- extern "C" surface of liblanpaint_hip.so (declared in include/lanpaint_hip.h).
// Plain pointers and sizes in, int status out; no device allocation, no global state, nothing read from the environment.
// Every entry point enqueues on the caller's stream and returns; the ones that block the calling thread or drive graph
// handles from the host (lp_node_call, lp_replay_call, lp_timer_elapsed_ns, lp_graph_*) are listed in the header's
// conventions block and are not capture-safe.
#include <cmath>

#include "lp_common.h"

namespace lp {
int step_dispatch(const lp_step_desc* d, hipStream_t stream, void* timer);
int replace_node_update(const lp_step_desc* d, hipGraphExec_t exec, const lp_graph_binding* b);
uint32_t replace_fingerprint(const lp_step_desc& d);
int timer_create(void** out);
int timer_destroy(void* h);
int timer_elapsed_ns(void* h, double* ns);
int coeffs_dispatch(const lp_hyper* h, const float* ve, int ve_stride, const float* abt, int abt_stride,
                    const float* rs, int rs_stride, const float* step_ov, int step_stride, const float* t_model,
                    int t_stride, int rows, float* table, hipStream_t stream);
int finalize_dispatch(const lp_final_desc* d, hipStream_t stream);
int sigma_times_dispatch(const float* sigma, int rows, const float* schedule, int schedule_len, int is_flow, float* times,
                         float* scalars, int32_t* seq_out, int32_t seq, hipStream_t stream);
int sigma_times_rule_dispatch(const float* sigma, int rows, const float* schedule, int schedule_len, int is_flow,
                              float* times, float* scalars, int32_t* seq_out, int32_t seq, int32_t n_steps, int32_t early_stop,
                              int32_t total_steps, double min_step_frac, int32_t guess, uint64_t* valid_out, hipStream_t stream);
int blend_dispatch(const lp_blend_desc* d, hipStream_t stream);
int philox_dispatch(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t slot, hipStream_t stream);
int ring_dispatch(const float* mask, float* ring, int64_t planes, int height, int width, hipStream_t stream);
int wmse_dispatch(const float* a, const float* b, const float* mask, const float* ring, int64_t n_el, double* acc,
                  double* scratch, int scratch_blocks, hipStream_t stream);
int torch_normal_dispatch(float* out, int64_t n, uint64_t seed, uint64_t offset, uint32_t bg, hipStream_t stream);
int pack_mask_dispatch(const float* mask, int64_t n_el, uint32_t flags, void* bits, int32_t* nonbinary, float* latent_out,
                       hipStream_t stream);
int reshape_mask_dispatch(const float* src, int sb, int sc, int sf, int sh, int sw, float* dst, int db, int dc, int df,
                          int dh, int dw, int taps, int flags, hipStream_t stream);
}  // namespace lp

static inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

extern "C" {

int lp_abi_version(void) { return LP_ABI_VERSION; }

const char* lp_strerror(int code) {
    switch (code) {
        case LP_OK: return "ok";
        case LP_E_INVALID: return "invalid argument (null pointer or inconsistent sizes/phases)";
        case LP_E_UNSUPPORTED: return "unsupported layout or flag combination";
        case LP_E_LAUNCH: return "HIP kernel launch failed";
        case LP_E_ALIGN: return "pointer not aligned for the requested layout";
        default: return "unknown lanpaint_hip error code";
    }
}

int lp_coeffs(const lp_hyper* hyper, const float* ve_sigma, int ve_stride, const float* abt, int abt_stride,
              const float* replace_sigma, int rs_stride, const float* step_override, int step_stride,
              const float* t_model, int t_stride, int rows, float* coef_table, void* stream) {
    return lp::coeffs_dispatch(hyper, ve_sigma, ve_stride, abt, abt_stride, replace_sigma, rs_stride, step_override,
                               step_stride, t_model, t_stride, rows, coef_table, as_stream(stream));
}

int lp_sigma_times(const float* sigma, int32_t rows, const float* schedule, int32_t schedule_len, int32_t is_flow,
                   float* times_out, float* scalars_out, void* stream) {
    return lp::sigma_times_dispatch(sigma, rows, schedule, schedule_len, is_flow, times_out, scalars_out, nullptr, 0,
                                    as_stream(stream));
}

int lp_sigma_times_mailbox(const float* sigma, int32_t rows, const float* schedule, int32_t schedule_len, int32_t is_flow,
                           float* times_out, float* scalars_out, int32_t* seq_out, int32_t seq, void* stream) {
    if (!seq_out) return LP_E_INVALID;
    return lp::sigma_times_dispatch(sigma, rows, schedule, schedule_len, is_flow, times_out, scalars_out, seq_out, seq,
                                    as_stream(stream));
}

int lp_step(const lp_step_desc* desc, void* stream) { return lp::step_dispatch(desc, as_stream(stream), nullptr); }

int lp_timer_create(void** timer) { return lp::timer_create(timer); }
int lp_timer_destroy(void* timer) { return lp::timer_destroy(timer); }
int lp_step_timed(const lp_step_desc* desc, void* stream, void* timer) {
    if (!timer) return LP_E_INVALID;
    return lp::step_dispatch(desc, as_stream(stream), timer);
}
int lp_timer_elapsed_ns(void* timer, double* ns) { return lp::timer_elapsed_ns(timer, ns); }

int lp_mask_blend(const lp_blend_desc* desc, void* stream) { return lp::blend_dispatch(desc, as_stream(stream)); }


int lp_finalize(const lp_final_desc* desc, void* stream) { return lp::finalize_dispatch(desc, as_stream(stream)); }

int lp_philox_normal(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t slot, void* stream) {
    return lp::philox_dispatch(out, n_el, seed, offset, slot, as_stream(stream));
}

int lp_boundary_ring(const float* mask, float* ring, int64_t planes, int32_t height, int32_t width, void* stream) {
    return lp::ring_dispatch(mask, ring, planes, height, width, as_stream(stream));
}

int lp_wmse_pair(const float* a, const float* b, const float* mask, const float* ring, int64_t n_el, double* acc,
                 double* block_scratch, int32_t scratch_blocks, void* stream) {
    return lp::wmse_dispatch(a, b, mask, ring, n_el, acc, block_scratch, scratch_blocks, as_stream(stream));
}

int lp_replay_call(const lp_call_desc* c, void* stream) {
    if (!c || (!c->replace && !c->graph_exec && !c->final)) return LP_E_INVALID;    // nothing to enqueue
    hipStream_t s = as_stream(stream);
    int rc = LP_OK;
    if (c->hyper)       // NULL: the replace launch carries LP_PH_COEFFS and writes the table itself
        rc = lp::coeffs_dispatch(c->hyper, c->ve_sigma, c->ve_stride, c->abt, c->abt_stride, c->replace_sigma,
                                 c->rs_stride, nullptr, 0, c->t_model, c->t_stride, c->rows, c->coef_table, s);
    if (rc != LP_OK) return rc;
    if (c->replace && c->replace_binding) {     // the replace launch is node 0 of the graph: refresh its arguments
        if (!c->graph_exec) return LP_E_INVALID;
        rc = lp::replace_node_update(c->replace, static_cast<hipGraphExec_t>(c->graph_exec), c->replace_binding);
    } else if (c->replace) {
        rc = lp::step_dispatch(c->replace, s, nullptr);
    }
    if (rc != LP_OK) return rc;
    if (c->graph_exec && hipGraphLaunch(static_cast<hipGraphExec_t>(c->graph_exec), s) != hipSuccess) return LP_E_LAUNCH;
    return c->final ? lp::finalize_dispatch(c->final, s) : LP_OK;    // NULL: lp_finalize is a node of the graph
}

int lp_graph_bind_replace(void* graph, const lp_step_desc* captured, lp_graph_binding* out) {
    if (!graph || !captured || !out) return LP_E_INVALID;
    hipGraph_t g = static_cast<hipGraph_t>(graph);
    size_t n_root = 0;
    if (hipGraphGetRootNodes(g, nullptr, &n_root) != hipSuccess || n_root != 1) return LP_E_UNSUPPORTED;
    hipGraphNode_t root = nullptr;
    if (hipGraphGetRootNodes(g, &root, &n_root) != hipSuccess || !root) return LP_E_UNSUPPORTED;
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(root, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) return LP_E_UNSUPPORTED;
    hipKernelNodeParams p{};
    if (hipGraphKernelNodeGetParams(root, &p) != hipSuccess || !p.func || !p.kernelParams) return LP_E_UNSUPPORTED;
    // the node must be the launch of `captured`: leading arguments x_t, C, x (LP_STEP_ARGS of a replace launch) and
    // the descriptor it carries by value
    void* const* kp = p.kernelParams;
    const lp_step_desc* dn = static_cast<const lp_step_desc*>(kp[8]);
    if (*static_cast<void* const*>(kp[0]) != captured->x_t || *static_cast<const void* const*>(kp[2]) != captured->x ||
        !dn || dn->phases != captured->phases || !(dn->phases & LP_PH_REPLACE) || dn->n_el != captured->n_el)
        return LP_E_UNSUPPORTED;
    out->node = root; out->func = p.func;
    out->grid[0] = p.gridDim.x; out->grid[1] = p.gridDim.y; out->grid[2] = p.gridDim.z;
    out->block[0] = p.blockDim.x; out->block[1] = p.blockDim.y; out->block[2] = p.blockDim.z;
    out->shared_bytes = p.sharedMemBytes; out->fingerprint = lp::replace_fingerprint(*captured);
    return LP_OK;
}

int lp_graph_clone_tail(void* graph, void** tail_graph_out, void** tail_exec_out) {
    if (!graph || !tail_graph_out || !tail_exec_out) return LP_E_INVALID;
    hipGraph_t clone = nullptr;
    if (hipGraphClone(&clone, static_cast<hipGraph_t>(graph)) != hipSuccess) return LP_E_LAUNCH;
    size_t n_root = 0;
    hipGraphNode_t root = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipGraphGetRootNodes(clone, nullptr, &n_root) == hipSuccess && n_root == 1 &&
        hipGraphGetRootNodes(clone, &root, &n_root) == hipSuccess && root && hipGraphDestroyNode(root) == hipSuccess &&
        hipGraphInstantiate(&exec, clone, nullptr, nullptr, 0) == hipSuccess) {
        *tail_graph_out = clone; *tail_exec_out = exec;
        return LP_OK;
    }
    (void)hipGraphDestroy(clone);
    return LP_E_UNSUPPORTED;
}

int lp_graph_clone_sigma_root(void* graph, const lp_step_desc* with_sigma, void** graph_out, void** exec_out,
                              lp_graph_binding* binding_out) {
    if (!graph || !with_sigma || !graph_out || !exec_out || !binding_out) return LP_E_INVALID;
    if (with_sigma->phases != (LP_PH_REPLACE | LP_PH_EMIT | LP_PH_COEFFS | LP_PH_SIGMA)) return LP_E_INVALID;
    // (1) the launch geometry and kernel of the sigma-folded replace launch: captured once on a private stream (the dispatcher
    //     is the only place that knows which instantiation and grid a descriptor maps to)
    hipStream_t tmp = nullptr;
    if (hipStreamCreateWithFlags(&tmp, hipStreamNonBlocking) != hipSuccess) return LP_E_LAUNCH;
    hipGraph_t probe = nullptr;
    int rc = LP_E_UNSUPPORTED;
    hipGraph_t clone = nullptr;
    hipGraphExec_t exec = nullptr;
    do {
        if (hipStreamBeginCapture(tmp, hipStreamCaptureModeThreadLocal) != hipSuccess) break;
        const int rc_launch = lp::step_dispatch(with_sigma, tmp, nullptr);
        const hipError_t end = hipStreamEndCapture(tmp, &probe);
        if (rc_launch != LP_OK) { rc = rc_launch; break; }
        if (end != hipSuccess || !probe) break;
        size_t n_root = 0;
        hipGraphNode_t probe_root = nullptr;
        if (hipGraphGetRootNodes(probe, nullptr, &n_root) != hipSuccess || n_root != 1 ||
            hipGraphGetRootNodes(probe, &probe_root, &n_root) != hipSuccess || !probe_root) break;
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(probe_root, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) break;
        hipKernelNodeParams p{};
        if (hipGraphKernelNodeGetParams(probe_root, &p) != hipSuccess || !p.func) break;
        // (2) the captured call with its root exchanged for that launch
        if (hipGraphClone(&clone, static_cast<hipGraph_t>(graph)) != hipSuccess) { clone = nullptr; break; }
        hipGraphNode_t root = nullptr;
        n_root = 0;
        if (hipGraphGetRootNodes(clone, nullptr, &n_root) != hipSuccess || n_root != 1 ||
            hipGraphGetRootNodes(clone, &root, &n_root) != hipSuccess || !root) break;
        if (hipGraphNodeGetType(root, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) break;
        // the root keeps its place in the graph (same node, same edges, same insertion order: a node added afterwards and wired
        // in by hand made the runtime leave its pre-recorded-packet path -- every node of every launch then went through the
        // ordinary dispatch, 3.4 us of host time each); only what it launches changes
        if (hipGraphKernelNodeSetParams(root, &p) != hipSuccess) break;
        hipGraphNode_t fresh = root;
        if (hipGraphInstantiate(&exec, clone, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; break; }
        binding_out->node = fresh; binding_out->func = p.func;
        binding_out->grid[0] = p.gridDim.x; binding_out->grid[1] = p.gridDim.y; binding_out->grid[2] = p.gridDim.z;
        binding_out->block[0] = p.blockDim.x; binding_out->block[1] = p.blockDim.y; binding_out->block[2] = p.blockDim.z;
        binding_out->shared_bytes = p.sharedMemBytes;
        binding_out->fingerprint = lp::replace_fingerprint(*with_sigma);
        *graph_out = clone; *exec_out = exec;
        rc = LP_OK;
    } while (false);
    if (probe) (void)hipGraphDestroy(probe);
    (void)hipStreamDestroy(tmp);
    if (rc != LP_OK) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (clone) (void)hipGraphDestroy(clone);
    }
    return rc;
}

int lp_graph_release(void* tail_graph, void* tail_exec) {
    if (tail_exec) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(tail_exec));
    if (tail_graph) (void)hipGraphDestroy(static_cast<hipGraph_t>(tail_graph));
    return LP_OK;
}

int32_t lp_effective_inner_steps(int32_t n_steps, double step_f, double frac, int32_t total_steps, int32_t early_stop,
                                 double min_step_frac) {
    return lp::effective_inner_steps(n_steps, step_f, frac, total_steps, early_stop, min_step_frac);
}

static int node_wait(const lp_node_call_desc* c, int32_t want, hipStream_t s) {
    for (int32_t k = 0; k < c->spin_limit; ++k)
        if (__atomic_load_n(c->seq_out, __ATOMIC_ACQUIRE) == want) return LP_OK;
    // a long backlog in front of the kernel: sleep on the stream instead of spinning
    if (hipStreamSynchronize(s) != hipSuccess) return LP_E_LAUNCH;
    return __atomic_load_n(c->seq_out, __ATOMIC_ACQUIRE) == want ? LP_OK : LP_E_LAUNCH;
}

// lp_node_call proper.  `*queued_spec` is set once a speculated run (which may have voided itself) is in the queue.
static int node_call_body(lp_node_call_desc* c, hipStream_t s, bool* queued_spec) {
    const auto exec_for = [&](int32_t n) -> hipGraphExec_t {
        return (c->exec_by_count && n >= 0 && n < c->n_counts) ? static_cast<hipGraphExec_t>(c->exec_by_count[n]) : nullptr;
    };
    // Speculation: with a guess for the count and the device word the captured lp_finalize checks, the WHOLE call is queued
    // before the device has said anything; the sigma kernel compares the guess with the true count and voids the run on a miss.
    // (Not with the inner early stop captured: a voided run would still post to its host mailbox.)
    const bool speculate = c->guess >= 0 && c->valid_word && exec_for(c->guess) != nullptr && c->replace && !c->replace->es_reset;
    c->speculated = speculate ? 1 : 0;
    c->hit = 0;
    c->launched = 0;
    int rc = LP_OK;
    // the replace launches queued here leave the "valid" word to the sigma rule (lp_step_desc.io_valid)
    lp_step_desc rep;
    if (c->replace) {
        rep = *c->replace;
        rep.io_valid = 0;
    }
    // On a speculated call the answer is only a confirmation, so its arriving a little later costs nothing: the sigma
    // algebra rides in the replace launch (LP_PH_SIGMA) instead of a launch of its own.  Otherwise the small kernel goes
    // first -- its answer is what the host is waiting for.
    const bool fold_sigma = speculate && c->replace->phases == (LP_PH_REPLACE | LP_PH_EMIT | LP_PH_COEFFS) &&
                            (c->replace->flags & LP_FL_MASK_BITS) && !c->replace->corr_el && !c->replace->es_reset &&
                            c->replace->replace_kind != LP_REPLACE_KNOWN && c->is_flow == ((c->replace->flags & LP_FL_FLOW) ? 1 : 0) &&
                            c->rows == c->replace->rows && c->fold_sigma;
    c->one_launch = 0;
    if (fold_sigma) {
        lp_step_desc d = rep;
        d.phases |= LP_PH_SIGMA;
        d.sg_sigma = c->sigma; d.sg_schedule = c->schedule; d.sg_schedule_len = c->schedule_len; d.sg_times_out = c->times_out;
        d.sg_scalars_out = c->scalars_out; d.sg_seq_out = c->seq_out; d.sg_seq = c->seq; d.sg_valid_out = c->valid_word;
        d.sg_n_steps = c->n_steps; d.sg_early_stop = c->early_stop; d.sg_total_steps = c->total_steps; d.sg_guess = c->guess;
        d.sg_min_step_frac = c->min_step_frac;
        // the whole call as ONE graph launch when a copy of the captured call with this very launch as its root exists for the
        // guessed count: refresh the root's arguments, launch, done -- nothing eager in front of the graph
        hipGraphExec_t full = (c->full_exec_by_count && c->full_binding_by_count && c->guess < c->n_counts)
                                  ? static_cast<hipGraphExec_t>(c->full_exec_by_count[c->guess]) : nullptr;
        const lp_graph_binding* fb = full ? c->full_binding_by_count[c->guess] : nullptr;
        if (full && fb) {
            rc = lp::replace_node_update(&d, full, fb);
            if (rc != LP_OK) return rc;
            if (hipGraphLaunch(full, s) != hipSuccess) return LP_E_LAUNCH;
            *queued_spec = true;
            c->one_launch = 1;
        } else {
            rc = lp::step_dispatch(&d, s, nullptr);
            if (rc != LP_OK) return rc;
            *queued_spec = true;
        }
    } else {
        rc = lp::sigma_times_rule_dispatch(c->sigma, c->rows, c->schedule, c->schedule_len, c->is_flow, c->times_out,
                                           c->scalars_out, c->seq_out, c->seq, c->n_steps, c->early_stop, c->total_steps,
                                           c->min_step_frac, speculate ? c->guess : -1, c->valid_word, s);
        if (rc != LP_OK) return rc;
        *queued_spec = speculate;
        if (c->replace) {                    // the part of the call that does not depend on the answer: queued before the wait
            rc = lp::step_dispatch(&rep, s, nullptr);
            if (rc != LP_OK) return rc;
        }
    }
    if (speculate && !c->one_launch && hipGraphLaunch(exec_for(c->guess), s) != hipSuccess) return LP_E_LAUNCH;
    rc = node_wait(c, c->seq, s);
    if (rc != LP_OK) return rc;
    c->step_f = c->scalars_out[0];
    c->frac = c->scalars_out[1];
    c->n_eff = lp::effective_inner_steps(c->n_steps, static_cast<double>(c->step_f), static_cast<double>(c->frac),
                                         c->total_steps, c->early_stop, c->min_step_frac);
    if (c->valid_word && static_cast<int32_t>(c->scalars_out[3]) != c->n_eff) return LP_E_UNSUPPORTED;   // host and device rule disagree
    if (speculate) {
        if (c->n_eff == c->guess) {          // the queued run is the call
            c->hit = 1;
            c->launched = 1;
            *queued_spec = false;            // (the word is 1: the rule confirmed the guess)
            return LP_OK;
        }
        // a miss: the queued run voided itself.  Queue the call again, unconditionally valid this time.
        rc = lp::sigma_times_rule_dispatch(c->sigma, c->rows, c->schedule, c->schedule_len, c->is_flow, c->times_out,
                                           c->scalars_out, c->seq_out, c->seq ^ 0x40000000, c->n_steps, c->early_stop,
                                           c->total_steps, c->min_step_frac, -1, c->valid_word, s);
        if (rc != LP_OK) return rc;
        *queued_spec = false;                // the word is 1 again from here on
        rc = lp::step_dispatch(&rep, s, nullptr);
        if (rc != LP_OK) return rc;
    }
    if (hipGraphExec_t e = exec_for(c->n_eff)) {
        if (hipGraphLaunch(e, s) != hipSuccess) return LP_E_LAUNCH;
        c->launched = 1;
    }
    return LP_OK;
}

int lp_node_call(lp_node_call_desc* c, void* stream) {
    if (!c || !c->sigma || !c->schedule || !c->times_out || !c->scalars_out || !c->seq_out || c->rows <= 0) return LP_E_INVALID;
    hipStream_t s = as_stream(stream);
    bool queued_spec = false;
    const int rc = node_call_body(c, s, &queued_spec);
    if (rc != LP_OK && queued_spec && c->valid_word) {
        // An error return with a speculated run in the queue: that run may have zeroed the word the captured lp_finalize
        // checks, and nothing after it is going to set it again -- every later replay on this device would be voided
        // silently.  Put the word back (best effort: the rule kernel with no guess stores 1; its mailbox sequence number is
        // one nobody waits for).  The engine's own replace launches also store 1 (lp_step_desc.io_valid), so the next
        // ordinary sigma call heals the word even if this launch fails too.
        (void)lp::sigma_times_rule_dispatch(c->sigma, c->rows, c->schedule, c->schedule_len, c->is_flow, c->times_out,
                                            c->scalars_out, c->seq_out, c->seq ^ 0x20000000, c->n_steps, c->early_stop,
                                            c->total_steps, c->min_step_frac, -1, c->valid_word, s);
    }
    return rc;
}

int lp_step_timed_burst(const lp_step_desc* desc, void* stream, void* const* timers, int32_t n) {
    if (!desc || !timers || n <= 0) return LP_E_INVALID;
    lp_step_desc d = *desc;
    for (int32_t i = 0; i < n; ++i) {
        d.rng_offset = desc->rng_offset + static_cast<uint64_t>(i);
        const int rc = lp::step_dispatch(&d, as_stream(stream), timers[i]);
        if (rc != LP_OK) return rc;
    }
    return LP_OK;
}

int lp_replay_burst(void* const* graph_execs, int32_t n, const lp_step_desc* before, const lp_step_desc* after,
                    int32_t repeats, void* stream) {
    if (!graph_execs || n <= 0 || repeats <= 0) return LP_E_INVALID;
    hipStream_t s = as_stream(stream);
    for (int32_t r = 0; r < repeats; ++r) {
        for (int32_t i = 0; i < n; ++i) {
            if (before) {
                const int rc = lp::step_dispatch(before, s, nullptr);
                if (rc != LP_OK) return rc;
            }
            if (graph_execs[i] && hipGraphLaunch(static_cast<hipGraphExec_t>(graph_execs[i]), s) != hipSuccess) return LP_E_LAUNCH;
            if (after) {
                const int rc = lp::step_dispatch(after, s, nullptr);
                if (rc != LP_OK) return rc;
            }
        }
    }
    return LP_OK;
}

int lp_torch_normal(float* out, int64_t n_el, uint64_t seed, uint64_t offset, uint32_t bg, void* stream) {
    return lp::torch_normal_dispatch(out, n_el, seed, offset, bg, as_stream(stream));
}

int lp_pack_mask(const float* mask, int64_t n_el, uint32_t flags, void* bits, int32_t* nonbinary, void* stream) {
    return lp::pack_mask_dispatch(mask, n_el, flags, bits, nonbinary, nullptr, as_stream(stream));
}

int lp_pack_mask_latent(const float* mask, int64_t n_el, uint32_t flags, void* bits, float* latent_out, void* stream) {
    if (!latent_out) return LP_E_INVALID;
    return lp::pack_mask_dispatch(mask, n_el, flags, bits, nullptr, latent_out, as_stream(stream));
}

int lp_reshape_mask(const float* src, int32_t src_b, int32_t src_c, int32_t src_f, int32_t src_h, int32_t src_w,
                    float* dst, int32_t batch, int32_t channels, int32_t dst_f, int32_t dst_h, int32_t dst_w,
                    int32_t temporal_taps, int32_t flags, void* stream) {
    return lp::reshape_mask_dispatch(src, src_b, src_c, src_f, src_h, src_w, dst, batch, channels, dst_f, dst_h, dst_w,
                                     temporal_taps, flags, as_stream(stream));
}

}  // extern "C"
