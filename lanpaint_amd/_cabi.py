"""ctypes binding of liblanpaint_hip.so (include/lanpaint_hip.h).

This is the whole host<->native boundary: plain pointers (`tensor.data_ptr()`),
sizes and two POD descriptors.  No pybind, no ATen linkage.  The library MUST be
present: there is no CPU or PyTorch fallback for the hot path -- a missing or
stale .so raises at import of the engine.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "liblanpaint_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
ABI_VERSION = 20

# --- constants mirrored from include/lanpaint_hip.h -------------------------------
LP_OK, LP_E_INVALID, LP_E_UNSUPPORTED, LP_E_LAUNCH, LP_E_ALIGN = 0, -1, -2, -3, -4
LP_COEF_STRIDE = 36
(LP_C_SCALE, LP_C_SQRT_ABT, LP_C_OMA, LP_C_ABT, LP_C_RSIGMA, LP_C_DTX, LP_C_DTY, LP_C_AX, LP_C_AY, LP_C_DX, LP_C_DY,
 LP_C_VALID) = range(12)
LP_C_REGION0, LP_C_REGION1, LP_C_TMODEL, LP_C_RSCALE = 12, 22, 32, 33
(LP_R_E_FULL, LP_R_K_FULL, LP_R_STD_FULL, LP_R_E_HALF, LP_R_K_HALF, LP_R_STD_HALF, LP_R_DT, LP_R_A, LP_R_CX0,
 LP_R_CXT) = range(10)

LP_PH_REPLACE, LP_PH_POST_FIRST, LP_PH_POST_STEADY, LP_PH_PRE_HALF, LP_PH_EMIT, LP_PH_COEFFS, LP_PH_SIGMA = 1, 2, 4, 8, 16, 32, 64
LP_FL_FLOW, LP_FL_MASK_DENOISE, LP_FL_MASK_U8, LP_FL_WRITE_X0S = 1, 2, 4, 8
LP_FL_X0_BF16, LP_FL_X0_F16, LP_FL_XIN_BF16, LP_FL_XIN_F16 = 16, 32, 64, 128
LP_FL_PER_ELEMENT, LP_FL_X0S_GIVEN, LP_FL_CFG_FUSED, LP_FL_MASK_BITS, LP_FL_NO_REGION_SKIP = 256, 512, 1024, 2048, 4096
LP_FL_ES, LP_FL_ES_GATED, LP_FL_ES_CLOSE, LP_FL_ES_RING_BITS, LP_FL_AV = 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17
LP_TUNE_VEC1, LP_TUNE_VEC4, LP_TUNE_ES_NO_DECIDE, LP_TUNE_ES_NO_FOLD, LP_TUNE_ES_NO_ATOMICS = 1, 2, 4, 8, 16


def mask_bits_bytes(n_el: int) -> int:
    """LP_MASK_BITS_BYTES of the header."""
    return ((int(n_el) + 63) // 64) * 8
LP_REPLACE_KNOWN, LP_REPLACE_VE, LP_REPLACE_FLOW = 0, 1, 2
LP_RNG_PHILOX, LP_RNG_TORCH = 0, 1
LP_NN_ATEN_SCALAR, LP_NN_ATEN_CPU_GENERIC_FMA, LP_NN_ATEN_CPU_GENERIC = 0, 1, 2     # nearest-exact source-index rules
LP_RESHAPE_BINARIZE, LP_RESHAPE_RULE_SHIFT = 1, 8


class LpHyper(C.Structure):
    _fields_ = [("lambda_", C.c_float), ("beta", C.c_float), ("step_size", C.c_float), ("min_step_frac", C.c_float),
                ("is_flow", C.c_int32), ("one_plus_lambda", C.c_float)]


class LpStepDesc(C.Structure):
    _fields_ = [
        ("n_el", C.c_int64), ("el_per_row", C.c_int64), ("rows", C.c_int32), ("phases", C.c_uint32),
        ("flags", C.c_uint32), ("replace_kind", C.c_int32),
        ("lambda_", C.c_float), ("one_plus_lambda", C.c_float), ("beta", C.c_float), ("step_size", C.c_float),
        ("min_step_frac", C.c_float), ("noise_scale", C.c_float), ("cfg_scale", C.c_float), ("cfg_scale_big", C.c_float),
        ("coef", C.c_void_p), ("x", C.c_void_p), ("known", C.c_void_p), ("noise", C.c_void_p), ("y", C.c_void_p),
        ("mask", C.c_void_p), ("x_t", C.c_void_p), ("C", C.c_void_p), ("x0s", C.c_void_p), ("x0", C.c_void_p),
        ("x0_big", C.c_void_p), ("x_in", C.c_void_p), ("xi_post", C.c_void_p), ("xi_pre", C.c_void_p),
        ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64), ("rng_offset_ptr", C.c_void_p),
        ("abt_el", C.c_void_p), ("ve_el", C.c_void_p), ("rsig_el", C.c_void_p), ("corr_el", C.c_void_p),
        ("t_ve", C.c_void_p), ("t_abt", C.c_void_p), ("t_rsig", C.c_void_p), ("t_model", C.c_void_p),
        ("coef_out", C.c_void_p), ("t_ve_stride", C.c_int32), ("t_abt_stride", C.c_int32),
        ("t_rsig_stride", C.c_int32), ("t_model_stride", C.c_int32),
        ("rng_kind", C.c_int32), ("rng_bg", C.c_uint32), ("rng_inc", C.c_uint32),
        ("rng_state_out", C.c_void_p), ("rng_state_val", C.c_uint64 * 2),
        ("io_table_out", C.c_void_p), ("io_table_val", C.c_uint64 * 2),
        ("es", C.c_void_p), ("es_x0s", C.c_void_p * 3), ("es_ring", C.c_void_p), ("es_partials", C.c_void_p), ("es_host", C.c_void_p),
        ("es_threshold", C.c_double), ("es_seq_base", C.c_int64), ("es_patience_eff", C.c_int32), ("es_index", C.c_int32),
        ("es_n_steps", C.c_int32), ("es_reset", C.c_int32),
        ("sg_sigma", C.c_void_p), ("sg_schedule", C.c_void_p), ("sg_times_out", C.c_void_p), ("sg_scalars_out", C.c_void_p),
        ("sg_seq_out", C.c_void_p), ("sg_valid_out", C.c_void_p), ("sg_min_step_frac", C.c_double),
        ("sg_schedule_len", C.c_int32), ("sg_seq", C.c_int32), ("sg_n_steps", C.c_int32), ("sg_early_stop", C.c_int32),
        ("sg_total_steps", C.c_int32), ("sg_guess", C.c_int32), ("clk_out", C.c_void_p), ("av_bits", C.c_void_p), ("av_frac", C.c_float), ("reserved2", C.c_uint32),
        ("es_xte", C.c_void_p),
        ("tune", C.c_uint32), ("io_valid", C.c_uint32),
    ]


class LpEsState(C.Structure):
    _fields_ = [("stopped", C.c_int32), ("counter", C.c_int32), ("n_ran", C.c_int32), ("cur_slot", C.c_int32),
                ("anchor_slot", C.c_int32), ("write_slot", C.c_int32), ("reserved0", C.c_uint32), ("enabled", C.c_int32),
                ("seq_base", C.c_int64), ("total_ran", C.c_int64), ("threshold_eff", C.c_double), ("abt_val", C.c_double), ("x0s_buf", C.c_void_p * 3)]


LP_ES_SEQ_DONE, LP_ES_TRACE0 = 0x10000, 8
LP_ES_ACC_SLOTS, LP_ES_ACC_SETS = 64, 3
LP_ES_ACC_DOUBLES = LP_ES_ACC_SETS * LP_ES_ACC_SLOTS * 8


class LpFinalDesc(C.Structure):
    _fields_ = [
        ("n_el", C.c_int64), ("flags", C.c_uint32), ("cfg_scale", C.c_float),
        ("model_out", C.c_void_p), ("uncond", C.c_void_p), ("y", C.c_void_p), ("mask", C.c_void_p), ("x_src", C.c_void_p),
        ("x_dst", C.c_void_p), ("out", C.c_void_p), ("rng_bump_ptr", C.c_void_p), ("rng_bump", C.c_uint64),
        ("io_table", C.c_void_p),
    ]


class LpGraphBinding(C.Structure):
    _fields_ = [("node", C.c_void_p), ("func", C.c_void_p), ("grid", C.c_uint32 * 3), ("block", C.c_uint32 * 3),
                ("shared_bytes", C.c_uint32), ("fingerprint", C.c_uint32)]


class LpCallDesc(C.Structure):
    _fields_ = [
        ("hyper", C.POINTER(LpHyper)),
        ("ve_sigma", C.c_void_p), ("ve_stride", C.c_int32),
        ("abt", C.c_void_p), ("abt_stride", C.c_int32),
        ("replace_sigma", C.c_void_p), ("rs_stride", C.c_int32),
        ("t_model", C.c_void_p), ("t_stride", C.c_int32),
        ("rows", C.c_int32),
        ("coef_table", C.c_void_p),
        ("replace", C.POINTER(LpStepDesc)),
        ("graph_exec", C.c_void_p),
        ("final", C.POINTER(LpFinalDesc)),
        ("replace_binding", C.POINTER(LpGraphBinding)),
    ]


class LpNodeCallDesc(C.Structure):
    _fields_ = [("sigma", C.c_void_p), ("rows", C.c_int32), ("schedule_len", C.c_int32), ("schedule", C.c_void_p),
                ("is_flow", C.c_int32), ("seq", C.c_int32), ("times_out", C.c_void_p), ("scalars_out", C.c_void_p),
                ("seq_out", C.c_void_p), ("replace", C.POINTER(LpStepDesc)), ("n_steps", C.c_int32),
                ("early_stop", C.c_int32), ("total_steps", C.c_int32), ("n_counts", C.c_int32),
                ("min_step_frac", C.c_double), ("exec_by_count", C.POINTER(C.c_void_p)), ("spin_limit", C.c_int32),
                ("guess", C.c_int32), ("valid_word", C.c_void_p), ("fold_sigma", C.c_int32), ("n_eff", C.c_int32), ("launched", C.c_int32),
                ("speculated", C.c_int32), ("hit", C.c_int32), ("step_f", C.c_float), ("frac", C.c_float),
                ("full_exec_by_count", C.POINTER(C.c_void_p)), ("full_binding_by_count", C.POINTER(C.POINTER(LpGraphBinding))),
                ("one_launch", C.c_int32), ("reserved0", C.c_int32)]


class LpBlendDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32),
                ("k", C.c_int32), ("mask_batch", C.c_int32), ("mask_h", C.c_int32), ("mask_w", C.c_int32),
                ("mask", C.c_void_p), ("image1", C.c_void_p), ("image2", C.c_void_p), ("out", C.c_void_p),
                ("smooth_out", C.c_void_p), ("nn_rule", C.c_int32), ("reserved0", C.c_int32)]


EXPORTS = {
    # name: (restype, argtypes)
    "lp_abi_version": (C.c_int, []),
    "lp_strerror": (C.c_char_p, [C.c_int]),
    "lp_coeffs": (C.c_int, [C.POINTER(LpHyper), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                            C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "lp_sigma_times": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lp_sigma_times_mailbox": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_void_p]),
    "lp_step": (C.c_int, [C.POINTER(LpStepDesc), C.c_void_p]),
    "lp_finalize": (C.c_int, [C.POINTER(LpFinalDesc), C.c_void_p]),
    "lp_mask_blend": (C.c_int, [C.POINTER(LpBlendDesc), C.c_void_p]),
    "lp_timer_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "lp_timer_destroy": (C.c_int, [C.c_void_p]),
    "lp_step_timed": (C.c_int, [C.POINTER(LpStepDesc), C.c_void_p, C.c_void_p]),
    "lp_step_timed_burst": (C.c_int, [C.POINTER(LpStepDesc), C.c_void_p, C.POINTER(C.c_void_p), C.c_int32]),
    "lp_timer_elapsed_ns": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "lp_replay_burst": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(LpStepDesc), C.POINTER(LpStepDesc), C.c_int32, C.c_void_p]),
    "lp_torch_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "lp_philox_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "lp_boundary_ring": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "lp_wmse_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_int32, C.c_void_p]),
    "lp_replay_call": (C.c_int, [C.POINTER(LpCallDesc), C.c_void_p]),
    "lp_graph_bind_replace": (C.c_int, [C.c_void_p, C.POINTER(LpStepDesc), C.POINTER(LpGraphBinding)]),
    "lp_graph_clone_tail": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "lp_graph_release": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lp_graph_clone_sigma_root": (C.c_int, [C.c_void_p, C.POINTER(LpStepDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(LpGraphBinding)]),
    "lp_node_call": (C.c_int, [C.POINTER(LpNodeCallDesc), C.c_void_p]),
    "lp_effective_inner_steps": (C.c_int32, [C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double]),
    "lp_pack_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lp_pack_mask_latent": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lp_reshape_mask": (C.c_int, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p]),
}


class LanPaintHipError(RuntimeError):
    """A C-ABI call returned a negative status (the reference raises Python
    exceptions only; C codes are mapped to RuntimeError here)."""


_lib = None


def load(path: str | None = None):
    """dlopen the HIP library once; raise loudly when it is missing or stale."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("LANPAINT_AMD_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m lanpaint_amd.build`). lanpaint_amd has no CPU / PyTorch fallback for the Langevin path.")
    lib = C.CDLL(p)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    v = lib.lp_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{p}: ABI version {v}, expected {ABI_VERSION}; rebuild the extension")
    if path is None:
        _lib = lib
    return lib


def check(code: int, what: str = "lanpaint_hip"):
    if code != LP_OK:
        msg = load().lp_strerror(code)
        raise LanPaintHipError(f"{what}: {msg.decode() if msg else code} (code {code})")
