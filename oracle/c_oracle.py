"""ctypes driver of oracle/langevin_oracle.c (TEST INFRASTRUCTURE ONLY).

Same call contract as `OracleLanPaint` for the per-row-scalar, B == 1 case; the
backbone stays a Python callable, every per-element operation runs in the C file."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liblanpaint_oracle.so")


class OrcRow(C.Structure):
    _fields_ = [("abt", C.c_float), ("ve_sigma", C.c_float), ("step", C.c_float), ("lambda_", C.c_float),
                ("one_plus_lambda", C.c_float), ("beta", C.c_float), ("is_flow", C.c_int)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "langevin_oracle.c")):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = C.CDLL(LIB)
        _lib.orc_nearest_exact_index.restype = C.c_int64
        _lib.orc_nearest_exact_index.argtypes = [C.c_int64] * 3
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class COracleLanPaint:
    def __init__(self, model, n_steps, lamb, beta, step_size, is_flow=False, min_step_frac=0.0, randn=None):
        self.model, self.n_steps, self.lamb, self.beta = model, n_steps, lamb, beta
        self.step_size, self.is_flow, self.min_step_frac = step_size, is_flow, min_step_frac
        self.randn = randn
        self.lib = load()

    def __call__(self, x, latent_image, noise, sigma, latent_mask, current_times, n_steps=None):
        lib = self.lib
        assert x.shape[0] == 1 and np.size(sigma) == 1
        n = x.size
        ve, abt, flow_t = (np.float32(np.ravel(t)[0]) for t in current_times)
        sig = np.float32(np.ravel(sigma)[0])
        row = OrcRow()
        row.abt, row.ve_sigma = float(abt), float(ve)
        row.step = float(np.float32(self.step_size) * np.maximum(np.float32(1) - abt, np.float32(self.min_step_frac)))
        row.lambda_, row.one_plus_lambda, row.beta, row.is_flow = self.lamb, 1.0 + self.lamb, self.beta, int(self.is_flow)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)   # noqa: E731
        y, m = f32(latent_image), f32(latent_mask)
        known = f32(sig * noise + (np.float32(1) - sig) * y) if self.is_flow else f32(y + noise * sig)
        x_t, xin = np.empty_like(y), np.empty_like(y)
        cbuf, x0s = np.empty_like(y), np.empty_like(y)
        lib.orc_replace_rescale(C.byref(row), C.c_int64(n), _p(f32(x)), _p(known), _p(m), _p(x_t))
        t_model = np.asarray([flow_t if self.is_flow else ve], dtype=np.float32)
        steps = self.n_steps if n_steps is None else n_steps
        for i in range(steps):
            if i > 0:
                lib.orc_half_step(C.byref(row), C.c_int64(n), _p(x_t), _p(m), _p(f32(self.randn(x_t))), _p(cbuf))
            lib.orc_to_model_space(C.byref(row), C.c_int64(n), _p(x_t), _p(xin))
            o = self.model(xin.copy(), t_model)
            x0, x0b = (o[0], o[1]) if isinstance(o, (tuple, list)) and len(o) >= 2 else \
                ((o[0], o[0]) if isinstance(o, (tuple, list)) else (o, o))
            fn = lib.orc_first_step if i == 0 else lib.orc_steady_post
            fn(C.byref(row), C.c_int64(n), _p(x_t), _p(f32(x0)), _p(f32(x0b)), _p(y), _p(m),
               _p(f32(self.randn(x_t))), _p(cbuf), _p(x0s))
        lib.orc_to_model_space(C.byref(row), C.c_int64(n), _p(x_t), _p(xin))
        o = self.model(xin.copy(), np.asarray([sig], dtype=np.float32))
        mo = f32(o[0] if isinstance(o, (tuple, list)) else o)
        out = np.empty_like(y)
        lib.orc_finalize(C.c_int64(n), _p(mo), _p(y), _p(m), _p(out))
        x[...] = xin
        return out


def reshape_mask_plane(src, dst_shape, taps):
    lib = load()
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.empty(dst_shape, dtype=np.float32)
    lib.orc_reshape_mask_plane(_p(src), *[C.c_int64(v) for v in src.shape], _p(dst), *[C.c_int64(v) for v in dst_shape],
                               C.c_int(taps))
    return dst
