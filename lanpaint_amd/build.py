"""Build liblanpaint_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m lanpaint_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["step_kernel.hip", "aux_kernels.hip", "blend_kernel.hip", "cabi.hip"]
HEADERS = [os.path.join(CSRC, "lp_common.h"), os.path.join(CSRC, "exports.map"), os.path.join(ROOT, "include", "lanpaint_hip.h")]
OUT = os.path.join(HERE, "liblanpaint_hip.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True, shader_clock=False, trace=False):
    """`shader_clock`: the profiling variant (-DLP_SHADER_CLOCK, scripts/shader_clock.py) as build/liblanpaint_hip_clk.so;
    `trace`: the coverage variant (-DLP_TRACE_INSTANTIATIONS, scripts/instantiation_coverage.py) as
    build/liblanpaint_hip_trace.so.  The product library is never built with either."""
    if trace:
        out = os.path.join(ROOT, "build", "liblanpaint_hip_trace.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        return _compile(out, ["-DLP_TRACE_INSTANTIATIONS"], verbose)
    if shader_clock:
        out = os.path.join(ROOT, "build", "liblanpaint_hip_clk.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        return _compile(out, ["-DLP_SHADER_CLOCK"], verbose)
    if not force and not needs_build():
        return OUT
    return _compile(OUT, [], verbose)


def _compile(out, extra, verbose):
    # -ffp-contract=on: a*b+c fuses only inside one source expression (never across statements).  Needed for
    # LP_RNG_TORCH to reproduce torch.randn bit for bit (csrc/lp_common.h); it also makes the arithmetic of every
    # kernel independent of what the optimiser happens to fuse.
    # -amdgpu-kernarg-preload-count: leading scalar / pointer kernel arguments arrive in SGPRs (csrc/step_kernel.hip)
    # -fhip-fp32-correctly-rounded-divide-sqrt, -fno-fast-math: hipcc's defaults, spelled out because parity rests on them --
    # `/` and sqrtf are IEEE-rounded, and lp_common.h::div_shared equals IEEE division only while its reciprocal `1.0f / c` is
    # correctly rounded (tests/test_gpu_kernels.py::test_shared_divisor_emit_equals_ieee_division)
    # -fvisibility=hidden: the extern "C" entry points (LP_API in include/lanpaint_hip.h) are the library's only dynamic symbols
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fno-fast-math",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-mllvm", "-amdgpu-kernarg-preload-count=14",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, shader_clock="--shader-clock" in sys.argv, trace="--trace" in sys.argv)
