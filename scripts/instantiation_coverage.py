#!/usr/bin/env python3
"""Which lp_step_kernel<VEC, MODE, PH, X0W, RNG, ST, ES> instantiations does the product library hold, and which of them
do the GPU parity tests really launch?

    # on the GPU box, with the coverage build of the library (python -m lanpaint_amd.build --trace):
    LANPAINT_AMD_LIB=build/liblanpaint_hip_trace.so LANPAINT_AMD_TRACE_FILE=gpurun_out/trace.txt python -m pytest tests -m gpu -q
    # anywhere:
    python scripts/instantiation_coverage.py gpurun_out/trace.txt profiles/r03_instantiation_coverage.json

The JSON is committed; tests/test_cabi_exports.py::test_every_step_kernel_instantiation_is_launched_by_a_gpu_test fails when
the library gains an instantiation the file does not list as launched (add a test that reaches it and regenerate, or drop
the instantiation)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lanpaint_amd", "liblanpaint_hip.so")
FIELDS = ("VEC", "MODE", "PH", "X0W", "RNG", "ST", "ES")
MODES = {"0": "row table, fp32 mask (soft values take the general branch)", "1": "per-element times", "2": "row table, bit-packed hard mask"}
PHASES = {1: "REPLACE", 2: "POST_FIRST", 4: "POST_STEADY", 8: "PRE_HALF", 16: "EMIT", 32: "COEFFS"}


def product_instantiations(lib=LIB):
    """Template argument lists of every lp_step_kernel the host library carries a launch stub for."""
    out = subprocess.run(["nm", "-C", lib], check=True, capture_output=True, text=True).stdout
    found = set()
    for m in re.finditer(r"__device_stub__lp_step_kernel<([^>]*)>", out):
        found.add(", ".join(a.strip() for a in m.group(1).split(",")))
    return found


def describe(inst):
    a = [x.strip() for x in inst.split(",")]
    ph = int(a[2].rstrip("u"))
    return {"args": inst, "VEC": int(a[0]), "MODE": MODES.get(a[1], a[1]),
            "PH": "|".join(n for b, n in PHASES.items() if ph & b) or "run-time phases",
            "X0W": {"0": "run-time dtype", "2": "bf16/fp16 heads", "4": "fp32 heads"}[a[3]],
            "RNG": {"0": "Philox2x32", "1": "torch stream", "2": "run-time"}[a[4]], "ST": a[5] == "true",
            "ES": {"0": "off", "1": "on (decision kernel follows)", "2": "on (verdict folded into the launch)"}[a[6]]}


def main():
    trace, dst = sys.argv[1], sys.argv[2]
    launched = {ln.strip() for ln in open(trace) if ln.strip()}
    have = product_instantiations()
    doc = {"_doc": "lp_step_kernel<%s> instantiations of liblanpaint_hip.so (nm) and the ones the GPU test suite launched "
                   "(coverage build of the same sources, scripts/instantiation_coverage.py)" % ", ".join(FIELDS),
           "library_bytes": os.path.getsize(LIB), "count": len(have), "launched_count": len(have & launched),
           "instantiations": [describe(i) for i in sorted(have)],
           "launched_by_gpu_tests": sorted(have & launched), "never_launched": sorted(have - launched),
           "launched_but_not_in_library": sorted(launched - have)}
    json.dump(doc, open(dst, "w"), indent=1)
    print(f"{len(have)} instantiations, {len(have & launched)} launched by the GPU tests, never launched: {len(have - launched)}")
    for i in sorted(have - launched):
        print("  never launched:", i)


if __name__ == "__main__":
    main()
