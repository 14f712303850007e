#!/usr/bin/env python3
"""Build recipe for oracle/_ref/: the UNMODIFIED reference engine, compiled -- not copied.

    python oracle/build_ref.py            (also: make -C oracle ref; __graft_entry__.build() runs it)

The reference's hot path is pure Python (src/LanPaint/{lanpaint,types,earlystop}.py: nothing to hand to gcc), so its
"compiled form" is CPython bytecode: each source is compiled WHERE IT LIES under /root/reference with py_compile and only
the resulting .pyc -- a build output, like a .so -- lands in oracle/_ref/LanPaint/.  No reference source text enters this
repository or its history: oracle/_ref/ is git-ignored (and NOT gpurun-ignored, so the binaries travel to the GPU box with
the product's own .so).  bench.py's cpu_baseline leg times this engine (kind "reference") on the GPU box's host cores;
nothing under lanpaint_amd/ may import it (tests/test_nodes_host.py::test_product_package_never_imports_the_oracle).

Without /root/reference (the GPU box, a user's checkout) the recipe does nothing and whatever is already staged stays.
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src/LanPaint"
MODULES = ("lanpaint", "types", "earlystop")          # the engine and the two modules it imports (lanpaint.py:1-5)
OUT = os.path.join(HERE, "_ref", "LanPaint")


def build(verbose=True):
    if not os.path.isdir(REF_SRC):
        if verbose:
            print(f"oracle/build_ref.py: {REF_SRC} not present; nothing to do")
        return None
    os.makedirs(OUT, exist_ok=True)
    manifest = {"python": sys.version.split()[0], "magic": importlib.util.MAGIC_NUMBER.hex(), "source_root": REF_SRC, "modules": {}}
    for m in MODULES:
        src = os.path.join(REF_SRC, m + ".py")
        dst = os.path.join(OUT, m + ".pyc")
        py_compile.compile(src, cfile=dst, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest["modules"][m] = {"source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
                                  "pyc_bytes": os.path.getsize(dst)}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    if verbose:
        print(f"oracle/_ref: reference engine compiled to {OUT} ({', '.join(MODULES)})")
    return OUT


if __name__ == "__main__":
    build()
