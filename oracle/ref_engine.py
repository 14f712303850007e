"""Loader for oracle/_ref/: the unmodified reference engine as compiled bytecode (oracle/build_ref.py).  TEST / BASELINE
INFRASTRUCTURE ONLY -- bench.py's cpu_baseline leg and tests may use it; the product package never does."""
from __future__ import annotations

import importlib
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref", "LanPaint")
_PKG = "_lanpaint_reference_engine"


def manifest():
    try:
        return json.load(open(os.path.join(REF_DIR, "MANIFEST.json")))
    except Exception:
        return None


def load_reference():
    """The reference's `LanPaint` class (src/LanPaint/lanpaint.py:7), or None when oracle/_ref is absent or was built by
    another CPython (bytecode magic mismatch)."""
    m = manifest()
    if m is None or m.get("magic") != importlib.util.MAGIC_NUMBER.hex():
        return None
    if not all(os.path.exists(os.path.join(REF_DIR, n + ".pyc")) for n in m.get("modules", {})):
        return None
    if _PKG not in sys.modules:            # a synthetic package whose path is the bytecode directory (sourceless import)
        pkg = types.ModuleType(_PKG)
        pkg.__path__ = [REF_DIR]
        pkg.__package__ = _PKG
        sys.modules[_PKG] = pkg
    try:
        return importlib.import_module(_PKG + ".lanpaint").LanPaint
    except Exception:
        return None
