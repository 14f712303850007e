"""The ONE JSON line bench.py prints: bounded in size (the round-5 line grew to 23.5 KB and the driver could no longer parse
it), printed as the last thing on stdout; everything else goes to a side-car file next to it."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_LINE_BYTES = 6000
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
_LINE_FD = None


def claim_stdout():
    """ONE JSON line on stdout, nothing else: libraries write to file descriptor 1 behind Python's back (gloo prints
    "[Gloo] Rank 0 is connected to ..." there), so fd 1 is pointed at stderr for the whole run and the line goes to the
    saved descriptor."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def _shrink(v, depth=0):
    """Round floats to 6 significant digits (the line is for reading and parsing, not for bit-exact replay)."""
    if isinstance(v, float):
        return float(f"{v:.6g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _shrink(x, depth + 1) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_shrink(x, depth + 1) for x in v]
    return v


def bounded(line, limit=MAX_LINE_BYTES):
    """The line as it will be printed: floats rounded, and -- should it still exceed `limit` -- the optional blocks dropped
    one by one (never a required key), with `truncated` naming what went.  Returns the dict to serialise."""
    out = _shrink(line)
    dropped = []
    for key in ("summary_notes", "dist", "repeats", "summary", "parity_check"):
        if len(json.dumps(out)) <= limit:
            break
        if key in out and out[key] is not None:
            out[key] = None
            dropped.append(key)
    if dropped:
        out["truncated"] = dropped
    if len(json.dumps(out)) > limit:            # last resort: the strings
        for blk in ("config", "roofline", "cpu_baseline"):
            if isinstance(out.get(blk), dict):
                out[blk] = {k: (v[:120] if isinstance(v, str) else v) for k, v in out[blk].items()}
    return out


def sidecar_path(arg=None):
    return arg or os.environ.get("LANPAINT_BENCH_SIDECAR") or os.path.join(ROOT, "bench_extras.json")


def write_sidecar(path, payload):
    """Everything that does not belong on the line (full roofline blocks, per-rank reports, secondary measurements).  Best
    effort: a read-only checkout must not cost the line."""
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(payload, f, indent=1, default=str)
        os.replace(tmp, path)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except Exception as e:
        return f"not written ({type(e).__name__})"


def emit_line(line):
    """Serialise, bound and print the line as the LAST line of stdout."""
    out = bounded(line)
    data = (json.dumps(out) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)
    return out
