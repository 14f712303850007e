"""The ONE JSON line bench.py prints: bounded in size (the round-5 line grew to 23.5 KB and the driver could no longer parse
it), printed as the last thing on stdout; everything else goes to a side-car file next to it."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

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


# ---------------------------------------------------------------- what goes on the line
def engine_keywords(args):
    """Keyword arguments of the timed engine: none at all in the drop-in configuration."""
    kw = {}
    if args.rng is not None:
        kw["rng"] = args.rng
    if args.graph is not None:
        kw["graph"] = bool(args.graph)
    if args.model_dtype == "bf16":
        kw["model_dtype"] = torch.bfloat16
    return kw


def pci_bus_id(index):
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def build_line(args, shape, n_sig, n_think, flow, value, tmax, parity, steady, cpu, summary, dist_info, captured_calls=None,
               mask_packed=None):
    """The headline line from the measurements of a run (`bounded` / `emit_line` above bound and print it)."""
    from .cpu import compact_cpu
    from .ranks import summarise_dist
    from .roofline import compact_roofline
    from .workloads import mask_description
    n_el = int(np.prod(shape))
    rng_name = args.rng or "torch"
    kw = engine_keywords(args)
    parity_failed = (parity is not None and not parity.get("ok")) or (dist_info is not None and not dist_info.get("parity_ok_all_ranks", True))
    line = {
        "metric": "langevin_think_iterations_per_sec", "value": None if parity_failed else value, "unit": "think-iterations/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * tmax / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",            # the path's arithmetic and state (--model-dtype bf16: backbone I/O only)
        "config": {"workload": f"{args.workload}: latent {'x'.join(map(str, shape))} per GPU, {n_sig} sigmas x {n_think} think "
                               f"iterations, {mask_description(shape, args.mask)}, stub backbone x->(0.9x,0.8x), "
                               f"{'flow' if flow else 'VE/Karras'} schedule",
                   "engine": ("drop-in: LanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize, IS_FLUX, IS_FLOW), no optional keyword"
                              if not kw and args.mask_format == "f32" else "keywords: " + repr(kw)),
                   "rng": rng_name, "graph": "auto" if args.graph is None else bool(args.graph), "mask_format": args.mask_format,
                   "mask_seen_by_kernels": "bits" if mask_packed else args.mask_format,
                   "captured_calls": captured_calls, "backbone_io": args.model_dtype, "replicas": args.gpus,
                   "rows_per_gpu": shape[0], "global_rows": shape[0] * args.gpus,
                   "iterations_per_step": n_sig * n_think, "latent_elements_per_gpu": n_el},
        "parity_check": (None if parity is None else {k: parity.get(k) for k in ("mse_x", "mse_denoised_max", "tolerance", "ok", "sigmas_checked", "launch_modes", "error") if k in parity}),
        "roofline": compact_roofline(steady), "cpu_baseline": compact_cpu(cpu), "summary": summary,
        "collective": (None if dist_info is None else ("rccl" if dist_info["backend"] == "nccl" else dist_info["backend"])),
        "distinct_devices": (1 if dist_info is None else dist_info["distinct_devices"]),
        "dist": summarise_dist(dist_info),
    }
    if parity_failed:
        line["error"] = ("parity_check failed: the timed configuration does not reproduce the oracle within the stated tolerance; "
                         f"no value is reported (measured {value:.1f} it/s is void)")
    return line


def error_line(args, message):
    return {"metric": "langevin_think_iterations_per_sec", "value": None, "unit": "think-iterations/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": args.workload}, "roofline": None,
            "cpu_baseline": None, "error": message, "nccl_debug": os.environ.get("NCCL_DEBUG")}
