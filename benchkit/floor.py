"""launch_floor: what the sigma schedule costs on this box with NO host code between the launches.

The headline shape is launch-bound: one sigma call is one hipGraphLaunch of 13 dependent kernel nodes plus the sampler's own
update launch, and the runtime's cost per graph launch and per dependent node is the limit, not the kernels and not (up to a
point) the host.  This module measures that limit directly instead of asserting it: the hipGraphExec_t handles of the job's
captured sigma calls are launched back to back from ONE host call (lp_replay_burst) with an elementwise launch standing for
the sampler's update between them -- same graphs, same kernels, same dependencies, zero Python -- and the measured schedule is
quoted as a fraction of it.  A ladder of variants shows what each layer adds:

  graphs_only            hipGraphLaunch x n_sigmas from C
  graphs_and_update      ... with the update launch between them (the floor of the measured loop)
  replay_call_loop       the engine's own C entry per call (lp_replay_call: node-0 argument refresh + launch) + update, driven
                         by a bare Python loop
  engine_loop            the measured thing: LanPaint.__call__ + torch.lerp per sigma
"""
from __future__ import annotations

import ctypes
import time

import torch

from .roofline import standalone_step
from .workloads import Job


def _wall_us(fn, dev, reps):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return 1e6 * (time.perf_counter() - t0) / reps


def engine_floor(_cabi, dev, workload="c2_sdxl", seed=0, steps=40, mask_kind=None, **engine_kw):
    """The ladder for the engine-direct schedule of `workload` (the headline's loop).  `engine_kw` empty = the drop-in engine."""
    lib = _cabi.load()
    job = Job(workload, dev, seed=seed, mask_kind=mask_kind, mask_format="f32" if not engine_kw else "bits")
    eng = job.engine(**engine_kw)
    for _ in range(6):
        job.run(eng)
    torch.cuda.synchronize(dev)
    cap = eng._last_cap
    if cap is None or cap.raw_exec is None or not cap.final_in_graph:
        return {"error": "the job's sigma call is not one raw graph launch on this engine (nothing to replay from C)"}
    n = job.n_sig
    execs = (ctypes.c_void_p * n)(*([cap.raw_exec] * n))
    before = None if cap.binding is not None else ctypes.pointer(cap.k0_desc)      # (the replace launch is node 0 of the graph)
    upd, keep, _n = standalone_step(_cabi, workload, dev, phase=_cabi.LP_PH_EMIT, mask_kind=mask_kind)   # 8 B / element, like the lerp
    st = torch.cuda.current_stream(dev).cuda_stream
    burst = lambda after: _cabi.check(lib.lp_replay_burst(execs, n, before, after, 1, st), "lp_replay_burst")   # noqa: E731
    for _ in range(5):
        burst(ctypes.pointer(upd))
    t_graphs = _wall_us(lambda: burst(None), dev, steps)
    t_floor = _wall_us(lambda: burst(ctypes.pointer(upd)), dev, steps)

    def replay_loop():
        for _ in range(n):
            _cabi.check(lib.lp_replay_call(ctypes.byref(cap.call), st), "lp_replay_call")
            _cabi.check(lib.lp_step(ctypes.byref(upd), st), "lp_step")
    t_replay = _wall_us(replay_loop, dev, steps)
    t_engine = _wall_us(lambda: job.run(eng), dev, steps)
    its = job.n_sig * job.n_think
    nodes = 2 * job.n_think + 3                     # replace, n x (backbone, step), final backbone call, finalize
    del keep
    return {"workload": workload, "engine": "drop-in" if not engine_kw else repr(engine_kw), "graph_nodes_per_sigma_call": nodes,
            "replace_launch": "node 0 of the graph" if cap.binding is not None else "eager, in front of the graph",
            "us_per_sigma_call": {"graphs_only": t_graphs / n, "graphs_and_update": t_floor / n, "replay_call_loop": t_replay / n,
                                  "engine_loop": t_engine / n},
            "floor_it_s": its / (t_floor * 1e-6), "engine_loop_it_s": its / (t_engine * 1e-6),
            "engine_over_floor": t_floor / t_engine, "steps_timed": steps,
            "note": "floor = the same captured graphs + an elementwise launch standing for the sampler's update, enqueued "
                    "from one C call (lp_replay_burst): no Python, no per-call argument refresh; engine_over_floor = 1 would "
                    "mean the host costs nothing"}


def node_floor(_cabi, dev, args, steps=40, **engine_kw):
    """The same for the node-default schedule through KSamplerX0Inpaint: per sigma the replace launch (eager, carrying the
    sigma -> times kernel work), the tail graph captured for that sigma's inner-step count, the update launch."""
    from .extras import build_node_sampler
    lib = _cabi.load()
    k, node_pass, n_sig = build_node_sampler(args, dev, **engine_kw)
    for _ in range(8):
        node_pass()
    torch.cuda.synchronize(dev)
    counts = []
    node_pass(record=counts)
    pm = k.PaintMethod
    cap0 = pm._last_cap
    table = getattr(cap0, "node_table", None) if cap0 is not None else None
    if table is None or len(counts) != n_sig or any(c is None for c in counts):
        return {"error": "the node path did not reach its one-call steady state (lp_node_call) on this run", "counts": counts}
    caps = table[1]
    handles = [caps[c].tail.graph_exec if (c < len(caps) and caps[c] is not None) else None for c in counts]
    if any(h is None for h in handles):
        return {"error": "no captured tail graph for some inner-step count", "counts": counts}
    execs = (ctypes.c_void_p * n_sig)(*handles)
    upd, keep, _n = standalone_step(_cabi, "c2_sdxl", dev, phase=_cabi.LP_PH_EMIT)
    st = torch.cuda.current_stream(dev).cuda_stream
    before = ctypes.pointer(cap0.k0_desc)
    burst = lambda: _cabi.check(lib.lp_replay_burst(execs, n_sig, before, ctypes.pointer(upd), 1, st), "lp_replay_burst")   # noqa: E731
    for _ in range(5):
        burst()
    t_floor = _wall_us(burst, dev, steps)
    # the whole call as ONE graph (node 0 = the replace launch with the sigma algebra folded in): what a speculated call launches
    t_one = None
    full = table[5] if len(table) > 5 else None
    if full is not None and all(full[c] for c in counts):
        execs1 = (ctypes.c_void_p * n_sig)(*[full[c] for c in counts])
        burst1 = lambda: _cabi.check(lib.lp_replay_burst(execs1, n_sig, None, ctypes.pointer(upd), 1, st), "lp_replay_burst")   # noqa: E731
        for _ in range(5):
            burst1()
        t_one = _wall_us(burst1, dev, steps)
    t_node = _wall_us(node_pass, dev, steps)
    its = sum(counts)
    del keep
    return {"inner_steps_per_sigma": counts, "iterations_per_step": its,
            "us_per_sigma_call": {"replace_graph_update": t_floor / n_sig, "one_graph_and_update": (t_one / n_sig) if t_one else None,
                                  "node_loop": t_node / n_sig},
            "floor_it_s": its / ((t_one or t_floor) * 1e-6), "node_loop_it_s": its / (t_node * 1e-6),
            "node_over_floor": (t_one or t_floor) / t_node,
            "steps_timed": steps,
            "note": "floor = per sigma [the whole-call graph captured for that sigma's count (node 0 = replace launch + sigma algebra), "
                    "update launch] from one C call (replace_graph_update: the round-5 form, an eager replace launch in front of the "
                    "tail graph); the node loop adds KSamplerX0Inpaint + lp_node_call (speculated count, verdict from the device) + torch.lerp"}
