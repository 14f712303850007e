"""N > 1: starting the ranks when no launcher did, the per-rank report, and its summary for the headline line."""
from __future__ import annotations

import os
import subprocess
import sys
import time

import numpy as np


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, port, base=None):
    """The environment of each of the `n` ranks `python bench.py --gpus n` starts when no launcher did: what
    torch.distributed.run would export (one process per GPU, rendezvous on 127.0.0.1).  One OpenMP / MKL thread per rank: N
    ranks on a container that grants fewer CPUs than N x (torch's default team) would otherwise oversubscribe the host
    before the timed region starts."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "LANPAINT_BENCH_LAUNCHER": "bench.py self-spawn"})
        env.setdefault("OMP_NUM_THREADS", "1")
        envs.append(env)
    return envs


def spawn_ranks(n, argv, script):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves -- one
    child process per GPU running `script`, rank 0's stdout (the JSON line) passed through -- and wait for all
    of them.  A rank that fails takes the others down (by PID) and the exit code is its code."""
    envs = rank_environments(n, free_port())
    procs = [subprocess.Popen([sys.executable, script] + list(argv), env=env,
                              stdout=None if r == 0 else subprocess.DEVNULL) for r, env in enumerate(envs)]
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _stats(vals):
    v = [float(x) for x in vals if x is not None]
    if not v:
        return None
    return {"min": min(v), "median": float(np.median(v)), "max": max(v)}


def summarise_dist(dist_info):
    """The `dist` block of the headline line: who took part and the spread over ranks, in a few hundred bytes whatever the
    world size (the per-rank reports go to the side-car file)."""
    if dist_info is None:
        return None
    per = dist_info.get("per_rank") or []
    keep = ("backend", "backend_requested", "world_size", "ranks_reporting", "distinct_devices", "rccl_version", "launcher",
            "broadcast_bytes", "broadcast_ms", "collectives_in_timed_region", "shared_checksums_equal", "global_rows",
            "parity_ok_all_ranks", "slowest_rank", "init_process_group_s")
    out = {k: dist_info.get(k) for k in keep if k in dist_info}
    out["it_s"] = _stats(r.get("it_s") for r in per)
    out["own_it_s"] = _stats(r.get("own_it_s") for r in per)
    out["steady_launch_us"] = _stats(r.get("steady_launch_us") for r in per)
    out["first_barrier_wait_s"] = _stats(r.get("t_first_barrier_wait_s") for r in per)
    out["process_time_over_elapsed"] = _stats(r.get("process_time_over_elapsed") for r in per)
    out["distinct_final_checksums"] = len({r.get("final_checksum") for r in per})
    out["iterations_per_rank"] = sorted({r.get("iterations") for r in per})
    return out
