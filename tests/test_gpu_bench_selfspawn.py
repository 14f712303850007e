"""`python bench.py --gpus 2` with NO launcher in front of it: bench.py starts its
own ranks, the ranks find each other on 127.0.0.1, rank 0 prints ONE JSON line whose `dist` block shows that the
backend carried two ranks and that each of them ran the full K steps.  On a one-GPU box both ranks share the device and
RCCL refuses that ("Duplicate GPU detected"), so the run asks for gloo; with two devices the same command line with
--dist-backend nccl goes over RCCL (second test, skipped where there is one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(backend, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", backend, "--steps", "3",
           "--warmup", "1", "--repeats", "1", "--prewarm-seconds", "0.05", "--cpu-seconds", "1.0", "--no-large-shape", *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _check(line, backend):
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    d = line["dist"]
    assert d["backend"] == backend and d["world_size"] == 2 and d["ranks_reporting"] == 2
    assert d["launcher"] == "bench.py self-spawn" and d["collectives_in_timed_region"] == 0
    assert len(d["per_rank_it_s"]) == 2 and all(v > 0 for v in d["per_rank_it_s"])
    per = d["per_rank"]
    assert [r["rank"] for r in per] == [0, 1] and all(r["steps"] == 3 for r in per)
    assert all(r["iterations"] == 3 * line["config"]["iterations_per_step"] for r in per)      # every rank ran the full K steps
    assert per[0]["pid"] != per[1]["pid"]
    assert per[0]["final_checksum"] != per[1]["final_checksum"]        # replicas differ through seed + rank only
    assert d["broadcast_bytes"] >= 2 * 4 * line["config"]["latent_elements_per_gpu"] and d["broadcast_ms"] > 0
    # whole-job value = all ranks' iterations / slowest rank's clock
    assert abs(line["value"] - sum(r["iterations"] for r in per) / max(r["elapsed_s"] for r in per)) / line["value"] < 0.05
    assert line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0     # rank 0 still reports both at N > 1
    return d


@pytest.mark.timeout(900)
def test_bench_two_ranks_without_a_launcher_gloo():
    d = _check(_run("gloo"), "gloo")
    assert d["rccl_version"] is None


@pytest.mark.timeout(900)
def test_bench_two_ranks_without_a_launcher_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        # asking for nccl with one device must still produce the line (automatic gloo fallback, said in the line)
        line = _run("nccl")
        d = _check(line, "gloo")
        assert d["backend_requested"] == "nccl" and d["distinct_devices"] == 1
        return
    d = _check(_run("nccl"), "nccl")
    assert d["rccl_version"] and d["distinct_devices"] == 2


@pytest.mark.timeout(900)
def test_bench_two_ranks_under_torch_distributed_run():
    """The launcher form the driver documents (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`)
    keeps working next to the self-started one: same line, `dist.launcher` says who started the ranks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3",
           "--warmup", "1", "--repeats", "0", "--prewarm-seconds", "0.05", "--no-cpu-baseline", "--no-large-shape"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    d = line["dist"]
    assert line["n_gpus"] == 2 and d["world_size"] == d["ranks_reporting"] == 2 and d["launcher"] == "external"
    assert all(r["iterations"] == 3 * line["config"]["iterations_per_step"] for r in d["per_rank"])
