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


def _cmd(backend, gpus=2, extra=(), sidecar=None):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dist-backend", backend, "--steps", "3",
            "--warmup", "1", "--repeats", "1", "--prewarm-seconds", "0.05", "--cpu-seconds", "1.0", "--no-large-shape",
            "--parity-sigmas", "4", *(("--sidecar", sidecar) if sidecar else ()), *extra]
            # (every rank checks its own replica: keep the numpy passes short here)


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _run(backend, extra=(), gpus=2, timeout=600):
    """(the headline line, the side-car's full `dist` block with the per-rank reports)"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, "extras.json")
        p = subprocess.run(_cmd(backend, gpus, extra, side), env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and p.stdout.rstrip().endswith(lines[0]), p.stdout[-2000:]      # ONE line, the last thing on stdout
        assert len(lines[0]) < 6000, len(lines[0])
        line = json.loads(lines[0])
        line["_dist_full"] = json.load(open(side))["dist"]
    return line


def _check(line, backend):
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["collective"] == ("rccl" if backend == "nccl" else backend)         # top level: impossible to miss
    assert line["parity_check"]["ok"]
    ds, d = line["dist"], line["_dist_full"]
    assert "per_rank" not in ds and ds["world_size"] == 2 and ds["it_s"]["min"] > 0 and ds["iterations_per_rank"] == [3 * line["config"]["iterations_per_step"]]
    assert ds["distinct_final_checksums"] == 2 and ds["parity_ok_all_ranks"] and ds["slowest_rank"] in (0, 1)
    assert line["distinct_devices"] == d["distinct_devices"] and d["shared_checksums_equal"]
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
    assert line["cpu_baseline"]["value"] > 0 and 0 < line["roofline"]["frac"] <= 1     # rank 0 still reports both at N > 1
    # round 5: every rank checked its own replica against the oracle and reports what explains a slow rank
    assert d["parity_ok_all_ranks"] and all(r["parity_ok"] is True and r["parity_mse_x"] < 1e-9 for r in per)
    for r in per:
        assert r["own_elapsed_s"] <= r["elapsed_s"] and r["own_it_s"] > 0 and 0 < r["process_time_over_elapsed"] < 4
        assert r["t_first_barrier_wait_s"] >= 0 and r["steady_launch_us"] > 0 and r["cpus_allowed"] >= 1
        assert r["setup_s"] > 0 and r["init_process_group_s"] >= 0
    assert d["slowest_rank"] in (0, 1) and d["own_it_s_spread"][0] <= d["own_it_s_spread"][1]
    return d


@pytest.mark.timeout(900)
def test_bench_two_ranks_without_a_launcher_gloo():
    d = _check(_run("gloo"), "gloo")
    assert d["rccl_version"] is None


@pytest.mark.timeout(900)
def test_bench_two_ranks_without_a_launcher_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        # RCCL cannot carry two ranks on one device.  Asking for it there is an ERROR (no line, exit code != 0) ...
        p = subprocess.run(_cmd("nccl"), env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert "--allow-gloo-fallback" in p.stderr
        # ... unless the caller explicitly accepts a gloo rehearsal, which the line then states at top level
        line = _run("nccl", extra=("--allow-gloo-fallback",))
        d = _check(line, "gloo")
        assert d["backend_requested"] == "nccl" and d["distinct_devices"] == 1 and line["collective"] == "gloo"
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
           "--warmup", "1", "--repeats", "0", "--prewarm-seconds", "0.05", "--no-cpu-baseline", "--no-large-shape",
           "--parity-sigmas", "3"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    d = line["dist"]
    assert line["n_gpus"] == 2 and d["world_size"] == d["ranks_reporting"] == 2 and d["launcher"] == "external"
    assert d["iterations_per_rank"] == [3 * line["config"]["iterations_per_step"]] and d["distinct_final_checksums"] == 2


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("workload,rows", [("c3_sdxl_b4", 4), ("c5_wan", 1)])
def test_bench_eight_ranks_rehearse_the_multi_gpu_configurations_baseline_names(workload, rows):
    """BASELINE.json configs[2] (SDXL batch 32 = 4 rows per GPU, mask / known latent / SDXL-shaped cond broadcast) and
    configs[4] (Wan video latent, batch-sharded over 8 GPUs) in their EIGHT-rank form: eight processes, one packed broadcast
    from rank 0, no collective in the timed region, every rank the full K steps on its own replica.  On a box with fewer
    than eight devices the ranks share them and travel over gloo -- a rehearsal of the code path (rendezvous, broadcast,
    per-rank seeds, reduction, the line), not a scaling measurement; with eight devices the same command runs over RCCL."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    line = _run(backend, gpus=8, timeout=1400,
                extra=("--workload", workload, "--steps", "2", "--repeats", "0", "--no-cpu-baseline", "--parity-sigmas", "2"))
    d = line["_dist_full"]
    assert "per_rank" not in line["dist"] and line["dist"]["world_size"] == 8 and line["dist"]["distinct_final_checksums"] == 8
    assert line["n_gpus"] == 8 and d["world_size"] == d["ranks_reporting"] == 8 and line["collective"] == ("rccl" if backend == "nccl" else "gloo")
    assert line["config"]["rows_per_gpu"] == rows and line["config"]["global_rows"] == 8 * rows == d["global_rows"]
    assert d["collectives_in_timed_region"] == 0 and d["shared_checksums_equal"]
    per = d["per_rank"]
    assert [r["rank"] for r in per] == list(range(8)) and len({r["pid"] for r in per}) == 8
    assert all(r["iterations"] == 2 * line["config"]["iterations_per_step"] for r in per)
    assert len({r["final_checksum"] for r in per}) == 8                               # eight different replicas (seed + rank)
    shared = d["shared_tensors"]
    assert set(shared) >= {"mask", "y", "cond"} and (workload != "c3_sdxl_b4" or (shared["cond"] == [1, 77, 2048] and shared["pooled"] == [1, 2816]))
    n_el = line["config"]["latent_elements_per_gpu"]
    assert d["broadcast_bytes"] >= 2 * 4 * n_el + 2 * 77 * 2048
    assert line["parity_check"]["ok"] and line["value"] > 0 and d["parity_ok_all_ranks"]
    assert all(r["parity_ok"] is True and r["steady_launch_us"] > 0 and r["t_first_barrier_wait_s"] >= 0 for r in per)
