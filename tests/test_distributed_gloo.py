"""The N > 1 path on CPU: world_size-2 gloo process group (one process per "GPU"), covering the
setup broadcast, row sharding, per-rank seeds, throughput reduction and the early-stop partial-sum
all-reduce.  No collective runs inside the think loop, so this is the whole distributed surface."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from lanpaint_amd import distributed as D
    r, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    shared = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        shared = {"mask": (torch.rand(1, 4, 8, 8, generator=g) > 0.5).float(),
                  "y": torch.randn(1, 4, 8, 8, generator=g),
                  "cond": torch.randn(1, 77, 32, generator=g).to(torch.bfloat16),
                  "pooled": torch.randn(1, 7, generator=g).to(torch.float16),
                  "ids": torch.arange(5, dtype=torch.int64)}
    stats = {}
    got = D.broadcast_job(shared, src=0, stats=stats)
    digest = {k: (tuple(v.shape), str(v.dtype), float(v.double().sum())) for k, v in got.items()}
    t, n = D.reduce_throughput(1.0 + rank, 150)
    acc = torch.tensor([[1.0 + rank, 2.0, 3.0, 4.0], [0.5, 0.5, 0.5, 0.5]], dtype=torch.float64)
    D.all_reduce_stop_sums(acc)
    rep = D.gather_rank_reports({"rank": rank, "device": f"cpu:{rank}", "pci_bus_id": None, "it_s": 100.0 * (rank + 1)})
    q.put((rank, digest, t, n, acc.tolist(), D.replica_seed(7, rank), D.shard_rows(32, world, rank), stats, rep))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_setup_and_reductions():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, d0, t0, n0, a0, s0, sh0, st0, rep0), (r1, d1, t1, n1, a1, s1, sh1, st1, rep1) = res
    # the evidence block bench.py prints at N > 1: who took part, over which backend, how much the set-up broadcast moved
    assert st0["bytes"] == st1["bytes"] > 0 and st0["backend"] == "gloo" and st0["ms"] >= 0.0
    assert rep0 == rep1 and rep0["backend"] == "gloo" and rep0["world_size"] == rep0["ranks_reporting"] == 2
    assert rep0["device_per_rank"] == ["cpu:0", "cpu:1"] and rep0["per_rank_it_s"] == [100.0, 200.0]
    assert rep0["distinct_devices"] == 2 and rep0["rccl_version"] is None
    assert d0 == d1 and set(d0) == {"mask", "y", "cond", "pooled", "ids"}      # every rank holds the same job
    assert d0["cond"][1] == "torch.bfloat16" and d0["ids"][1] == "torch.int64"
    assert t0 == t1 == 2.0 and n0 == n1 == 300                                 # max time, summed units
    assert a0 == a1 == [[3.0, 4.0, 6.0, 8.0], [1.0, 1.0, 1.0, 1.0]]
    assert (s0, s1) == (7, 8)
    assert (sh0, sh1) == ((0, 16), (16, 32))


def test_shard_rows_and_single_process_identities():
    from lanpaint_amd import distributed as D
    for batch, world in [(32, 8), (5, 4), (3, 8), (1, 1)]:
        spans = [D.shard_rows(batch, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == batch
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    t = {"m": torch.ones(2, 2)}
    assert D.broadcast_job(t)["m"] is t["m"]
    assert D.reduce_throughput(1.5, 10) == (1.5, 10)
    acc = torch.ones(2, 4, dtype=torch.float64)
    assert D.all_reduce_stop_sums(acc) is acc
    assert D.env_world()[1] >= 1
    assert D.gather_rank_reports({"rank": 0}) is None            # no process group: no evidence block


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` without torch.distributed.run must not die on plumbing: the environment each
    self-started rank gets is what the launcher would export, rendezvous on 127.0.0.1."""
    import bench
    envs = bench.rank_environments(4, 29511, base={"KEEP": "1"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
               and e["KEEP"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert 1024 < bench._free_port() < 65536
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "spawn_ranks(args.gpus" in src and "launch with torch.distributed.run" not in src


@pytest.mark.timeout(300)
def test_self_started_ranks_fail_fast_and_loudly_without_a_gpu():
    """The launcher-less N > 1 entry point on a box with no GPU: every rank dies on its first device call; the parent must
    come back promptly with a non-zero exit code (no hang waiting for a rendezvous that will never complete, no stray
    children) and print no JSON line."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "stopping the other ranks" in p.stderr or "Error" in p.stderr or "error" in p.stderr


@pytest.mark.timeout(180)
def test_single_rank_group_goes_through_the_collectives():
    """A ONE-rank process group is not a shortcut (round 5): broadcast_job, reduce_throughput, all_reduce_stop_sums and
    gather_rank_reports issue their collectives whenever a group exists, so a one-GPU box can exercise the code path the
    8-GPU run takes.  Here over gloo, in a child process (the GPU twin runs RCCL: tests/test_gpu_rccl_single_rank.py)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c",
                        "import sys, json; sys.path.insert(0, %r); from lanpaint_amd import distributed as d; "
                        "print(json.dumps(d.single_rank_selftest(backend='gloo')))" % ROOT],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=150)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["backend"] == "gloo" and rec["tensors_byte_identical"] and rec["reports_gathered"] == 1, rec
    assert rec["broadcast_bytes"] > 2 * 256 * 1024 + 77 * 2048 * 2 and rec["rccl_version"] is None
    assert len(rec["collectives"]) == 6


def _worker8(rank, world, port, q):
    """One of eight ranks sharing ONE batch of 30 rows (uneven: 4,4,4,4,4,4,3,3): rank 0 owns the job, every rank gets the shared
    tensors from the one packed broadcast, takes its row slice, and reports what the bench's evidence block is built from."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    from lanpaint_amd import distributed as D
    r, w = D.init(backend="gloo", timeout_s=120)
    assert (r, w) == (rank, world)
    batch = 30
    shared = None
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        shared = {"mask": (torch.rand(1, 4, 16, 16, generator=g) > 0.5).float(),
                  "y": torch.randn(batch, 4, 16, 16, generator=g),                        # the known latent of every row of the batch
                  "cond": torch.randn(1, 77, 64, generator=g).to(torch.bfloat16),
                  "odd": torch.arange(13, dtype=torch.uint8)}                             # keeps the 16-byte padding of the pack honest
    stats = {}
    got = D.broadcast_job(shared, src=0, stats=stats)
    lo, hi = D.shard_rows(batch, world, rank)
    mine = got["y"][lo:hi]
    t, n = D.reduce_throughput(0.5 + 0.01 * rank, 150 * (hi - lo))
    acc = torch.tensor([float(hi - lo), 1.0, float(rank), 2.0], dtype=torch.float64)
    D.all_reduce_stop_sums(acc)
    rep = D.gather_rank_reports({"rank": rank, "device": f"cpu:{rank}", "pci_bus_id": None, "rows": hi - lo, "it_s": 1000.0 + rank,
                                 "own_it_s": 1001.0 + rank, "iterations": 150, "final_checksum": float(mine.double().sum()),
                                 "steady_launch_us": 3.0 + 0.1 * rank, "t_first_barrier_wait_s": 0.001 * rank,
                                 "process_time_over_elapsed": 0.9, "parity_ok": True})
    q.put((rank, (lo, hi), float(got["y"].double().sum()), float(mine.double().sum()), t, n, acc.tolist(), stats, rep,
           D.replica_seed(11, rank), str(got["cond"].dtype), got["odd"].tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_gloo_uneven_batch_and_the_compact_dist_block():
    """the eight-rank form of the set-up path before any eight-GPU box runs it -- 30 rows over 8 ranks
    (4,4,4,4,4,4,3,3), one packed broadcast, reductions, the report gather -- and the `dist` block of the headline line built
    from the gathered reports stays a few hundred bytes (per-rank detail goes to the side-car)."""
    import json
    from benchkit.ranks import summarise_dist
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spans = [r[1] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 30 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert [b - a for a, b in spans] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert len({r[2] for r in res}) == 1                                        # every rank received the same y
    assert abs(sum(r[3] for r in res) - res[0][2]) < 1e-6 * max(1.0, abs(res[0][2]))   # the slices tile the batch
    assert all(r[4] == pytest.approx(0.57) and r[5] == 150 * 30 for r in res)  # slowest rank's clock, all rows' iterations
    assert all(r[6] == [30.0, 8.0, 28.0, 16.0] for r in res)
    assert len({r[7]["bytes"] for r in res}) == 1 and res[0][7]["bytes"] >= 30 * 4 * 16 * 16 * 4
    assert [r[9] for r in res] == list(range(11, 19)) and all(r[10] == "torch.bfloat16" and r[11] == list(range(13)) for r in res)
    rep = res[0][8]
    assert all(r[8] == rep for r in res) and rep["world_size"] == rep["ranks_reporting"] == 8 and rep["distinct_devices"] == 8
    rep.update({"backend_requested": "nccl", "launcher": "external", "collectives_in_timed_region": 0, "shared_checksums_equal": True,
                "global_rows": sum(r["rows"] for r in rep["per_rank"]), "parity_ok_all_ranks": True, "slowest_rank": 7,
                "broadcast_bytes": res[0][7]["bytes"], "broadcast_ms": res[0][7]["ms"]})
    d = summarise_dist(rep)
    assert "per_rank" not in d and d["global_rows"] == 30 and d["world_size"] == 8 and d["slowest_rank"] == 7
    assert d["it_s"] == {"min": 1000.0, "median": 1003.5, "max": 1007.0} and d["distinct_final_checksums"] == 8
    assert d["steady_launch_us"]["max"] == pytest.approx(3.7) and d["iterations_per_rank"] == [150]
    assert len(json.dumps(d)) < 900


def _worker_host_group(q):
    """A host application's OWN one-rank gloo group must not turn the engine's reductions into collectives."""
    sys.path.insert(0, ROOT)
    import tempfile
    import torch.distributed as dist
    from lanpaint_amd import distributed as D
    store = tempfile.mktemp(prefix="lp_host_pg_")
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=0, world_size=1)
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real(*a, **k))[1]
    t = {"m": torch.ones(2, 2)}
    same = D.broadcast_job(t)["m"] is t["m"]
    acc = torch.ones(4, dtype=torch.float64)
    D.all_reduce_stop_sums(acc)
    tp = D.reduce_throughput(1.5, 10)
    none_rep = D.gather_rank_reports({"rank": 0}) is None
    shortcut = (same, tp, none_rep, list(calls))
    os.environ["LANPAINT_AMD_FORCE_COLLECTIVES"] = "1"          # ... unless the real path is asked for
    D.all_reduce_stop_sums(acc)
    forced = list(calls)
    dist.destroy_process_group()
    q.put((shortcut, forced))


@pytest.mark.timeout(180)
def test_a_host_applications_one_rank_group_keeps_the_shortcuts():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_host_group, args=(q,))
    p.start()
    shortcut, forced = q.get(timeout=120)
    p.join(30)
    assert p.exitcode == 0
    assert shortcut == (True, (1.5, 10), True, []) and forced == ["all_reduce"]


@pytest.mark.timeout(120)
def test_a_missing_peer_raises_instead_of_hanging():
    """init(timeout_s=...) bounds the rendezvous: rank 0 of a two-rank world whose peer never shows up comes back with an
    exception within the limit (bench.py turns it into a line with `error`, exit code 5)."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); from lanpaint_amd import distributed as d\n"
                        "try:\n    d.init('gloo', timeout_s=5)\n    print('JOINED')\nexcept Exception as e:\n    print('RAISED', type(e).__name__)" % ROOT],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=100)
    assert "RAISED" in p.stdout and "JOINED" not in p.stdout, (p.stdout, p.stderr[-500:])
    assert time.time() - t0 < 90
