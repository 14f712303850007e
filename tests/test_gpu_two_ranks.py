"""The N > 1 path on real hardware: two ranks, one process each, sharing whatever GPUs the box
has (RCCL over xGMI when there are two devices; on a one-GPU box RCCL refuses two ranks on one device --
"Duplicate GPU detected" -- so the process group falls back to gloo, which moves device tensors through the
host).  Each rank: packed broadcast of mask / known latent from rank 0, per-rank seed, sigma calls replayed as
hipGraphs beside the other rank, whole-job throughput reduction.  Checked: every rank holds the broadcast job,
rank r's trajectory equals a single-process run with seed + r bit for bit, ranks differ only through the seed.
Second scenario (the one exchange the path can need, SURVEY.md 8e exception 1): ONE batch sharded over the ranks
with the inner early stop on -- `early_stop_group` all-reduces the partial sums, so both ranks stop where a single
process holding the whole batch stops (a half of the batch alone would go on longer) and produce its rows."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE, N_SIG, N_THINK, SEED = (2, 4, 32, 32), 4, 3, 1234


def _job(seed_rank, mask, y, dev, graph):
    """`N_SIG` sigma calls with an Euler update between them; returns (final x, list of denoised)."""
    import torch
    from lanpaint_amd import LanPaint
    from tests import golden_cases as gc
    from tests.stubs import MODELS
    g = torch.Generator(device="cpu").manual_seed(seed_rank)
    noise = torch.randn(SHAPE, generator=g).to(dev)
    sig = gc.karras_sigmas(N_SIG)[:-1]
    x = y + noise * float(sig[0])
    eng = LanPaint(MODELS["linear_tuple"](), N_THINK, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=seed_rank, graph=graph)
    outs = []
    for i in range(N_SIG):
        s = torch.full((SHAPE[0],), float(sig[i]), dtype=torch.float32, device=dev)
        den = eng(x, y, noise, s, mask, gc.times_from_sigma(s, False), None, seed_rank)
        outs.append(den)
        if i + 1 < N_SIG:
            x = torch.lerp(den, x, float(sig[i + 1] / sig[i]))
    torch.cuda.synchronize()
    return x.cpu(), [o.cpu() for o in outs], eng


def _shared_job():
    import torch
    from tests import golden_cases as gc
    g = torch.Generator(device="cpu").manual_seed(99)
    return {"mask": torch.from_numpy(gc.box_mask(SHAPE)), "y": torch.randn(SHAPE, generator=g)}


ES_SHAPE, ES_N = (4, 4, 12, 12), 8      # two rows per rank (per-row sigma on both sides of the comparison)


def _es_inputs():
    """One batch of four rows for the sharded early-stop scenario (numpy seed: identical on every rank)."""
    rng = np.random.default_rng(77)
    y = rng.standard_normal(ES_SHAPE, dtype=np.float32)
    noise = rng.standard_normal(ES_SHAPE, dtype=np.float32)
    mask = np.ones(ES_SHAPE, dtype=np.float32)
    mask[..., 3:9, 3:9] = 0.0
    draws = [rng.standard_normal(ES_SHAPE, dtype=np.float32) for _ in range(2 * ES_N)]
    return y, noise, mask, draws


def _es_run(rows, dev, group):
    """The inner early stop on rows [rows] of the batch; `group`: the partial sums are all-reduced over the ranks."""
    import torch
    from lanpaint_amd import LanPaint
    from tests import golden_cases as gc
    from tests.stubs import MODELS
    y, noise, mask, draws = _es_inputs()
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a[rows])).to(dev)   # noqa: E731
    it = iter([tt(d) for d in draws])
    trace = []
    eng = LanPaint(MODELS["linear_tuple"](), ES_N, 15.0, 5.0, 1.0, 0.2, rng=lambda like: next(it), early_stop_group=group)
    s = torch.full((y[rows].shape[0],), 1.0, dtype=torch.float32, device=dev)
    x = tt(y + noise * 1.0)
    mo = {"lanpaint_semantic_stop": {"threshold": 0.3, "patience": 1}, "lanpaint_semantic_trace": trace}
    out = eng(x, tt(y), tt(noise), s, tt(mask), gc.times_from_sigma(s, False), mo, 0)
    torch.cuda.synchronize()
    return x.cpu(), out.cpu(), eng.iterations_run, [(t["dist"], t["patience_counter"], t["stopped"]) for t in trace]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from lanpaint_amd import distributed as D
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", rank % n_dev)
    torch.cuda.set_device(dev)
    backend = "nccl" if n_dev >= world else "gloo"
    r, w = D.init(backend=backend, device=dev)
    assert (r, w) == (rank, world)
    job = D.broadcast_job({k: v.to(dev) for k, v in _shared_job().items()} if rank == 0 else None, src=0, device=dev)
    x, outs, eng = _job(D.replica_seed(SEED, rank), job["mask"], job["y"], dev, graph=True)
    t, n = D.reduce_throughput(1.0 + rank, eng.iterations_run, dev)
    # ONE batch of two rows sharded over the two ranks: the stop metric is defined over the whole batch (earlystop.py:52-55)
    es = _es_run(slice(2 * rank, 2 * rank + 2), dev, True)
    torch.save({"x": x, "outs": outs, "mask": job["mask"].cpu(), "y": job["y"].cpu(), "t": t, "n": n, "backend": backend,
                "graphs": len(eng._graphs), "iters": eng.iterations_run, "es": es}, os.path.join(outdir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_replay_graphs_side_by_side_and_match_single_process_runs():
    import torch
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, world, port, outdir)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(500)
            assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    shared = _shared_job()
    dev = torch.device("cuda", 0)
    for r in range(world):
        assert torch.equal(res[r]["mask"], shared["mask"]) and torch.equal(res[r]["y"], shared["y"])   # the broadcast job
        assert res[r]["graphs"] == 1 and res[r]["iters"] == N_SIG * N_THINK
        assert res[r]["t"] == 2.0 and res[r]["n"] == world * N_SIG * N_THINK                            # max time, summed units
        x1, outs1, _ = _job(SEED + r, shared["mask"].to(dev), shared["y"].to(dev), dev, graph=True)      # single process, same seed
        assert torch.equal(res[r]["x"], x1)
        for a, b in zip(res[r]["outs"], outs1):
            assert torch.equal(a, b)
    assert not torch.equal(res[0]["x"], res[1]["x"])                    # ranks differ -- only through the seed
    # graph replay vs eager launches use different Philox sequence numbers, so they agree statistically only
    xe, _, _ = _job(SEED, shared["mask"].to(dev), shared["y"].to(dev), dev, graph=False)
    assert abs(float(xe.std()) / float(res[0]["x"].std()) - 1.0) < 0.1
    # known region of the denoised output is the known latent on every rank (lanpaint.py:154)
    m = shared["mask"].numpy() > 0.5
    for r in range(world):
        np.testing.assert_array_equal(res[r]["outs"][-1].numpy()[m], shared["y"].numpy()[m])
    # sharded early stop: both ranks take the single-process decision and produce their rows of the single-process result
    x_all, out_all, ran_all, tr_all = _es_run(slice(0, 4), dev, None)
    assert ran_all == 4                  # (the first half of the batch alone would go on to 5 iterations at this threshold)
    assert _es_run(slice(0, 2), dev, None)[2] != ran_all or _es_run(slice(2, 4), dev, None)[2] != ran_all
    for r in range(world):
        xr, outr, ran_r, tr_r = res[r]["es"]
        assert ran_r == ran_all and [t[1:] for t in tr_r] == [t[1:] for t in tr_all]
        np.testing.assert_allclose([t[0] for t in tr_r], [t[0] for t in tr_all], rtol=1e-5)
        assert torch.equal(xr, x_all[2 * r: 2 * r + 2]) and torch.equal(outr, out_all[2 * r: 2 * r + 2])
