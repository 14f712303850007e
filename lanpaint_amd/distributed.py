"""One-node multi-GPU use of the Langevin path: batch-of-images replicas, one process per GPU.

The reference has no multi-device code at all; the path shards naturally
because every operation of lanpaint.py:56-157 is elementwise per latent element with per-row
scalars.  So the design is: shard the batch rows across ranks, broadcast what the ranks share
(mask, known latent, conditioning) ONCE at setup in a single packed RCCL broadcast over xGMI,
and run the think loop with NO collective in it.  The only optional exchange is the inner
early stop on a sharded batch, whose metric is defined over the whole batch tensor
(earlystop.py:52-55): the four partial sums are all-reduced so every rank takes the same
decision as a single process would.

`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) as torchrun / torch.distributed.run export them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


_FORCE_COLLECTIVES = False      # set by init(single_rank_group=True): a ONE-rank group goes through the real collectives


def _collectives_on(group=None) -> bool:
    """Is there somebody to talk to?  No process group: no.  A group of ONE rank: only when the real path was asked for
    (`init(single_rank_group=True)`, the self-test; or LANPAINT_AMD_FORCE_COLLECTIVES=1) -- a host application's own
    one-rank group (say a gloo group ComfyUI never created for us) must not turn every reduction into a collective: with
    gloo that is a host copy, with nccl a per-iteration RCCL call on the host-stopper path."""
    if not dist.is_initialized():
        return False
    if dist.get_world_size(group) > 1:
        return True
    return _FORCE_COLLECTIVES or os.environ.get("LANPAINT_AMD_FORCE_COLLECTIVES") == "1"


def init(backend: Optional[str] = None, device: Optional[torch.device] = None, single_rank_group: bool = False,
         timeout_s: Optional[float] = None) -> Tuple[int, int]:
    """Initialise the default process group from the environment.  World size 1 is a no-op unless `single_rank_group`: then
    a ONE-rank group is brought up over a file store (no port to race for), so that the collectives below really go through
    the library (RCCL for "nccl") instead of their no-group shortcuts -- `single_rank_selftest`, the GPU tests.
    `timeout_s`: rendezvous and collective timeout (a stuck peer raises instead of hanging the job)."""
    global _FORCE_COLLECTIVES
    import datetime
    rank, world, local_rank = env_world()
    if world <= 1 and not single_rank_group:
        return 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if timeout_s is not None:
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    if world <= 1:
        _FORCE_COLLECTIVES = True
        if not dist.is_initialized():
            import tempfile
            if backend == "nccl":
                device = device or torch.device("cuda", torch.cuda.current_device())
                torch.cuda.set_device(device)
                kw["device_id"] = device
            kw.setdefault("timeout", datetime.timedelta(seconds=120))
            store = tempfile.NamedTemporaryFile(prefix="lanpaint_amd_pg_", delete=False)
            store.close()
            try:
                os.unlink(store.name)                  # the FileStore creates it; a stale file would be read as an old rendezvous
            except OSError:
                pass
            dist.init_process_group(backend, init_method=f"file://{store.name}", rank=0, world_size=1, **kw)
        return 0, 1
    if not dist.is_initialized():
        if backend == "nccl":
            device = device or torch.device("cuda", local_rank)
            torch.cuda.set_device(device)
            kw["device_id"] = device
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def bind_to_device_numa(device_index: int) -> Optional[dict]:
    """Keep this process on the CPU cores of the NUMA node its GPU hangs off (one process per GPU on a two-socket node:
    a launch-bound loop notices doorbell writes and pinned-mailbox polls crossing the socket link).  Best effort --
    sysfs layout, permissions or a container may not allow it; returns what was done ({"numa_node", "cpus"}) or None."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"numa_node": node, "cpus": len(allowed)}
    except Exception:
        return None


def replica_seed(seed: int, rank: int) -> int:
    """Per-rank RNG seed (xi stream and initial noise): seed + rank."""
    return (int(seed) + int(rank)) & 0xFFFFFFFFFFFFFFFF


def shard_rows(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of the batch rows owned by `rank`; sizes differ by at most 1."""
    base, extra = divmod(int(global_batch), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _pack_header(tensors: Dict[str, torch.Tensor]):
    return [(k, tuple(t.shape), str(t.dtype).replace("torch.", "")) for k, t in tensors.items()]


def broadcast_job(tensors: Optional[Dict[str, torch.Tensor]], src: int = 0,
                  device: Optional[torch.device] = None, group=None, stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Broadcast the tensors every replica shares (prepared mask, known latent `y`, cond tensors)
    from `src` to all ranks: one tiny object broadcast for the layout, then ONE collective over a
    single packed byte buffer (xGMI is per-link bound: one large transfer, not many small ones).
    Non-src ranks may pass None.  Returns {name: tensor} on `device` on every rank.  `stats` (optional dict) receives
    {"bytes": size of the packed buffer, "ms": wall time of the one data collective on this rank, "backend"}."""
    if not _collectives_on(group):        # nobody to share with (no group; a one-rank group unless the real path was asked for)
        return {k: (t.to(device) if device is not None else t) for k, t in (tensors or {}).items()}
    rank = dist.get_rank(group)
    header = [_pack_header(tensors) if rank == src else None]
    dist.broadcast_object_list(header, src=src, group=group)
    layout = header[0]
    sizes = []
    for _name, shape, dtype in layout:
        n = 1
        for s in shape:
            n *= s
        sizes.append(n * torch.empty((), dtype=getattr(torch, dtype)).element_size())
    offsets, total = [], 0
    for sz in sizes:
        offsets.append(total)
        total += (sz + 15) // 16 * 16                      # keep every tensor 16-byte aligned in the pack
    buf_dev = device if device is not None else (next(iter(tensors.values())).device if tensors else torch.device("cpu"))
    buf = torch.empty(max(total, 1), dtype=torch.uint8, device=buf_dev)
    if rank == src:
        for (name, _shape, _dtype), off, sz in zip(layout, offsets, sizes):
            buf[off:off + sz] = tensors[name].contiguous().view(-1).view(torch.uint8).to(buf_dev)
    import time
    if buf.is_cuda:
        torch.cuda.synchronize(buf.device)
    t0 = time.perf_counter()
    dist.broadcast(buf, src=src, group=group)
    if buf.is_cuda:
        torch.cuda.synchronize(buf.device)
    if stats is not None:
        stats.update({"bytes": int(total), "ms": 1e3 * (time.perf_counter() - t0), "backend": dist.get_backend(group)})
    out = {}
    for (name, shape, dtype), off, sz in zip(layout, offsets, sizes):
        out[name] = buf[off:off + sz].view(getattr(torch, dtype)).reshape(shape).clone()
    return out


def reduce_throughput(elapsed_s: float, units: int, device: Optional[torch.device] = None, group=None) -> Tuple[float, int]:
    """(max elapsed over ranks, total units over ranks): whole-job throughput = units / elapsed."""
    if not _collectives_on(group):
        return float(elapsed_s), int(units)
    dev = device if (device is not None and dist.get_backend(group) == "nccl") else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    n = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), int(round(n.item()))


def collective_library_version() -> Optional[str]:
    """Version of the library behind the "nccl" backend (RCCL on ROCm), or None."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(int(x)) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def gather_rank_reports(report: dict, group=None) -> Optional[dict]:
    """Every rank contributes one small dict; rank 0 gets the evidence block of a multi-rank run:
    {backend, world_size, ranks_reporting, device_per_rank, rccl_version, per_rank_it_s, per_rank}.  None without a
    process group.  One object all-gather, outside any timed region."""
    if not _collectives_on(group):
        return None
    world = dist.get_world_size(group)
    reports = [None] * world
    dist.all_gather_object(reports, report, group=group)
    reports = [r for r in reports if isinstance(r, dict)]
    reports.sort(key=lambda r: r.get("rank", 0))
    backend = dist.get_backend(group)
    return {"backend": backend, "world_size": world, "ranks_reporting": len(reports),
            "device_per_rank": [r.get("device") for r in reports],
            "distinct_devices": len({(r.get("device"), r.get("pci_bus_id")) for r in reports}),
            "rccl_version": collective_library_version() if backend == "nccl" else None,
            "per_rank_it_s": [r.get("it_s") for r in reports], "per_rank": reports}


def all_reduce_stop_sums(acc: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the early-stop partial sums {sum(w d^2), sum(w)} x {inpaint, ring} over the ranks that
    share one (sharded) batch, in place; identity when no process group is up."""
    if _collectives_on(group):
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc


def single_rank_selftest(backend: Optional[str] = None, device_index: int = 0) -> dict:
    """ONE rank, the real library: bring up a one-rank process group ("nccl" = RCCL when a GPU is there) and push a job of
    SDXL-job size through every collective the N > 1 path issues -- broadcast_job (broadcast_object_list for the layout + ONE
    packed uint8 device broadcast), reduce_throughput (fp64 device all-reduces MAX / SUM), all_reduce_stop_sums,
    gather_rank_reports (all_gather_object) -- and compare what comes back with what went in, byte for byte.  Returns a
    JSON-able record; destroys the group it created."""
    global _FORCE_COLLECTIVES
    import time
    use_cuda = torch.cuda.is_available()
    backend = backend or ("nccl" if use_cuda else "gloo")
    dev = torch.device("cuda", device_index) if (use_cuda and backend == "nccl") else torch.device("cpu")
    rec = {"backend": backend, "world_size": 1, "device": str(dev), "ok": False}
    created = not dist.is_initialized()
    t0 = time.perf_counter()
    try:
        init(backend, dev if dev.type == "cuda" else None, single_rank_group=True)
        rec["init_s"] = time.perf_counter() - t0
        g = torch.Generator(device="cpu").manual_seed(1234)
        job = {"mask": (torch.rand((1, 4, 128, 128), generator=g) > 0.5).float(),            # 256 KiB fp32
               "y": torch.randn((1, 4, 128, 128), generator=g),
               "cond": torch.randn((1, 77, 2048), generator=g).to(torch.bfloat16),           # SDXL text states
               "pooled": torch.randn((1, 2816), generator=g).to(torch.bfloat16),
               "odd": torch.arange(7, dtype=torch.uint8)}                                    # (keeps the 16-byte padding honest)
        job = {k: v.to(dev) for k, v in job.items()}
        stats: dict = {}
        got = broadcast_job(job, src=0, device=dev, stats=stats)
        same = all(got[k].dtype == v.dtype and got[k].shape == v.shape and
                   torch.equal(got[k].contiguous().view(torch.uint8), v.contiguous().view(torch.uint8)) for k, v in job.items())
        t_max, n_sum = reduce_throughput(1.25, 150, dev if dev.type == "cuda" else None)
        acc = torch.arange(8, dtype=torch.float64, device=dev)
        acc2 = all_reduce_stop_sums(acc.clone())
        rep = gather_rank_reports({"rank": 0, "device": str(dev), "it_s": 1.0})
        rec.update({"broadcast_bytes": stats.get("bytes"), "broadcast_ms": stats.get("ms"), "tensors_byte_identical": bool(same),
                    "reduce_throughput_ok": bool(t_max == 1.25 and n_sum == 150), "all_reduce_identity_ok": bool(torch.equal(acc, acc2)),
                    "reports_gathered": None if rep is None else rep.get("ranks_reporting"),
                    "rccl_version": collective_library_version() if backend == "nccl" else None,
                    "collectives": ["broadcast_object_list", "broadcast(uint8 pack)", "all_reduce(f64 MAX)", "all_reduce(f64 SUM)",
                                    "all_reduce(f64 x8 SUM)", "all_gather_object"]})
        rec["ok"] = bool(same and rec["reduce_throughput_ok"] and rec["all_reduce_identity_ok"] and rec["reports_gathered"] == 1
                         and stats.get("backend") == backend)
    except Exception as e:                                     # the record says what failed; the caller decides what that means
        rec["error"] = repr(e)
    finally:
        if created and dist.is_initialized():
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:
                pass
            _FORCE_COLLECTIVES = False
    rec["total_s"] = time.perf_counter() - t0
    return rec
