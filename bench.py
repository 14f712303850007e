#!/usr/bin/env python3
"""bench.py -- Langevin think-iterations/sec of the HIP path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over the whole sigma schedule of the workload (C2: SDXL 1x4x128x128 latent per GPU,
30 Karras sigmas x 5 think iterations = 150 think iterations + 30 final denoise calls), engine driven directly (SURVEY.md
8d), stub backbone x -> (0.9x, 0.8x), synthetic inputs resident in HBM before the timed region.  value = think iterations
of ALL ranks / max-over-ranks wall time.  One process per GPU; ranks are independent replicas (mask / known latent / cond
are broadcast from rank 0 over RCCL at set-up; no collective inside the loop) -> weak scaling.

The timed configuration is the DROP-IN one: the engine exactly as `LanPaint(Model, NSteps, Friction, Lambda, Beta, StepSize,
IS_FLUX, IS_FLOW)` builds it (reference lanpaint.py:8) -- rng="torch" (the reference's torch.randn_like stream, generated
in-kernel), graph="auto", the reference's fp32 mask (which the engine bit-packs by itself).  --rng / --graph / --mask-format
select other configurations; `summary.philox_bits_it_s` carries the fastest one beside the headline.

ONE JSON line (< 6 KB, the last line of stdout) carries metric/value/config, `parity_check` (the timed engine against the CPU
oracle before anything is timed), `roofline` (the dominant kernel: algorithmic bytes per launch / mean launch duration from
HIP events on the launch stream, THIS run), `cpu_baseline` (the CPU port of the reference on this box's host cores, bounded
sample) and a few scalars; everything else goes to the side-car file `bench_extras.json` (benchkit/line.py).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

# RCCL / IPC between the ranks of one node needs dmabuf IPC on this driver stack; must be in the environment before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit import line as bl                                             # noqa: E402
from benchkit.line import build_line, engine_keywords, error_line, pci_bus_id   # noqa: E402,F401
from benchkit.cpu import cpu_baseline, compact_cpu, usable_cpus as _usable_cpus   # noqa: E402,F401
from benchkit.parity import PARITY_TOL, Bf16StubOracle, parity_check, check_job   # noqa: E402,F401
from benchkit.ranks import free_port as _free_port, rank_environments, spawn_ranks, summarise_dist   # noqa: E402,F401
from benchkit.roofline import (BYTES_PER_EL_STEADY, HBM_PEAK_GBPS, _profile_order, committed_profile, compact_roofline,   # noqa: E402,F401
                               graph_burst_us_per_launch, measure_steady_launch, pmc_traffic, roofline_fields, shape_regime,
                               standalone_step, steady_bytes_per_launch, steady_kernel_name, timed_burst, tune_from_env, warm_burst)
from benchkit.workloads import (HYPER, WORKLOADS, Job, StubBackbone, StubSampling, attach_mask_format, euler_ratios,   # noqa: E402,F401
                                flow_sigmas, karras_sigmas, make_inputs, make_mask, mask_description, schedule_pass,
                                shared_conditioning, temporal_known_frames, times_from_sigma)


def run_gpu(args):
    import torch.distributed as dist
    from lanpaint_amd import _cabi
    from lanpaint_amd import distributed as lpd

    rank, world, local_rank = lpd.env_world()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.set_num_threads(1)            # N ranks on a box whose container grants few CPUs: no OpenMP teams in front of the timed region
    n_dev = max(1, torch.cuda.device_count())
    dev_index = local_rank % n_dev                      # identity on an N-GPU node; lets a 1-GPU box rehearse N > 1
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.dist_backend
    if world > 1 and backend == "nccl" and n_dev < world:
        # RCCL refuses two ranks on one device ("Duplicate GPU detected"): a box with fewer GPUs than ranks can only REHEARSE
        # the N > 1 path over gloo, and only when asked to (the line's `collective` says which library carried the run)
        if not args.allow_gloo_fallback:
            if rank == 0:
                print(f"bench.py: --gpus {world} over RCCL needs {world} devices, this box has {n_dev}.  Pass --dist-backend gloo "
                      "(or --allow-gloo-fallback) to rehearse the multi-rank path on fewer devices; no line is printed.",
                      file=sys.stderr, flush=True)
            raise SystemExit(4)
        backend = "gloo"
    affinity0 = os.sched_getaffinity(0)
    numa = lpd.bind_to_device_numa(dev_index) if not args.no_numa_bind else None
    t_proc0 = time.perf_counter()
    rng_name = args.rng or "torch"
    kernel_rng = "torch" if rng_name.startswith("torch") else "philox"

    # Per-dispatch event timing of the dominant kernel, taken FIRST in the process (before graph captures and other streams
    # exist) and by EVERY rank, before the process group exists: no rank ever waits in a collective for another's local work.
    try:
        steady = measure_steady_launch(_cabi, dev, workload=args.workload, launches=120, mask_kind=args.mask, rng=kernel_rng,
                                       mask_format="bits")          # (a binary fp32 / u8 mask reaches the kernels bit-packed)
    except Exception as e:
        steady = {"error": repr(e)}

    job = Job(args.workload, dev, seed=lpd.replica_seed(args.seed, rank), mask_kind=args.mask, mask_format="f32")
    def fail(what, e):
        """A rendezvous / RCCL failure: a line with `error` naming the rank and the step (RCCL's own NCCL_DEBUG=WARN output is
        on stderr), exit code 5 -- not a hang: init and every collective run under --dist-timeout."""
        if rank == 0:
            bl.emit_line(error_line(args, f"rank {rank}: {what} failed over {backend}: {e!r}"))
        else:
            print(f"bench.py: rank {rank}: {what} failed over {backend}: {e!r}", file=sys.stderr, flush=True)
        raise SystemExit(5)

    if world > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")
    t_init0 = time.perf_counter()
    try:
        lpd.init(backend, dev, timeout_s=args.dist_timeout)       # "nccl" IS RCCL on ROCm; no-op at world size 1
    except Exception as e:
        fail("init_process_group", e)
    t_init = time.perf_counter() - t_init0
    bcast, cond = {}, None
    if world > 1:      # all replicas inpaint the same image with the same mask under the same conditioning: ONE packed
        # broadcast from rank 0 at set-up -- mask, known latent, cond tensors -- and nothing afterwards
        shared = dict({"mask": job.mask, "y": job.y}, **shared_conditioning(args.workload, dev, args.seed)) if rank == 0 else None
        try:
            got = lpd.broadcast_job(shared, src=0, device=dev, stats=bcast)
        except Exception as e:
            fail("broadcast_job", e)
        job.mask, job.y = got["mask"], got["y"]
        cond = {k: v for k, v in got.items() if k not in ("mask", "y")}
        job.renoise()
    if len(job.shape) == 5 and (args.mask or "temporal") == "temporal" and rank == 0:
        # job set-up as a workflow does it: the pixel-resolution video mask goes through reshape_mask's video path
        # (lp_reshape_mask: nearest-exact + 5-tap temporal union) and must give the latent mask used here
        from lanpaint_amd import nodes as lpn
        frames = 4 * (job.shape[2] - 1) + 1
        pix = torch.zeros((frames, job.shape[3] * 8, job.shape[4] * 8), device=dev)
        pix[frames // 2:] = 1.0                                    # ComfyUI denoise mask: 1 = inpaint
        lat = lpn.reshape_mask(pix, (1,) + tuple(job.shape[1:]), video_inpainting=True)
        assert torch.equal(1.0 - (lat > 0.5).float(), job.mask[:1]), "reshape_mask disagrees with the analytic temporal mask"
        del pix, lat
    job.mask = attach_mask_format(job.mask, args.mask_format)      # once per job, outside the timed region ("f32": nothing)
    engine = job.engine(**dict(engine_keywords(args), **({"philox_seed": lpd.replica_seed(args.seed, rank)} if rng_name == "philox" else {})))
    torch.manual_seed(lpd.replica_seed(args.seed, rank))           # rng="torch": the device generator IS the noise stream
    waits = []

    def barrier():
        """dist.barrier + device sync; remembers how long THIS rank waited in the collective."""
        if world > 1:
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            dist.barrier()
            waits.append(time.perf_counter() - t_b)
        torch.cuda.synchronize()

    # Before anything is timed: does THIS engine, in THIS configuration, compute what the reference computes?  One schedule
    # pass against the CPU oracle on the same draws (bounded on the video-latent shapes and at N > 1, where every rank checks
    # its own replica).  No `value` is printed when the pass is off by more than the stated tolerance.
    parity = None
    if not args.no_parity_check:
        n_par = args.parity_sigmas if args.parity_sigmas > 0 else (job.n_sig if (job.n_el <= 512 * 1024 and world == 1) else (8 if job.n_el <= 512 * 1024 else 2))
        try:
            parity = check_job(job, engine, max_sigmas=n_par)
        except Exception as e:
            parity = {"ok": False, "error": repr(e)}

    # untimed set-up before the W warm-up steps: graph capture and lazy initialisation, then ~0.3 s of the very workload so
    # the clocks have ramped
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        job.run(engine)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        job.run(engine)
    t_setup = time.perf_counter() - t_proc0
    barrier()
    first_wait = waits[-1] if waits else None
    it0, cpu0, t0 = engine.iterations_run, time.process_time(), time.perf_counter()
    for _ in range(args.steps):
        x_last = job.run(engine)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0             # this rank's own K steps, before it waits for the others
    cpu_busy = time.process_time() - cpu0
    barrier()
    elapsed = time.perf_counter() - t0
    iters_local = engine.iterations_run - it0
    assert torch.isfinite(x_last).all(), "bench produced non-finite latents"

    tmax, iters_total = lpd.reduce_throughput(elapsed, iters_local, dev)
    dist_info = lpd.gather_rank_reports({
        "rank": rank, "device": f"cuda:{dev_index}", "device_name": torch.cuda.get_device_name(dev),
        "pci_bus_id": pci_bus_id(dev_index), "numa": numa, "pid": os.getpid(), "steps": args.steps, "iterations": int(iters_local),
        "elapsed_s": elapsed, "it_s": iters_local / elapsed, "final_checksum": float(x_last.double().sum().item()),
        "rows": int(job.shape[0]), "own_elapsed_s": own_elapsed, "own_it_s": iters_local / own_elapsed,
        "process_time_over_elapsed": cpu_busy / own_elapsed, "t_first_barrier_wait_s": first_wait,
        "closing_barrier_wait_s": (waits[-1] if len(waits) > 1 else None), "setup_s": t_setup, "init_process_group_s": t_init,
        "steady_launch_us": (steady or {}).get("mean_launch_us"), "cpus_allowed": len(os.sched_getaffinity(0)),
        "parity_ok": (None if parity is None else bool(parity.get("ok"))), "parity_mse_x": (None if parity is None else parity.get("mse_x")),
        "captured_calls": len(engine._graphs), "launch_modes": (None if parity is None else parity.get("launch_modes")),
        # what this rank holds of the shared job after the broadcast: equal on every rank, or the broadcast did not deliver
        "shared_checksum": (float(sum(t.double().sum().item() for t in (job.mask, job.y, *(cond or {}).values()))) if world > 1 else None)})
    if dist_info is not None and rank == 0:
        per = dist_info["per_rank"]
        dist_info.update({"backend_requested": args.dist_backend, "broadcast_bytes": bcast.get("bytes"), "broadcast_ms": bcast.get("ms"),
                          "launcher": os.environ.get("LANPAINT_BENCH_LAUNCHER", "external"), "collectives_in_timed_region": 0,
                          "init_process_group_s": t_init,
                          "shared_tensors": {k: list(v.shape) for k, v in dict({"mask": job.mask, "y": job.y}, **(cond or {})).items()},
                          "shared_checksums_equal": len({r.get("shared_checksum") for r in per}) == 1,
                          "global_rows": sum(r.get("rows", 0) for r in per),
                          "parity_ok_all_ranks": all(r.get("parity_ok") is not False for r in per),
                          "slowest_rank": max(per, key=lambda r: r.get("own_elapsed_s") or 0.0).get("rank"),
                          "own_it_s_spread": [min(r.get("own_it_s") or 0.0 for r in per), max(r.get("own_it_s") or 0.0 for r in per)]})

    # run-to-run spread of the same measurement: further blocks of K steps, each bracketed like the timed region
    # (the headline `value` stays the first block -- exactly K steps, as the contract says)
    repeat_values = []
    for _ in range(max(0, args.repeats)):
        barrier()
        itr, tr = engine.iterations_run, time.perf_counter()
        for _ in range(args.steps):
            job.run(engine)
        barrier()
        t_r, n_r = lpd.reduce_throughput(time.perf_counter() - tr, engine.iterations_run - itr, dev)
        repeat_values.append(n_r / t_r)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    value = iters_total / tmax
    parity_failed = (parity is not None and not parity.get("ok")) or (dist_info is not None and not dist_info.get("parity_ok_all_ranks", True))

    # ---- rank 0, outside every timed region: the secondary numbers (never at the cost of the line) ----
    sidecar = {"roofline": steady, "parity_check": parity, "dist": dist_info, "argv": sys.argv[1:]}
    summary = {"repeats_median": float(np.median(repeat_values)) if repeat_values else None,
               "repeats_min": min(repeat_values) if repeat_values else None, "repeats_max": max(repeat_values) if repeat_values else None}
    live = None
    if world == 1 and args.pmc:
        try:
            from benchkit.extras import live_pmc_traffic
            live = live_pmc_traffic(args.workload, kernel_rng, args.mask)
            sidecar["live_pmc"] = live
            if live.get("traffic_bytes_per_launch") and isinstance(steady, dict) and "frac" in steady:
                steady.update(roofline_fields(steady["algorithmic_bytes_per_launch"], steady["duration_used_us"],
                                              live["traffic_bytes_per_launch"], steady["every_stream_bytes_per_launch"],
                                              traffic_source="live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch in this run"))
        except Exception as e:
            sidecar["live_pmc"] = {"error": repr(e)}
    if world == 1 and not args.no_summary:
        try:
            from benchkit.extras import summary_scalars
            s, detail = summary_scalars(args, dev, None if parity_failed else value, _cabi)
            summary.update(s)
            sidecar["summary_detail"] = detail
        except Exception as e:
            sidecar["summary_detail"] = {"error": repr(e)}
    if world == 1 and args.extras:
        try:
            from benchkit.extras import extra_measurements
            sidecar["extras"] = extra_measurements(args, dev, value)
        except Exception as e:
            sidecar["extras"] = {"error": repr(e)}
    cpu = None
    if not args.no_cpu_baseline:     # rank 0 only, after the process group is gone: host time nobody waits for
        try:
            try:
                os.sched_setaffinity(0, affinity0)      # the baseline is the port on the BOX's host cores, not on one NUMA node's
            except Exception:
                pass
            cpu = cpu_baseline(args.workload, args.cpu_seconds)
        except Exception as e:
            cpu = {"error": repr(e)}
    sidecar["cpu_baseline"] = cpu
    line = build_line(args, job.shape, job.n_sig, job.n_think, job.flow, value, tmax, parity, steady, cpu, summary, dist_info,
                      captured_calls=len(engine._graphs), mask_packed=getattr(job.mask, "_lp_bits", None) is not None)
    line["extras_file"] = bl.write_sidecar(bl.sidecar_path(args.sidecar), dict(sidecar, line=line))
    bl.emit_line(line)
    if parity_failed:
        raise SystemExit(3)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (C2: 400 x 1.3 ms = 0.5 s)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="further timed blocks of --steps steps (summary.repeats_*)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3, help="untimed set-up (graph capture, lazy init, clock ramp)")
    ap.add_argument("--workload", default="c2_sdxl", choices=sorted(WORKLOADS))
    ap.add_argument("--rng", default=None, choices=["philox", "torch", "torch-eager"],
                    help="default: the engine's own default, 'torch' (the reference's randn stream reproduced in-kernel)")
    ap.add_argument("--graph", type=int, default=None, help="default: the engine's own default, 'auto'; 1 / 0 force replay / eager")
    ap.add_argument("--mask-format", default="f32", choices=["bits", "u8", "f32"],
                    help="f32 (default): the reference's fp32 mask, bit-packed by the engine itself; bits: packed by the caller")
    ap.add_argument("--mask", default=None, choices=["box", "temporal", "blob"])
    ap.add_argument("--model-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds before a stuck rendezvous / collective raises")
    ap.add_argument("--allow-gloo-fallback", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--parity-sigmas", type=int, default=0)
    ap.add_argument("--no-numa-bind", action="store_true")
    ap.add_argument("--no-summary", action="store_true", help="skip the summary scalars (philox / node schedule / HBM-bound shapes)")
    ap.add_argument("--no-large-shape", action="store_true", help="skip the HBM-bound shapes of the summary")
    ap.add_argument("--extras", type=int, default=0, help="1: also run the secondary measurements (side-car file only)")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic live (re-runs the steady launch under rocprofv3 --pmc)")
    ap.add_argument("--sidecar", default=None, help="path of the side-car JSON (default: bench_extras.json next to bench.py)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher (the torch.distributed.run form keeps working -- it sets WORLD_SIZE)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:], os.path.abspath(__file__)))
    bl.claim_stdout()
    run_gpu(args)


if __name__ == "__main__":
    main()
