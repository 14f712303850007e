"""First contact with RCCL (VERDICT r04 next #2a): ONE rank, the real library.  torch.distributed "nccl" is RCCL on ROCm; a
one-rank group on the box's single MI355X pushes an SDXL-sized job through every collective the N > 1 path issues --
`broadcast_object_list` for the layout, ONE packed `uint8` device broadcast, the fp64 device all-reduces of the throughput
reduction and of the early-stop sums, `all_gather_object` for the rank reports -- and what comes back must equal what went
in, byte for byte.  In a child process: the pytest process keeps no process group, and a library fault cannot take the
suite down.  (RCCL refuses two ranks on one device, so more than one rank cannot run on this box: no scaling curve here.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-c", "import sys, json; sys.path.insert(0, %r); %s" % (ROOT, code)],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(300)
def test_one_rank_rccl_group_carries_the_job_byte_for_byte():
    rec = _child("from lanpaint_amd import distributed as d; print(json.dumps(d.single_rank_selftest()))")
    assert rec["backend"] == "nccl" and rec["device"].startswith("cuda"), rec
    assert rec["ok"], rec
    assert rec["tensors_byte_identical"] and rec["reduce_throughput_ok"] and rec["all_reduce_identity_ok"], rec
    assert rec["reports_gathered"] == 1 and rec["rccl_version"], rec          # the library behind "nccl" reports its version
    assert rec["broadcast_bytes"] >= 2 * 262144 + 77 * 2048 * 2 + 2816 * 2 + 7, rec


@pytest.mark.timeout(300)
def test_engine_runs_beside_a_live_one_rank_rccl_group():
    """The think loop next to a live RCCL communicator (its watchdog thread issues HIP calls of its own): sigma calls
    captured and replayed as hipGraphs, the job taken from an RCCL broadcast, the result equal to a run without any group."""
    code = r'''
import numpy as np, torch
from lanpaint_amd import LanPaint, distributed as d
import bench
dev = torch.device("cuda", 0)
def run(job_through_rccl):
    shape, flow, n_sig, n_think = bench.WORKLOADS["c1_sd15"]
    sig_np = bench.karras_sigmas(n_sig)[:6]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig_np[0]), 0, dev, tt)
    stats = {}
    if job_through_rccl:
        job = d.broadcast_job({"mask": mask, "y": y}, src=0, device=dev, stats=stats)
        mask, y = job["mask"], job["y"]
    eng = LanPaint(bench.StubBackbone(flow), n_think, 15.0, 5.0, 1.0, 0.2, rng="philox", philox_seed=3, graph=True)
    x = x0.clone()
    for s in sig_np:
        st = torch.full((1,), float(s), device=dev)
        den = eng(x, y, noise, st, mask, bench.times_from_sigma(st, flow), None, 0, n_steps=n_think)
        x = torch.lerp(den, x, 0.9)
    torch.cuda.synchronize()
    return x.cpu(), len(eng._graphs), stats
plain, g0, _ = run(False)
d.init("nccl", dev, single_rank_group=True)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
with_group, g1, stats = run(True)
t, n = d.reduce_throughput(1.0, 30, dev)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print(json.dumps({"equal": bool(torch.equal(plain, with_group)), "graphs": [g0, g1], "bytes": stats.get("bytes"),
                  "backend": stats.get("backend"), "reduced": [t, n], "rccl": d.collective_library_version()}))
'''
    rec = _child(code)
    assert rec["equal"] and rec["graphs"] == [1, 1] and rec["backend"] == "nccl" and rec["bytes"] >= 2 * 65536, rec
    assert rec["reduced"] == [1.0, 30] and rec["rccl"], rec
