#!/usr/bin/env python3
"""Soak run of the default engine: does anything grow with the number of sigma calls?

    python scripts/soak.py [calls=3000]

Four legs on one engine each, device memory (torch allocator), host RSS and the number of live captures sampled along the way:
  A  one job replayed `calls` times (the sampler's steady state);
  B  a NEW mask tensor on every call (a workflow that rebuilds its inputs per step: packs, rings and captures are keyed on it);
  C  shapes cycling through four latents with the inner early stop armed (workspaces, stop buffers and captures per shape);
  D  graph=True with a new mask tensor per call (one capture per call: the capture cache must stay bounded).
Prints one line per leg; exits 1 when device memory or RSS keeps growing over the second half of a leg."""
import os
import sys

import numpy as np
import psutil
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from lanpaint_amd import LanPaint               # noqa: E402

dev = torch.device("cuda", 0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
proc = psutil.Process()


def inputs(shape, flow, seed=0):
    sig = bench.flow_sigmas(8) if flow else bench.karras_sigmas(8)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    x0, y, noise, mask = bench.make_inputs(shape, flow, float(sig[0]), seed, dev, tt)
    s = torch.full((shape[0],), float(sig[3]), device=dev)
    return x0, y, noise, mask, s, bench.times_from_sigma(s, flow)


def sample():
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated(dev) / 2**20, proc.memory_info().rss / 2**20


def leg(name, n, step):
    marks = []
    for k in range(n):
        step(k)
        if k in (n // 10, n // 2, n - 1):
            marks.append(sample())
    (d0, r0), (d1, r1), (d2, r2) = marks
    grow_dev, grow_rss = d2 - d1, r2 - r1
    bad = grow_dev > 8.0 or grow_rss > 64.0          # second half of the leg: MiB
    print(f"leg {name}: {n} calls; device MiB at 10% / 50% / 100%: {d0:.1f} / {d1:.1f} / {d2:.1f}; host RSS MiB: {r0:.0f} / {r1:.0f} / {r2:.0f}"
          f"{'  <-- GROWS' if bad else ''}", flush=True)
    return bad


h = bench.HYPER
bad = False

shape, flow = (1, 4, 128, 128), False
x0, y, noise, mask, s, times = inputs(shape, flow)
eng = LanPaint(bench.StubBackbone(flow), 5, h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=flow)
mo = {}
x = x0.clone()
bad |= leg("A (one job replayed)", calls, lambda k: eng(x, y, noise, s, mask, times, mo, 0))
print("   captures alive:", len(eng._graphs), " iterations run:", eng.iterations_run, flush=True)

eng_b = LanPaint(bench.StubBackbone(flow), 5, h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=flow)


def step_b(k):
    m = mask.clone()
    m[..., : (k % 64) + 1, :] = 1.0
    eng_b(x, y, noise, s, m, times, mo, 0)


bad |= leg("B (a new mask tensor per call)", max(200, calls // 5), step_b)
print("   captures alive:", len(eng_b._graphs), flush=True)

shapes = [((1, 4, 64, 64), False), ((1, 4, 128, 128), False), ((2, 16, 32, 32), True), ((1, 16, 5, 30, 52), True)]
jobs = [inputs(sh, fl, seed=i) for i, (sh, fl) in enumerate(shapes)]
engs = {fl: LanPaint(bench.StubBackbone(fl), 6, h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=fl) for fl in (False, True)}
mo_es = {"lanpaint_semantic_stop": {"threshold": 8.0, "patience": 1}}


def step_c(k):
    (sh, fl), (x0_, y_, n_, m_, s_, t_) = shapes[k % 4], jobs[k % 4]
    engs[fl](x0_.clone(), y_, n_, s_, m_, t_, mo_es, 0)


bad |= leg("C (four shapes in turn, early stop armed)", max(400, calls // 3), step_c)
print("   captures alive:", {fl: len(e._graphs) for fl, e in engs.items()}, flush=True)

# D  forced graph mode with a new mask tensor per call: every call captures -- the capture cache has to stay bounded
eng_d = LanPaint(bench.StubBackbone(flow), 5, h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLOW=flow, graph=True)


def step_d(k):
    m = mask.clone()
    m[..., : (k % 64) + 1, :] = 1.0
    eng_d(x, y, noise, s, m, times, mo, 0)


bad |= leg("D (graph=True, a new mask tensor per call: a capture per call)", max(200, calls // 10), step_d)
print("   captures alive:", len(eng_d._graphs), flush=True)
bad |= len(eng_d._graphs) > 64
sys.exit(1 if bad else 0)
