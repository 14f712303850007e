#!/usr/bin/env python3
"""Where a sigma call's time goes at the headline shape, measured (MI355X box):

  * the launch-floor ladders of benchkit/floor.py: the captured graphs launched back to back from C, then with the update launch,
    then the engine's own C entry from a bare Python loop, then the measured loop -- for the drop-in engine, the Philox engine
    and the node-default schedule through KSamplerX0Inpaint;
  * a cProfile breakdown of the host side of one replayed sigma call, engine-direct and through the node path.

    python scripts/launch_floor.py > profiles/r06_host_sigma_call.md
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from benchkit import floor                                     # noqa: E402
from benchkit.extras import build_node_sampler                 # noqa: E402
from benchkit.workloads import Job                             # noqa: E402
from lanpaint_amd import _cabi                                 # noqa: E402

dev = torch.device("cuda", 0)
args = bench.parse_args([])


def table(title, d):
    print(f"\n### {title}\n")
    if "error" in d:
        print("error:", d["error"])
        return
    print("| layer | us per sigma call |\n|---|---|")
    for k, v in d["us_per_sigma_call"].items():
        if v is not None:
            print(f"| {k} | {v:.2f} |")
    rest = {k: v for k, v in d.items() if k not in ("us_per_sigma_call", "note")}
    print("\n```json\n" + json.dumps(rest, default=float) + "\n```\n" + d.get("note", ""))


def profile(fn, calls, top=16):
    pr = cProfile.Profile()
    pr.enable()
    fn()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
    txt = s.getvalue()
    body = txt[txt.index("   ncalls"):] if "   ncalls" in txt else txt
    print(f"(cProfile adds its own cost per Python call; {calls} sigma calls profiled)\n\n```\n" + body.rstrip()[:5000] + "\n```")


print("# Host cost and launch floor of one sigma call at C2 (1x4x128x128, 30 sigmas x 5), round 6\n")
print(f"device: {torch.cuda.get_device_name(dev)}; torch {torch.__version__}; produced by scripts/launch_floor.py")
table("engine-direct schedule, the drop-in engine (rng='torch', graph='auto', fp32 mask)", floor.engine_floor(_cabi, dev))
table("engine-direct schedule, rng='philox', graph=True, caller-packed mask", floor.engine_floor(_cabi, dev, rng="philox", philox_seed=0, graph=True))
table("node-default schedule through KSamplerX0Inpaint (drop-in engine)", floor.node_floor(_cabi, dev, args))

job = Job("c2_sdxl", dev)
eng = job.engine()
for _ in range(6):
    job.run(eng)
torch.cuda.synchronize()
print("\n### cProfile: host side of the replayed sigma call, engine-direct (5 schedule passes)\n")
profile(lambda: [job.run(eng) for _ in range(5)], 5 * job.n_sig)
k, node_pass, n_sig = build_node_sampler(args, dev)
for _ in range(8):
    node_pass()
torch.cuda.synchronize()
print("\n### cProfile: host side of the replayed sigma call, node path (5 schedule passes)\n")
profile(lambda: [node_pass() for _ in range(5)], 5 * n_sig)
