"""bench.py's contract pieces that do not need a GPU: workload table, schedules, the cpu_baseline leg
(bounded sample, JSON-able dict), traffic lookup, and that bench/smoke are the only non-test users of oracle/."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workloads_match_baseline_configs():
    import bench
    assert bench.WORKLOADS["c2_sdxl"] == ((1, 4, 128, 128), False, 30, 5)
    assert bench.WORKLOADS["c1_sd15"] == ((1, 4, 64, 64), False, 20, 5)
    assert bench.WORKLOADS["c3_sdxl_b4"][0] == (4, 4, 128, 128)
    assert bench.WORKLOADS["c4_flux"] == ((1, 16, 64, 64), True, 28, 10)
    assert bench.WORKLOADS["c5_wan"] == ((1, 16, 21, 60, 104), True, 30, 5)
    s = bench.karras_sigmas(30)
    assert len(s) == 30 and abs(s[0] - 14.6146) < 1e-3 and abs(s[-1] - 0.0292) < 1e-4 and all(a > b for a, b in zip(s, s[1:]))
    f = bench.flow_sigmas(28)
    assert len(f) == 28 and 0 < f[-1] < f[0] < 1.0
    assert bench.BYTES_PER_EL_STEADY == 36 and bench.HBM_PEAK_GBPS == 8000.0


def test_cpu_baseline_leg_is_bounded_and_json_serialisable():
    """cpu_baseline: the CPU port of the reference (oracle/lanpaint_oracle.py on torch-CPU tensors) on this host's cores,
    kind "port" -- the Python reference does not travel to the GPU box in any form, so nothing else can be timed there."""
    import bench
    out = bench.cpu_baseline("c1_sd15", 0.6)
    json.dumps(out)
    assert out["kind"] == "port" and out["unit"] == "think-iterations/s" and out["value"] > 0
    n_all = bench._usable_cpus()
    assert out["cpu_model"] and out["usable_cpus"] == n_all and "1" in out["threads"]
    assert str(n_all) in (set(out["threads"]) | set(out["threads_not_sampled"] or {})) and int(out["cores"]) in map(int, out["threads"])
    assert all(1 <= v["passes"] <= 5 and v["sigma_calls_per_pass"] >= 1 for v in out["threads_detail"].values())   # median of <= 5, bounded sample
    assert out["leg_seconds"] < 20.0 and "oracle/lanpaint_oracle.py" in out["sample"]
    c = bench.compact_cpu(out)
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and len(json.dumps(c)) < 900


def test_nothing_of_the_reference_travels():
    """The reference is a Python package: it is imported in the build container to write tests/golden/*.npz (make_golden.py)
    and never shipped -- no source, no bytecode.  No loader for staged reference bytecode exists any more, the build writes
    nothing under oracle/_ref, and nothing the GPU box runs mentions /root/reference outside a docstring or a generator script."""
    assert not os.path.exists(os.path.join(ROOT, "oracle", "ref_engine.py")) and not os.path.exists(os.path.join(ROOT, "oracle", "build_ref.py"))
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    assert not os.path.isdir(ref_dir) or not [f for _r, _d, fs in os.walk(ref_dir) for f in fs if f.endswith((".pyc", ".py"))]
    for f in ("bench.py", "__graft_entry__.py", *[os.path.join("benchkit", n) for n in os.listdir(os.path.join(ROOT, "benchkit")) if n.endswith(".py")]):
        src = open(os.path.join(ROOT, f)).read()
        assert "ref_engine" not in src and "load_reference" not in src and "py_compile" not in src, f


def _synthetic_line(n_ranks):
    """A headline line with every block filled the way a real run fills it (long strings included) and an N-rank dist block."""
    import bench
    args = bench.parse_args(["--gpus", str(n_ranks), "--steps", "20", "--warmup", "5"])
    steady = bench.roofline_fields(32.125 * 65536, 3.4, 2173952, 32.125 * 65536)
    steady.update({"bytes_model": {"x": 1.0} , "kernel": bench.steady_kernel_name("c2_sdxl", "torch", "bits"), "regime": bench.shape_regime(65536)[1],
                   "launches_timed": 120, "mean_launch_us": 3.4, "committed_profile": bench.committed_profile("c2_sdxl"),
                   "timer": "t" * 300, "workload": "w" * 200, "frac_rocprofv3": 0.0763})
    cpu = {"value": 827.71842716, "unit": "think-iterations/s", "cores": 1, "kind": "port", "sample": "s" * 260,
           "threads": {"1": 827.7, "16": 753.2}, "threads_detail": {"1": {"x": "y" * 500}}, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
           "host_cpus": 256, "usable_cpus": 16, "port_over_reference_build_container": 1.0668364918311142, "leg_seconds": 4.2}
    parity = {"mse_x": 4.9e-12, "mse_denoised_max": 6.3e-11, "tolerance": 1e-5, "ok": True, "sigmas_checked": 30, "launch_modes": {"torch": 30},
              "checker": "c" * 400, "draws": 270}
    summary = {"repeats_median": 119000.123456, "repeats_min": 118000.1, "repeats_max": 119900.9, "philox_bits_it_s": 119779.86,
               "value_over_philox_bits": 0.98, "node_default_schedule_it_s": 95000.0, "hbm_frac_c5_wan": 0.79,
               "hbm_frac_c5_wan_torch_stream": 0.6, "hbm_frac_past_l3": 0.78, "launch_floor_it_s": 128000.0}
    dist_info = None
    if n_ranks > 1:
        per = [{"rank": r, "device": f"cuda:{r}", "device_name": "AMD Instinct MI355X", "pci_bus_id": "0000:%02x:00.0" % r, "numa": {"numa_node": r // 4, "cpus": 64},
                "pid": 1000 + r, "steps": 20, "iterations": 3000, "elapsed_s": 0.0251 + r * 1e-4, "it_s": 119000.0 - r, "final_checksum": 1.5 + r,
                "rows": 4, "own_elapsed_s": 0.025, "own_it_s": 120000.0 - r, "process_time_over_elapsed": 0.99, "t_first_barrier_wait_s": 0.01 * r,
                "closing_barrier_wait_s": 0.001, "setup_s": 3.2, "init_process_group_s": 1.1, "steady_launch_us": 3.4 + 0.01 * r, "cpus_allowed": 16,
                "parity_ok": True, "parity_mse_x": 5e-12, "captured_calls": 1, "launch_modes": {"torch": 8}, "shared_checksum": 12.5} for r in range(n_ranks)]
        dist_info = {"backend": "nccl", "world_size": n_ranks, "ranks_reporting": n_ranks, "device_per_rank": [r["device"] for r in per],
                     "distinct_devices": n_ranks, "rccl_version": "2.22.3", "per_rank_it_s": [r["it_s"] for r in per], "per_rank": per,
                     "backend_requested": "nccl", "broadcast_bytes": 845328, "broadcast_ms": 0.4, "launcher": "external",
                     "collectives_in_timed_region": 0, "init_process_group_s": 1.1, "shared_tensors": {"mask": [4, 4, 128, 128]},
                     "shared_checksums_equal": True, "global_rows": 4 * n_ranks, "parity_ok_all_ranks": True, "slowest_rank": n_ranks - 1,
                     "own_it_s_spread": [119990.0, 120000.0]}
    line = bench.build_line(args, (1, 4, 128, 128), 30, 5, False, 119779.8638084328, 0.0250459, parity, steady, cpu, summary, dist_info,
                            captured_calls=1, mask_packed=True)
    line["extras_file"] = "bench_extras.json"
    return line


def test_headline_line_is_small():
    """the round-5 line was 23.5 KB and the driver could not parse it.  The line bench.py prints is bounded:
    < 6000 bytes with every block filled, at N = 1 and with an eight-rank `dist` block (per-rank reports go to the side-car),
    the required keys present, `roofline` and `cpu_baseline` carrying the fields the contract names."""
    from benchkit import line as bl
    for n in (1, 8):
        out = bl.bounded(_synthetic_line(n))
        text = json.dumps(out)
        assert len(text) < bl.MAX_LINE_BYTES == 6000, (n, len(text))
        assert json.loads(text)["value"] > 0 and "truncated" not in out
        for k in bl.REQUIRED_KEYS:
            assert k in out, k
        r, c = out["roofline"], out["cpu_baseline"]
        assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
                          "duration_used_us", "kernel"} and 0 < r["frac"] <= 1
        assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and "threads_detail" not in c
        assert out["config"]["rng"] == "torch" and "drop-in" in out["config"]["engine"]
        if n == 8:
            d = out["dist"]
            assert "per_rank" not in d and d["world_size"] == 8 and d["it_s"]["min"] <= d["it_s"]["median"] <= d["it_s"]["max"]
            assert d["slowest_rank"] == 7 and d["distinct_final_checksums"] == 8 and out["collective"] == "rccl"
    # the bound holds even when somebody stuffs the optional blocks: they are dropped, never the required keys
    fat = _synthetic_line(8)
    fat["summary"] = {f"k{i}": "x" * 100 for i in range(80)}
    out = bl.bounded(fat)
    assert len(json.dumps(out)) < 6000 and "summary" in out["truncated"] and out["roofline"]["frac"] > 0 and out["value"] > 0


def test_default_run_is_the_drop_in_configuration():
    """`python bench.py` with no flag times the engine as `LanPaint(Model, NSteps, Friction, Lambda, Beta,
    StepSize, IS_FLUX, IS_FLOW)` builds it -- no optional keyword -- on the reference's fp32 mask; extras are off by default."""
    import bench
    a = bench.parse_args([])
    assert a.rng is None and a.graph is None and a.mask_format == "f32" and a.extras == 0 and a.gpus == 1
    assert bench.engine_keywords(a) == {}
    assert bench.engine_keywords(bench.parse_args(["--rng", "philox", "--graph", "1"])) == {"rng": "philox", "graph": True}


def test_pmc_traffic_lookup_reads_committed_profile():
    import bench
    t = bench.pmc_traffic("c2_sdxl")
    assert t is None or 0.9 * 36 * 65536 < t < 1.3 * 36 * 65536
    assert bench.pmc_traffic("no_such_workload") is None


def test_roofline_block_is_a_bandwidth_statement():
    """`frac_algorithmic` (36 B x elements / duration) and `frac_counter` (min(algorithmic, PMC) bytes / duration) side by
    side; a committed profile is quoted next to the live number with its file name, never folded into it."""
    import bench
    r = bench.roofline_fields(36 * 65536, 4.0, 2167808)
    assert abs(r["achieved"] - 36 * 65536 / 4e-6 / 1e9) < 1e-6 and r["frac"] == r["frac_algorithmic"] == r["achieved"] / 8000.0
    assert abs(r["frac_counter"] - 2167808 / 4e-6 / 1e9 / 8000.0) < 1e-12 and r["bound"] == "hbm" and r["peak"] == 8000.0
    r2 = bench.roofline_fields(100, 1.0, 250)               # counters above the algorithmic bytes: re-reads do not earn credit
    assert r2["frac_counter"] == r2["frac_algorithmic"]
    assert bench.roofline_fields(100, 1.0, None)["frac_counter"] is None
    names = ["r03_x.json", "r04a_x.json", "r04_x.json", "r04b_x.json", "r10_x.json", "r9_x.json", "notes.json"]
    assert sorted(names, key=bench._profile_order) == ["notes.json", "r03_x.json", "r04a_x.json", "r04b_x.json", "r04_x.json", "r9_x.json", "r10_x.json"]
    c = bench.committed_profile("c2_sdxl")
    assert c["kernel_durations_file"].startswith("r") and c["pmc_traffic_file"].startswith("r")
    assert c["rocprofv3_mean_launch_us"] > 0 and c["pmc_traffic_bytes_per_launch"] > 0
    assert bench.committed_profile("nope")["rocprofv3_mean_launch_us"] is None
    assert "L3-resident" in bench.shape_regime(2096640)[1] and "past the 256 MiB" in bench.shape_regime(33546240)[1]
    assert "L2" in bench.shape_regime(65536)[1]


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lanpaint_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liblanpaint_oracle" not in src and "ref_engine" not in src and "_lanpaint_reference_engine" not in src, f


def test_roofline_fraction_is_a_fraction_in_every_committed_round5_line():
    """`roofline.frac` above 1 is not a roofline (round 4's C5 line printed 1.09: 36 B per element against a
    launch that legitimately skips streams).  Every bench line committed from round 5 on carries 0 < frac <= 1 and
    frac_counter <= 1 in its `roofline*` blocks; the every-operand figure lives in `frac_every_stream`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05*_bench_*.json")))
    assert files, "no committed round-5 bench lines under profiles/"

    def blocks(line):
        for key in ("roofline", "roofline_hbm_bound_shape", "roofline_hbm_past_l3"):
            b = line.get(key)
            if isinstance(b, dict) and "frac" in b:
                yield key, b
                if isinstance(b.get("region_aware_streams"), dict):
                    yield key + ".region_aware_streams", b["region_aware_streams"]
        for k, b in (line.get("bf16_heads") or {}).items():
            if isinstance(b, dict) and "frac" in b:
                yield "bf16_heads." + k, b

    seen = 0
    for f in files:
        line = json.load(open(f))
        for key, b in blocks(line):
            seen += 1
            assert 0.0 < b["frac"] <= 1.0, (os.path.basename(f), key, b["frac"])
            assert b.get("frac_counter") is None or 0.0 < b["frac_counter"] <= 1.0, (os.path.basename(f), key)
            assert b["algorithmic_bytes_per_launch"] <= b["every_stream_bytes_per_launch"] + 1e-6
            assert b.get("traffic") is None or "committed_constant" in b["traffic_source"]
    assert seen >= len(files)


def test_steady_bytes_model():
    """The bytes a steady launch REQUIRES (bench.steady_bytes_per_launch): SURVEY 8(d)'s 36 B with the mask's own width, and for
    the region-aware launches (bit mask, > 512 Ki elements) the mask actually used."""
    import numpy as np
    import bench
    half = np.zeros((1, 4, 8, 8), np.float32)
    half[..., :4] = 1.0
    r = bench.steady_bytes_per_launch(half, "f32", 65536)
    assert r["bytes_per_element_every_stream"] == 36.0 and r["required"] == r["every_stream"] == 36 * 65536
    r = bench.steady_bytes_per_launch(half, "bits", 65536)                       # latency-bound size: every operand streamed
    assert r["bytes_per_element_required"] == 32.125 and not r["region_aware_launch"]
    r = bench.steady_bytes_per_launch(half, "bits", 2096640)                     # streaming size, 50 % box
    assert r["region_aware_launch"] and abs(r["bytes_per_element_required"] - 26.125) < 1e-9
    assert r["bytes_per_element_every_stream"] == 32.125
    mask = bench.make_mask((1, 16, 21, 60, 104), "temporal")                     # C5: latent frames >= 8 of 21 inpainted
    r = bench.steady_bytes_per_launch(mask, "bits", 2096640)
    assert abs(r["known_fraction"] - 8 / 21) < 1e-9 and abs(r["bytes_per_element_required"] - (20.125 + 4 * 13 / 21 + 8 * 8 / 21)) < 1e-9
    assert bench.steady_bytes_per_launch(mask, "bits", 2096640, every_stream=True)["required"] == 32.125 * 2096640
    r = bench.steady_bytes_per_launch(mask, "bits", 2096640, model_dtype="bf16")
    assert r["bytes_per_element_every_stream"] == 26.125


def test_every_committed_round6_line_is_small_and_complete():
    """The lines bench.py printed on the GPU boxes of round 6 (profiles/r06*_bench_*.json, side-cars excluded): each under
    6000 bytes, every required key present, `roofline` and `cpu_baseline` carrying the contract's fields, 0 < frac <= 1, parity ok,
    the default runs in the drop-in configuration, the eight-rank rehearsals with a compact `dist` block."""
    import glob
    from benchkit import line as bl
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_*.json"))) if "sidecar" not in f]
    assert len(files) >= 8, files
    for f in files:
        text = open(f).read().strip()
        name = os.path.basename(f)
        assert len(text) < bl.MAX_LINE_BYTES, (name, len(text))
        line = json.loads(text)
        for k in bl.REQUIRED_KEYS:
            assert k in line, (name, k)
        r = line["roofline"]
        assert 0.0 < r["frac"] <= 1.0 and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["kernel"].startswith("lp::lp_step_kernel<"), name
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["duration_used_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"], name
        assert r["traffic"] is None or ("live" in r["traffic_source"] or "committed_constant" in r["traffic_source"]), name
        assert line["parity_check"]["ok"] and line["value"] > 0 and line["vs_baseline"] is None and line["dtype"] == "f32", name
        c = line["cpu_baseline"]
        assert c is None or (c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "oracle/lanpaint_oracle.py" in c["sample"]), name
        if line["n_gpus"] > 1:
            d = line["dist"]
            assert "per_rank" not in d and d["world_size"] == line["n_gpus"] == d["ranks_reporting"] and d["parity_ok_all_ranks"], name
            assert d["distinct_final_checksums"] == line["n_gpus"] and d["collectives_in_timed_region"] == 0, name
    default = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_c2_driver_form.json")))
    assert default["config"]["rng"] == "torch" and default["config"]["graph"] == "auto" and "drop-in" in default["config"]["engine"]
    assert default["summary"]["philox_bits_it_s"] > default["value"] and 0.9 < default["summary"]["value_over_launch_floor"] <= 1.05
    live = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_c2_live_pmc.json")))["roofline"]
    assert "live" in live["traffic_source"] and abs(live["traffic"] - 2173952) < 0.03 * 2173952       # within 3 % of the committed passes
