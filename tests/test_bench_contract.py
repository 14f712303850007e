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
    import bench
    from oracle import ref_engine
    out = bench.cpu_baseline("c1_sd15", 0.6)
    json.dumps(out)
    staged = ref_engine.load_reference() is not None
    assert out["kind"] == ("reference" if staged else "port") and out["unit"] == "think-iterations/s" and out["value"] > 0
    n_all = bench._usable_cpus()
    assert out["cpu_model"] and out["usable_cpus"] == n_all and "1" in out["threads"]
    assert str(n_all) in (set(out["threads"]) | set(out["threads_not_sampled"] or {})) and int(out["cores"]) in map(int, out["threads"])
    assert all(1 <= v["passes"] <= 5 and v["sigma_calls_per_pass"] >= 1 for v in out["threads"].values())   # median of <= 5, bounded sample
    assert out["leg_seconds"] < 20.0
    if staged:
        assert "oracle/_ref" in out["sample"] and len(out["reference_source_sha256"]) == 3
        assert out.get("port_over_reference") is None or 0.4 < out["port_over_reference"] < 2.5     # (timed when the budget allows)
    else:
        assert "oracle/lanpaint_oracle.py" in out["sample"]


def test_cpu_baseline_falls_back_to_the_port_without_the_staged_reference(monkeypatch):
    import bench
    from oracle import ref_engine
    monkeypatch.setattr(ref_engine, "load_reference", lambda: None)
    out = bench.cpu_baseline("c1_sd15", 0.4)
    assert out["kind"] == "port" and "oracle/lanpaint_oracle.py" in out["sample"] and out["value"] > 0


def test_staged_reference_is_the_reference():
    """oracle/_ref (bytecode compiled from /root/reference by oracle/build_ref.py) IS the unmodified engine: where the
    reference tree is present the staged sources' hashes equal the tree's, and the staged engine reproduces a golden fixture
    bit for bit (the fixtures were written by importing that same file)."""
    import hashlib
    import numpy as np
    import pytest
    import torch
    from oracle import ref_engine
    from tests import golden_cases as gc
    from tests.helpers import load_golden, xi_list
    from tests.stubs import MODELS
    cls = ref_engine.load_reference()
    if cls is None:
        pytest.skip("oracle/_ref not staged in this checkout")
    m = ref_engine.manifest()
    for name, rec in m["modules"].items():
        src = os.path.join("/root/reference/src/LanPaint", name + ".py")
        if os.path.exists(src):
            assert hashlib.sha256(open(src, "rb").read()).hexdigest() == rec["source_sha256"], name
    case, g = gc.build_case("ve_basic"), load_golden("ve_basic")
    draws = iter(xi_list(g))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731
    orig = torch.randn_like
    torch.randn_like = lambda like, *a, **k: t(next(draws)).to(like.dtype)
    try:
        h = case["hyper"]
        eng = cls(MODELS[case["model"]](), h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLUX=False,
                  IS_FLOW=False, MinStepFrac=h["MinStepFrac"])
        x = t(case["x"].copy())
        out = eng(x, t(case["y"]), t(case["noise"]), t(case["sigma"]), t(case["mask"]), tuple(t(v) for v in case["times"]), None, 0)
    finally:
        torch.randn_like = orig
    assert np.array_equal(x.numpy(), g["x_out"]) and np.array_equal(out.numpy(), g["out"])


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
