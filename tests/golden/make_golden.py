#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference engine
(/root/reference/src/LanPaint/lanpaint.py) on CPU fp32.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
The reference is imported, never copied.  torch.randn_like is wrapped so the
exact xi stream the reference consumed is stored next to its outputs; the
oracle and the HIP path are then fed that same stream.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)      # ROOT first: both trees have a `tests` package
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

from src.LanPaint.lanpaint import LanPaint as RefLanPaint          # noqa: E402  (reference, imported)
from src.LanPaint.earlystop import _boundary_weight, _weighted_mse  # noqa: E402

from tests import golden_cases as gc                                # noqa: E402
from tests.stubs import MODELS                                      # noqa: E402


class XiRecorder:
    def __init__(self):
        self.draws = []
        self._orig = torch.randn_like

    def __enter__(self):
        def rec(t, *a, **k):
            out = self._orig(t, *a, **k)
            self.draws.append(out.detach().clone().numpy())
            return out
        torch.randn_like = rec
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def ref_engine(case, model):
    h = case["hyper"]
    return RefLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"],
                       IS_FLUX=case.get("flux", False), IS_FLOW=case["flow"], MinStepFrac=h["MinStepFrac"])


def audio_tensors(case):
    a = case["audio"]
    shape = case["shape"]
    ai = np.zeros(shape, dtype=np.float32)
    ai[..., a["split"]:] = 1.0
    flow_a = np.asarray([a["flow_a"]], dtype=np.float32)
    abt_a = (1 - flow_a) ** 2 / ((1 - flow_a) ** 2 + flow_a ** 2)
    ve_a = flow_a / (1 - flow_a)
    corr = ((1.0 - ai) + a["corr"] * ai).astype(np.float32)
    return ai, (ve_a.astype(np.float32), abt_a.astype(np.float32), flow_a), corr


def run_case(name):
    case = gc.build_case(name)
    torch.manual_seed(1000 + len(name))
    model = MODELS[case["model"]](flow=case["flow"] or case["flux"])
    eng = ref_engine(case, model)
    x = _t(case["x"].copy())
    kw = {}
    extra = {}
    if case["audio"] is not None:
        ai, times_a, corr = audio_tensors(case)
        kw = dict(current_times_audio=tuple(_t(t) for t in times_a), audio_indicator=_t(ai), audio_correction=_t(corr))
        extra = dict(audio_indicator=ai, ve_a=times_a[0], abt_a=times_a[1], flow_a=times_a[2], audio_correction=corr)
    mo = case["model_options"]
    if mo is not None:
        mo = {k: dict(v) if isinstance(v, dict) else v for k, v in mo.items()}
        mo["lanpaint_semantic_trace"] = []
    if case["xi_seed"] is not None:
        # compact fixture: the reference's torch.randn_like is fed from a numpy seed; only the outputs are stored
        draws = iter(gc.seeded_xi(case["xi_seed"], case["shape"], 64))
        used = []
        orig = torch.randn_like

        def fed(t, *a, **k):
            d = next(draws)
            used.append(1)
            return torch.from_numpy(d).to(t.dtype)
        torch.randn_like = fed
        try:
            out = eng(x, _t(case["y"]), _t(case["noise"]), _t(case["sigma"]), _t(case["mask"]),
                      tuple(_t(t) for t in case["times"]), mo, 0, n_steps=case["n_steps"], **kw)
        finally:
            torch.randn_like = orig
        common = dict(n_draws=np.int64(len(used)), model_calls=np.int64(model.calls), xi_seed=np.int64(case["xi_seed"]),
                      shape=np.asarray(case["shape"], dtype=np.int64))
        if case["digest"]:      # the tensors are MiBs: store slice sums + sampled elements (golden_cases.digest)
            dx, do = gc.digest(x.numpy(), case["xi_seed"] + 1), gc.digest(out.numpy(), case["xi_seed"] + 2)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **common,
                                **{f"x_{k}": v for k, v in dx.items()}, **{f"out_{k}": v for k, v in do.items()})
        else:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), x_out=x.numpy(), out=out.numpy(), **common)
        return len(used), model.calls
    with XiRecorder() as rec:
        out = eng(x, _t(case["y"]), _t(case["noise"]), _t(case["sigma"]), _t(case["mask"]),
                  tuple(_t(t) for t in case["times"]), mo, 0, n_steps=case["n_steps"], **kw)
    rec_d = dict(x_in=case["x"], y=case["y"], noise=case["noise"], mask=case["mask"], sigma=case["sigma"],
                 ve=case["times"][0], abt=case["times"][1], flow_t=case["times"][2],
                 x_out=x.numpy(), out=out.numpy(), n_draws=np.int64(len(rec.draws)),
                 model_calls=np.int64(model.calls), **extra)
    for i, d in enumerate(rec.draws):
        rec_d[f"xi_{i}"] = d
    if mo is not None:
        tr = mo["lanpaint_semantic_trace"]
        rec_d["trace_dist"] = np.asarray([t["dist"] for t in tr], dtype=np.float64)
        rec_d["trace_counter"] = np.asarray([t["patience_counter"] for t in tr], dtype=np.int64)
        rec_d["trace_stopped"] = np.asarray([t["stopped"] for t in tr], dtype=np.bool_)
        # the whole per-iteration record the reference appends (earlystop.py:315-334); None -> NaN
        nan = lambda v: np.nan if v is None else float(v)      # noqa: E731
        for key in ("dist_inpaint", "dist_ring", "dist_drift", "threshold", "threshold_eff", "abt"):
            rec_d["trace_" + key] = np.asarray([nan(t[key]) for t in tr], dtype=np.float64)
        rec_d["trace_inner_step"] = np.asarray([t["inner_step"] for t in tr], dtype=np.int64)
        rec_d["trace_patience_eff"] = np.asarray([t["patience_eff"] for t in tr], dtype=np.int64)
        rec_d["trace_keys"] = np.asarray(sorted(tr[0].keys()) if tr else [])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec_d)
    return len(rec.draws), model.calls


def run_schedule(name):
    sc = gc.build_schedule(name)
    torch.manual_seed(77)
    model = MODELS[sc["model"]](flow=sc["flow"])
    eng = ref_engine(dict(hyper=sc["hyper"], flow=sc["flow"]), model)
    x = _t(sc["x"].copy())
    y, noise, mask = _t(sc["y"]), _t(sc["noise"]), _t(sc["mask"])
    sig = sc["sigmas"]
    b = sc["shape"][0]
    denoised_all, iters, traces = [], [], []
    with XiRecorder() as rec:
        for i in range(len(sig) - 1):
            s = torch.full((b,), float(sig[i]), dtype=torch.float32)
            ve, abt, ft = gc.times_from_sigma(s, sc["flow"])
            mo = None
            if sc.get("model_options") is not None:      # a fresh options dict per call, with a trace list (earlystop.py:121)
                mo = {k: dict(v) if isinstance(v, dict) else v for k, v in sc["model_options"].items()}
                mo["lanpaint_semantic_trace"] = []
            calls = model.calls
            den = eng(x, y, noise, s, mask, (ve, abt, ft), mo, 0)
            iters.append(model.calls - calls - 1)
            if mo is not None:
                traces.append(mo["lanpaint_semantic_trace"])
            denoised_all.append(den.numpy().copy())
            d = (x - den) / float(sig[i])
            x = x + d * float(sig[i + 1] - sig[i])
    rec_d = dict(x_in=sc["x"], y=sc["y"], noise=sc["noise"], mask=sc["mask"], sigmas=sig,
                 x_final=x.numpy(), denoised=np.stack(denoised_all), n_draws=np.int64(len(rec.draws)))
    for i, d in enumerate(rec.draws):
        rec_d[f"xi_{i}"] = d
    if traces:      # iterations the reference ran per sigma call and its stopper's records, call after call
        nan = lambda v: np.nan if v is None else float(v)      # noqa: E731
        flat = [t for tr in traces for t in tr]
        rec_d["iterations"] = np.asarray(iters, dtype=np.int64)
        rec_d["trace_call"] = np.asarray([k for k, tr in enumerate(traces) for _ in tr], dtype=np.int64)
        for key in ("dist", "dist_inpaint", "dist_ring", "dist_drift", "threshold_eff", "abt"):
            rec_d["trace_" + key] = np.asarray([nan(t[key]) for t in flat], dtype=np.float64)
        rec_d["trace_counter"] = np.asarray([t["patience_counter"] for t in flat], dtype=np.int64)
        rec_d["trace_stopped"] = np.asarray([t["stopped"] for t in flat], dtype=np.bool_)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec_d)
    return len(rec.draws)


def run_node_schedule(name):
    """The reference's OWN KSamplerX0Inpaint.__call__ (nodes.py:229-315; ComfyUI stubbed as its tests stub it) driving the
    reference engine over a whole schedule with the node defaults; xi from a numpy seed fed to torch.randn_like.
    Stored: the inner-step count the callable chose at every sigma (backbone calls - 1), every denoised, the final x."""
    ref = _import_ref_nodes()
    sc = gc.build_node_schedule(name)
    h, flow = sc["hyper"], sc["flow"]
    model = MODELS["linear_tuple"](flow=flow)
    model.model_type = ref.ModelType.FLOW if flow else "EPS"
    sig = sc["sigmas"]
    k = ref.KSamplerX0Inpaint(model, _t(sig))
    k.latent_image, k.noise = _t(sc["y"]), _t(sc["noise"])
    k.PaintMethod = RefLanPaint(model, h["NSteps"], h["Friction"], h["Lambda"], h["Beta"], h["StepSize"], IS_FLUX=False,
                                IS_FLOW=flow, MinStepFrac=h["MinStepFrac"])
    k.LanPaint_early_stop, k.LanPaint_min_step_frac = h["EarlyStop"], h["MinStepFrac"]
    draws = iter(gc.seeded_xi(sc["xi_seed"], sc["shape"], 256))
    used, orig = [], torch.randn_like

    def fed(t, *a, **kw):
        used.append(1)
        return torch.from_numpy(next(draws)).to(t.dtype)
    torch.randn_like = fed
    x, dm = _t(sc["x"].copy()), _t(sc["denoise_mask"])
    b = sc["shape"][0]
    n_eff, denoised = [], []
    try:
        for i in range(len(sig) - 1):
            s = torch.full((b,), float(sig[i]), dtype=torch.float32)
            calls = model.calls
            den = k(x, s, dm, model_options={}, seed=0)
            n_eff.append(model.calls - calls - 1)
            denoised.append(den.numpy().copy())
            x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
    finally:
        torch.randn_like = orig
    np.savez_compressed(os.path.join(HERE, name + ".npz"), sigmas=sig, n_eff=np.asarray(n_eff, dtype=np.int64),
                        denoised=np.stack(denoised), x_final=x.numpy(), n_draws=np.int64(len(used)),
                        model_calls=np.int64(model.calls), xi_seed=np.int64(sc["xi_seed"]),
                        shape=np.asarray(sc["shape"], dtype=np.int64))
    return n_eff, len(used)


def run_full_schedule(name):
    """A BASELINE configuration as a whole schedule through the reference engine (the bench's walk: Euler update between
    sigmas); xi from a numpy seed fed to torch.randn_like; digests of every FULL_SCHEDULE_STRIDE-th denoised + the final x."""
    sc = gc.build_full_schedule(name)
    model = MODELS["linear_tuple"](flow=sc["flow"])
    eng = ref_engine(dict(hyper=sc["hyper"], flow=sc["flow"]), model)
    x = _t(sc["x"].copy())
    extra = {}
    if sc["video_mask"]:
        # the job's mask as a workflow builds it: the reference's reshape_mask (video path) of the pixel-resolution denoise
        # mask, then KSamplerX0Inpaint's threshold + inversion (nodes.py:281-283)
        ref = _import_ref_nodes()
        dm = ref.reshape_mask(_t(gc.video_pixel_mask(sc["shape"])), sc["shape"], video_inpainting=True)
        lm = (1.0 - (dm > 0.5).float()).contiguous()
        assert tuple(lm.shape) == tuple(sc["shape"])
        sc["mask"] = lm.numpy()
        extra["mask_bits"] = np.packbits(lm.numpy().reshape(-1) > 0.5)
        extra["mask_known"] = np.int64(int(lm.sum().item()))
    y, noise, mask = _t(sc["y"]), _t(sc["noise"]), _t(sc["mask"])
    sig, b = sc["sigmas"], sc["shape"][0]
    row_scale = _t(sc["row_scale"])
    draws, used, orig = gc.seeded_xi_stream(sc["xi_seed"], sc["shape"]), [], torch.randn_like

    def fed(t, *a, **kw):
        used.append(1)
        return torch.from_numpy(next(draws)).to(t.dtype)
    torch.randn_like = fed
    rec = {}
    try:
        for i in range(len(sig)):
            s = torch.full((b,), float(sig[i]), dtype=torch.float32) * row_scale
            den = eng(x, y, noise, s, mask, gc.times_from_sigma(s, sc["flow"]), None, 0)
            if i % gc.FULL_SCHEDULE_STRIDE == 0 or i == len(sig) - 1:
                rec.update({f"den{i}_{k}": v for k, v in gc.digest(den.numpy(), sc["xi_seed"] + 10 + i).items()})
            if i + 1 < len(sig):
                x = x + (x - den) / float(sig[i]) * float(sig[i + 1] - sig[i])
    finally:
        torch.randn_like = orig
    rec.update({f"x_{k}": v for k, v in gc.digest(x.numpy(), sc["xi_seed"] + 1).items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n_draws=np.int64(len(used)), model_calls=np.int64(model.calls),
                        xi_seed=np.int64(sc["xi_seed"]), shape=np.asarray(sc["shape"], dtype=np.int64), sigmas=sig, **extra, **rec)
    return len(used), model.calls


def coefficient_kat():
    """Known answers straight from the reference's own prepare_step_size + OU closed
    form (float64 of its fp32 outputs), for the per-region coefficient table."""
    rows = []
    for flow, sig, msf in ((False, 2.0, 0.0), (False, 2.0, 1.0), (True, 0.5, 0.0), (False, 0.0292, 0.0),
                           (False, 14.6146, 0.0), (True, 0.97, 0.0), (True, 0.05, 1.0)):
        eng = RefLanPaint(None, 5, 15.0, 5.0, 1.0, 0.2, IS_FLOW=flow, MinStepFrac=msf)
        eng.img_dim_size = 4
        s = torch.tensor([sig], dtype=torch.float32)
        ve, abt, ft = gc.times_from_sigma(s, flow)
        step = eng.add_none_dims(0.2 * (1 - abt).clamp(min=msf))
        one = eng.add_none_dims(abt ** 0)
        _, abt_o, dtx, dty, _, _, a_x, a_y, d_x, d_y = eng.prepare_step_size((ve, abt, ft), step, one, one)
        rows.append([float(flow), sig, msf, float(abt), float(dtx), float(dty), float(a_x), float(a_y), float(d_x)])
    np.savez_compressed(os.path.join(HERE, "kat_step_coefficients.npz"),
                        table=np.asarray(rows, dtype=np.float64),
                        columns=np.asarray(["flow", "sigma", "min_step_frac", "abt", "dtx", "dty", "A_x", "A_y", "D"]))


def boundary_kat():
    """earlystop.py ring weight + weighted MSE on a few masks."""
    rng = np.random.default_rng(5)
    masks, rings, mses = [], [], []
    for k in range(4):
        m = (rng.random((2, 3, 9, 11)) > (0.3 + 0.15 * k)).astype(np.float32)
        a = rng.standard_normal(m.shape, dtype=np.float32)
        b = rng.standard_normal(m.shape, dtype=np.float32)
        inp = (1 - _t(m)).float()
        ring = _boundary_weight(_t(m), inp)
        masks.append(m)
        rings.append(ring.numpy())
        mses.append([_weighted_mse(_t(a), _t(b), inp), _weighted_mse(_t(a), _t(b), ring)])
    np.savez_compressed(os.path.join(HERE, "kat_boundary.npz"), masks=np.stack(masks), rings=np.stack(rings),
                        mses=np.asarray(mses, dtype=np.float64), seed=np.int64(5))


def _import_ref_nodes():
    """The reference's nodes.py with ComfyUI stubbed the way its own tests do (tests/test_reshape_mask.py:18-54)."""
    import importlib
    import types
    comfy_mod = types.ModuleType("comfy")
    comfy_mod.__path__ = []
    utils = types.ModuleType("comfy.utils")
    utils.repeat_to_batch_size = lambda t, b: t
    samplers = types.ModuleType("comfy.samplers")
    samplers.KSAMPLER = type("KSAMPLER", (), {})
    samplers.KSampler = type("KSampler", (), {"SCHEDULERS": ["karras"]})
    mb = types.ModuleType("comfy.model_base")
    mb.ModelType = types.SimpleNamespace(FLUX="FLUX", FLOW="FLOW")
    mb.WAN22 = type("WAN22", (), {})
    ver = types.ModuleType("comfyui_version")
    ver.__version__ = "0.6.0"
    comfy_mod.utils, comfy_mod.samplers, comfy_mod.model_base = utils, samplers, mb
    for name, mod in (("comfy", comfy_mod), ("comfy.utils", utils), ("comfy.samplers", samplers),
                      ("comfy.model_base", mb), ("nodes", types.ModuleType("nodes")),
                      ("latent_preview", types.ModuleType("latent_preview")), ("comfyui_version", ver)):
        sys.modules[name] = mod
    return importlib.import_module("src.LanPaint.nodes")


def blend_kat():
    """MaskBlend.blend_images / merge_video_with_mask / gaussian_kernel_2d of the reference
    (nodes.py:592-647, 1049-1088) on small random images."""
    ref = _import_ref_nodes()
    rng = np.random.default_rng(21)
    rec = {}
    for idx, (b, h, w, k) in enumerate([(2, 20, 24, 1), (1, 17, 23, 3), (2, 20, 24, 7), (1, 33, 40, 11), (1, 16, 16, 21)]):
        i1 = rng.random((b, h, w, 3), dtype=np.float32)
        i2 = rng.random((b, h, w, 3), dtype=np.float32)
        m = (rng.random((b, h, w)) > 0.8).astype(np.float32)
        out, = ref.MaskBlend().blend_images(_t(i1), _t(i2), _t(m), k)
        rec[f"blend{idx}_i1"], rec[f"blend{idx}_i2"], rec[f"blend{idx}_mask"] = i1, i2, m
        rec[f"blend{idx}_k"], rec[f"blend{idx}_out"] = np.int64(k), out.numpy()
    for idx, (f, h, w, mh, mw, k, mdim) in enumerate([(3, 20, 24, 20, 24, 5, 3), (4, 24, 32, 12, 16, 7, 3),
                                                    (2, 16, 16, 16, 16, 3, 4), (3, 18, 22, 9, 11, 9, 2)]):
        o = rng.random((f, h, w, 3), dtype=np.float32)
        p = rng.random((f + 1, h, w, 3), dtype=np.float32)
        if mdim == 2:
            m = (rng.random((mh, mw)) > 0.7).astype(np.float32)
        elif mdim == 4:
            m = (rng.random((f, 1, mh, mw)) > 0.7).astype(np.float32)
        else:
            m = (rng.random((f + 2, mh, mw)) > 0.7).astype(np.float32)
        out = ref.merge_video_with_mask(_t(o), _t(p), _t(m), k)
        rec[f"merge{idx}_orig"], rec[f"merge{idx}_inp"], rec[f"merge{idx}_mask"] = o, p, m
        rec[f"merge{idx}_k"], rec[f"merge{idx}_out"] = np.int64(k), out.numpy()
    for k in (1, 3, 7, 51):
        rec[f"gauss{k}"] = ref.gaussian_kernel_2d(k).numpy()
    np.savez_compressed(os.path.join(HERE, "kat_mask_blend.npz"), **rec)


def main():
    only = sys.argv[1:]                   # `make_golden.py name ...` regenerates just those cases
    for name in gc.NODE_SCHEDULES:
        if name in only:
            n_eff, nd = run_node_schedule(name)
            print(f"{name:24s} draws={nd} n_eff={n_eff}")
    for name in gc.FULL_SCHEDULES:
        if name in only:
            print(f"{name:24s} draws, model calls = {run_full_schedule(name)}")
    for name in gc.SCHEDULES:
        if name in only:
            print(f"{name:24s} draws={run_schedule(name)}")
    for name in gc.CASES:
        if only and name not in only:
            continue
        nd, mc = run_case(name)
        print(f"{name:24s} draws={nd:3d} model_calls={mc}")
    if only:
        return
    for name in gc.SCHEDULES:
        print(f"{name:24s} draws={run_schedule(name)}")
    for name in gc.NODE_SCHEDULES:
        n_eff, nd = run_node_schedule(name)
        print(f"{name:24s} draws={nd} n_eff={n_eff}")
    for name in gc.FULL_SCHEDULES:
        print(f"{name:24s} draws, model calls = {run_full_schedule(name)}")
    coefficient_kat()
    boundary_kat()
    blend_kat()


if __name__ == "__main__":
    main()
