"""The think loops off the fast path, and the public single-iteration entry.

The fast path (engine.py) is one fused launch per iteration with nobody watching.  Here: the loop whose stop rule runs on the
device (`_loop_es`: watched eager launches / device-gated captured launches), the loop with a host-side stopper (`_loop_unfused`:
a user distance_fn, a batch sharded over ranks), the reference-shaped loop for subclasses that override langevin_dynamics /
score_model / prepare_step_size (`_loop_compat`), and `langevin_dynamics` itself with the reference's signature."""
from __future__ import annotations

import ctypes
from functools import partial

import torch

from . import _cabi
from ._cabi import (LP_FL_ES, LP_FL_ES_GATED, LP_FL_ES_CLOSE, LP_FL_CFG_FUSED, LP_FL_FLOW, LP_FL_MASK_BITS, LP_FL_MASK_U8, LP_FL_XIN_BF16,
                    LP_FL_XIN_F16, LP_FL_PER_ELEMENT, LP_FL_WRITE_X0S, LP_FL_X0_BF16, LP_FL_X0_F16, LP_FL_X0S_GIVEN, LP_PH_EMIT,
                    LP_PH_POST_FIRST, LP_PH_POST_STEADY, LP_PH_PRE_HALF, LP_PH_REPLACE, LP_REPLACE_FLOW, LP_REPLACE_KNOWN,
                    LP_REPLACE_VE)
from ._util import _as_f32c, _state_x0
from .buffers import _DeviceStop
from .types import LangevinState


class ThinkLoops:
    """Mixin of LanPaint (lanpaint.py)."""

    def _device_stop(self, like, n_steps):
        if self._ds is None or not self._ds.matches(like, n_steps):
            self._ds = _DeviceStop(like, n_steps)
        return self._ds

    def _es_trace(self, es, ds, i):
        """Append the reference's trace record of iteration i (earlystop.py:315-334) from the mailbox."""
        trace = es["trace"]
        if trace is None or ds.f64[3] == 0.0:
            return
        rec = ds.f64[_cabi.LP_ES_TRACE0 + 8 * i: _cabi.LP_ES_TRACE0 + 8 * i + 8]
        thr_eff, opt = float(ds.f64[4]), (lambda v: None if v != v else float(v))
        trace.append({"case_id": es["tags"][0], "outer_step": es["tags"][1], "bench_timestep": es["tags"][2],
                      "inner_step": i + 1, "dist": float(rec[0]), "dist_inpaint": opt(rec[1]), "dist_ring": opt(rec[2]),
                      "dist_drift": opt(rec[3]), "threshold": thr_eff, "threshold_eff": thr_eff,
                      "patience_counter": int(rec[4]), "patience_eff": int(es["patience_eff"]), "abt": float(ds.f64[5]),
                      "custom_dist": False, "stopped": bool(rec[5])})

    def _es_resolve(self):
        """A replayed early-stop loop reports how far it ran: wait for its "done" word, account the iterations, hand
        over the trace records, and put torch's generator where the reference leaves it after that many iterations."""
        p = self._es_pending
        if p is None:
            return
        self._es_pending = None
        ds, seq, n_steps, es, dev, inc = p
        ds.wait(seq + _cabi.LP_ES_SEQ_DONE, dev)
        n_ran = int(ds.f64[1])
        total = int(ds.f64[6])             # the device counts across calls: every replay since the last collection
        self._iterations_run += total - ds.seen_total
        ds.seen_total = total
        self.last_inner_steps = n_ran
        for i in range(n_ran):
            self._es_trace(es, ds, i)
        if inc and n_ran < n_steps:      # the launches past the stop drew nothing the reference would have drawn
            gen = self._generator(dev)
            back = 2 * (n_steps - n_ran) * inc
            gen.set_offset(gen.get_offset() - back)
            self._torch_consumed -= back

    # ------------------------------------------------------------------ loops off the fast path
    def _x0s_buffer(self, ws, exclude):
        """A rotating x0s buffer not aliased by any tensor in `exclude`."""
        ptrs = {t.data_ptr() for t in exclude if t is not None}
        for buf in ws.x0s:
            if buf.data_ptr() not in ptrs:
                return buf
        buf = torch.empty_like(ws.x_t)
        ws.x0s.append(buf)
        return buf

    def _loop_es(self, st, n_steps, model_options, seed):
        """Inner early stop with the default metric, evaluated on the device (LP_FL_ES): the POST launch of every
        iteration also reduces the weighted MSEs of earlystop.py:279-306 and applies the threshold / patience /
        drift-anchor rule in its last block.
        Eager: the host reads the verdict from the pinned mailbox once per iteration and leaves the loop like the
        reference does (no backbone call is wasted); the PRE half-step of the next iteration is a launch of its own.
        Captured (hipGraph): nobody watches -- the launches are gated on the device-side flag (LP_FL_ES_GATED), keep
        the fused one-launch-per-iteration shape, and the backbone calls after the stop still run (their results are
        ignored).  Returns the iterations run (0 while capturing: the replay reports it, _es_resolve)."""
        d, ws, shape, base_flags, stream = self._desc, st.ws, st.shape, st.base_flags, st.stream
        es, ds = st.es, st.es["ds"]
        d.es_n_steps = n_steps
        gated = self._capturing is not None
        ran = 0
        for i in range(n_steps):
            last = i == n_steps - 1
            if i > 0 and not gated:                   # first half-step of iteration i, committed (lanpaint.py:280)
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
            output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
            if gated:
                close = LP_FL_ES_CLOSE if (last and self._es_close) else 0
                alive = self._set_model_output(d, output, base_flags | LP_FL_ES | LP_FL_ES_GATED | es["ring_flag"] | close | self._emit(st, last), shape)
                d.phases = (LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY) | (0 if last else LP_PH_PRE_HALF) | LP_PH_EMIT
                self._set_xi(d, ws.x_t, want_pre=not last)
            else:
                alive = self._set_model_output(d, output, base_flags | LP_FL_ES | es["ring_flag"], shape)
                d.phases = LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY
                self._set_xi(d, ws.x_t, want_pre=False)
            d.es_index = i
            self._launch_step(stream)
            del alive
            if gated:
                continue
            ds.wait(es["seq"] + i + 1, st.xc.device)
            ran += 1
            self._es_trace(es, ds, i)
            if ds.f64[2] != 0.0:                      # stopped
                break
            if i == 0 and ds.f64[3] == 0.0 and not last:
                # The stopper can never fire in this call (threshold_eff <= 0 at this abt, or nothing to inpaint:
                # earlystop.py:111-117 -- the reference's from_options returns None and runs its plain loop).  Do the same:
                # no more verdicts to wait for, no early-stop streams, one fused launch per iteration from here on.
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
                for j in range(1, n_steps):
                    last_j = j == n_steps - 1
                    output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
                    alive = self._set_model_output(d, output, base_flags | self._emit(st, last_j), shape)
                    d.phases = LP_PH_POST_STEADY | (0 if last_j else LP_PH_PRE_HALF) | LP_PH_EMIT
                    self._set_xi(d, ws.x_t, want_pre=not last_j)
                    self._launch_step(stream)
                    del alive
                ds.seen_total += 1                # (the device counted iteration 0 only)
                return n_steps
        if not gated:
            ds.seen_total += ran                  # this loop's iterations are accounted by its caller
            d.phases, d.flags = LP_PH_EMIT, base_flags | self._emit(st, True)
            self._launch_step(stream)
        return ran

    def _loop_unfused(self, st, n_steps, model_options, seed, stopper):
        """Early stop enabled: the stopper decides after every iteration, so the POST
        half of iteration i cannot be fused with the PRE half of iteration i+1."""
        d, ws, shape, base_flags, stream = self._desc, st.ws, st.shape, st.base_flags, st.stream
        args = None
        ran = 0
        for i in range(n_steps):
            x_t_before = ws.x_t.clone() if args is None or stopper.has_custom_distance_fn else None
            if i > 0:
                d.phases, d.flags = LP_PH_PRE_HALF | LP_PH_EMIT, base_flags | self._emit(st, False)
                self._set_xi(d, ws.x_t, want_pre=True, want_post=False)
                self._launch_step(stream)
            output = self.inner_model(st.x_in, st.t_model, model_options=model_options, seed=seed)
            x0s = self._x0s_buffer(ws, [args.x0 if args else None, stopper.anchor])
            alive = self._set_model_output(d, output, base_flags | LP_FL_WRITE_X0S, shape)
            d.x0s = x0s.data_ptr()
            d.phases = LP_PH_POST_FIRST if i == 0 else LP_PH_POST_STEADY
            self._set_xi(d, ws.x_t, want_pre=False)
            self._launch_step(stream)
            del alive
            prev_args, args = args, LangevinState(None, ws.C, x0s)
            ran += 1
            ctx = {"step": i, "steps_done": i + 1, "n_steps": n_steps, "mask": st.m, "latent_image": st.y,
                   "current_times": st.current_times, "seed": seed}
            if stopper.observe(i, x_before=x_t_before, x_after=ws.x_t,
                               x_prev_for_user=x_t_before if stopper.has_custom_distance_fn else None,
                               x0_prev=prev_args.x0 if prev_args is not None else None, x0_cur=args.x0, ctx=ctx):
                break
        d.x0s = None
        d.phases, d.flags = LP_PH_EMIT, base_flags | self._emit(st, True)
        self._launch_step(stream)
        return ran

    def _loop_compat(self, ws, shape, m, y, abt, current_times, n_steps, model_options, seed, stopper):
        """A subclass / instance overrides langevin_dynamics, score_model or
        prepare_step_size: run the reference's per-iteration call structure
        (lanpaint.py:113-142) so the overrides see the calls they expect."""
        abt_b = self.add_none_dims(abt)
        step_size = self.add_none_dims(self.step_size * (1 - abt).clamp(min=self.min_step_frac))
        x_t, args, ran = ws.x_t, None, 0
        for i in range(n_steps):
            score_func = partial(self.score_model, y=y, mask=m, abt=abt_b, sigma=self.add_none_dims(current_times[0]),
                                 tflow=self.add_none_dims(current_times[2]), model_options=model_options, seed=seed)
            prev_args = args
            x_prev = x_t.detach().clone() if (stopper is not None and stopper.has_custom_distance_fn) else None
            x_before = x_t.detach().clone() if stopper is not None else None
            x_t, args = self.langevin_dynamics(x_t, score_func, m, step_size, current_times,
                                               sigma_x=self.add_none_dims(self.sigma_x(abt)),
                                               sigma_y=self.add_none_dims(self.sigma_y(abt)), args=args)
            ran += 1
            if stopper is not None:
                ctx = {"step": i, "steps_done": i + 1, "n_steps": n_steps, "mask": m, "latent_image": y,
                       "current_times": current_times, "seed": seed}
                if stopper.observe(i, x_before=x_before, x_after=x_t, x_prev_for_user=x_prev, x0_prev=_state_x0(prev_args),
                                   x0_cur=_state_x0(args), ctx=ctx):
                    break
        if x_t.data_ptr() != ws.x_t.data_ptr():
            ws.x_t.copy_(x_t)
        return ran

    def langevin_dynamics(self, x_t, score, mask, step_size, current_times, sigma_x=1, sigma_y=0, args=None):
        """Public single-iteration entry with the reference signature (lanpaint.py:192-293):
        `score` is any callable x_t -> score tensor.  Returns (x_t_new, LangevinState)."""
        if args is not None and not isinstance(args, LangevinState) and isinstance(args, tuple):
            if len(args) == 2:
                args = LangevinState(args[0], args[1], None)
            elif len(args) >= 3:
                args = LangevinState(args[0], args[1], args[2])
        if not x_t.is_cuda:
            raise RuntimeError("lanpaint_amd.LanPaint.langevin_dynamics runs on a HIP device only; no CPU fallback")
        if self.img_dim_size is None:
            self.img_dim_size = x_t.ndim
        step_sizes = self.prepare_step_size(current_times, step_size, sigma_x, sigma_y)
        _sig, abt_b, dtx = step_sizes[0], step_sizes[1], step_sizes[2]
        if torch.mean(dtx) <= 0.0:                                   # lanpaint.py:205
            return x_t, args
        lib = self._lib
        shape, rows = x_t.shape, x_t.shape[0]
        xt = _as_f32c(x_t).clone()
        mk = _as_f32c(mask if mask.shape == shape else mask.expand(shape))
        stream = self._stream(xt.device)
        flow = bool(self.IS_FLUX or self.IS_FLOW)
        sx = torch.as_tensor(sigma_x, dtype=torch.float32, device=xt.device)
        sy = torch.as_tensor(sigma_y, dtype=torch.float32, device=xt.device)
        step_t = torch.as_tensor(step_size, dtype=torch.float32, device=xt.device)
        if sx.numel() > 1 and bool((sx != sx.reshape(-1)[0]).any()) or sy.numel() > 1 and bool((sy != sy.reshape(-1)[0]).any()):
            raise NotImplementedError("non-uniform sigma_x / sigma_y are not supported by the HIP path")
        sx0, sy0 = float(sx.reshape(-1)[0]), float(sy.reshape(-1)[0])
        VE_Sigma, abt, _ft = current_times
        d = _cabi.LpStepDesc()
        hyp = _cabi.LpHyper()
        hyp.lambda_, hyp.step_size, hyp.min_step_frac = float(self.chara_lamb), float(self.step_size), 0.0
        hyp.beta = (sy0 / sx0) if sx0 != 0.0 else 0.0
        hyp.is_flow, hyp.one_plus_lambda = int(flow), 1.0 + float(self.chara_lamb)
        base = (LP_FL_FLOW if flow else 0) | LP_FL_X0S_GIVEN | LP_FL_WRITE_X0S
        d.n_el, d.el_per_row, d.rows = xt.numel(), xt.numel() // rows, rows
        d.lambda_, d.one_plus_lambda, d.beta = hyp.lambda_, hyp.one_plus_lambda, hyp.beta
        d.noise_scale = 1.0
        d.mask, d.x_t = mk.data_ptr(), xt.data_ptr()
        if abt.numel() not in (1, rows) or step_t.numel() not in (1, rows):
            # per-element times (AV packs): the kernel derives the step from abt itself,
            # StepSize*max(1-abt, MinStepFrac) -- what the engine passes as `step_size` here
            base |= LP_FL_PER_ELEMENT
            abt_el = _as_f32c(self.add_none_dims(abt).expand(shape))
            ve_el = _as_f32c(self.add_none_dims(VE_Sigma).expand(shape))
            d.abt_el, d.ve_el, d.coef = abt_el.data_ptr(), ve_el.data_ptr(), None
            d.step_size, d.min_step_frac = float(self.step_size) * sx0, float(self.min_step_frac)
        else:
            coef = torch.empty((rows, _cabi.LP_COEF_STRIDE), dtype=torch.float32, device=xt.device)
            ve_r, abt_r = _as_f32c(VE_Sigma.reshape(-1)), _as_f32c(abt.reshape(-1))
            step_r = _as_f32c((step_t * sx0).reshape(-1))
            _cabi.check(lib.lp_coeffs(ctypes.byref(hyp), ve_r.data_ptr(), int(ve_r.numel() > 1), abt_r.data_ptr(),
                                      int(abt_r.numel() > 1), None, 0, step_r.data_ptr(), int(step_r.numel() > 1), None, 0,
                                      rows, coef.data_ptr(), stream), "lp_coeffs")
            d.step_size, d.min_step_frac = hyp.step_size, 0.0
            d.coef = coef.data_ptr()
        d.rng_seed = int(self.philox_seed or 0)
        if args is None:
            c_buf = torch.empty_like(xt)
            d.C = c_buf.data_ptr()
        else:
            c_buf = _as_f32c(args.C).clone()
            d.C = c_buf.data_ptr()
            d.phases, d.flags = LP_PH_PRE_HALF, base                # first half-step with the old C
            self._set_xi(d, xt, want_pre=True, want_post=False)
            self._launch_step_desc(d, stream)
        x0s_in = _as_f32c(xt + score(xt))                            # Coef_C: x0 = x_t + score(x_t)
        x0s_out = torch.empty_like(xt)
        d.x0, d.x0_big, d.x0s = x0s_in.data_ptr(), x0s_in.data_ptr(), x0s_out.data_ptr()
        d.phases, d.flags = (LP_PH_POST_FIRST if args is None else LP_PH_POST_STEADY), base
        self._set_xi(d, xt, want_pre=False)
        self._launch_step_desc(d, stream)
        return xt.to(x_t.dtype), LangevinState(None, c_buf, x0s_out)

    def _launch_step_desc(self, d, stream):
        _cabi.check(self._lib.lp_step(ctypes.byref(d), stream), "lp_step")
