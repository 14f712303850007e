"""hipGraph capture and replay of whole sigma calls: the state machine between "eager launches" and "one hipGraphLaunch per call".

A sigma call (replace launch, N x [backbone, fused step], final backbone call, finalise) is launch-bound at image-latent sizes; it
is captured once per (shape, inner-step count, tensors, options) and replayed with only the caller's pointers patched.  This
module holds everything that decides WHETHER a call may replay (`_same_call`, `_graph_eligible`, graph="auto"'s verification
against eager launches), HOW it is captured (`_capture`: warm-up, generator bookkeeping, the replace launch as node 0, the
finalise fed through a device-side I/O table) and the replay paths (`_replay_fast`, `begin_call` / `finish_call`, `node_call`).
The arithmetic and the launch descriptors themselves live in engine.py; walked as a state machine by
tests/test_gpu_state_machine.py."""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _cabi
from ._util import _gc_hold, _gc_release
from .buffers import _CapturedCall, _Workspace
from .masks import _compact_mask


class GraphReplay:
    """Mixin of LanPaint (lanpaint.py): capture / replay.  Uses the engine core's `_prologue`, `_think_and_final_model`,
    `_epilogue`, `_fill_final_desc`, descriptors and generator helpers."""

    MAX_GRAPHS = 16          # captured sigma calls kept per engine (one per distinct n_steps / tensor set)
    AUTO_MAX_BACKBONE_HOST_US = 100.0   # graph="auto": only loops whose backbone call costs the host less than this are captured

    # ------------------------------------------------------------------ split-phase call (KSamplerX0Inpaint)
    def begin_call(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed):
        """First half of `__call__` for a caller that does not know `n_steps` yet: enqueue what does not depend on it
        -- the replace step, the VP rescale, the coefficient table, the I/O table -- and return a token for
        `finish_call(token, n_steps)`; None when the call is not a steady-state graph replay (then use `__call__`).
        KSamplerX0Inpaint needs the device to tell it sigma's position in the schedule (nodes.py:286-299) before it
        can fix the inner-step count; with the head of the call already queued the GPU goes on working while the host
        picks the graph."""
        cap = self._last_cap
        if cap is None or cap.tail is None or self.model_dtype is not None or not x.is_cuda:
            return None
        self.img_dim_size = len(x.shape)
        self.latent_image, self.noise = latent_image, noise
        self.audio_indicator = self.current_times_audio = self.audio_correction = None
        self._noise_regenerated = self._noise_is_zero(noise)
        if not self._same_call(cap, x, sigma, latent_mask, current_times, cap.ident[4], model_options, seed):
            return None
        lib, stream = self._lib, self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        out = torch.empty_like(x)
        k0 = cap.k0_desc
        k0.x, k0.noise = x.data_ptr(), noise.data_ptr()
        k0.t_ve, k0.t_abt, k0.t_rsig = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr()
        k0.t_model = (current_times[2] if cap.flow else ve).data_ptr()
        k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        off = 0
        if self.rng == "torch":        # publish the generator state; finish_call advances it by what its graph draws
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
        _cabi.check(lib.lp_step(ctypes.byref(k0), stream), "lp_step")
        return (cap, x, out, sigma, latent_mask, current_times, model_options, seed, off, stream)

    def finish_call(self, token, n_steps):
        """Second half: replay the think loop + final backbone call + finalise captured for `n_steps`."""
        cap0, x, out, sigma, latent_mask, current_times, model_options, seed, off, stream = token
        if n_steps is None:
            n_steps = self.n_steps
        cap = cap0
        if cap0.ident[4] != n_steps:
            cap = cap0.siblings.get(n_steps)
            if cap is None or not cap.alive:
                cap = self._graphs.get(cap0.key[:2] + (int(n_steps),) + cap0.key[3:])
                if cap is None or cap.tail is None or cap.model_options is not model_options or cap.ws is not cap0.ws:
                    # no capture for this count yet: the ordinary path captures it (and re-enqueues the replace step,
                    # which reads the same untouched x and publishes the same generator state)
                    return self(x, self.latent_image, self.noise, sigma, latent_mask, current_times, model_options, seed,
                                n_steps=n_steps)
                cap0.siblings[n_steps] = cap
                self._graphs.move_to_end(cap.key)
        if self.rng == "torch" and cap.launches:
            self._generator(x.device).set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        _cabi.check(self._lib.lp_replay_call(ctypes.byref(cap.tail), stream), "lp_replay_call")
        return out

    def node_call(self, x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, nd):
        """The steady state of KSamplerX0Inpaint.__call__ in ONE trip through the FFI (lp_node_call): sigma -> times and
        the two scalars of the inner-step rule, the replace launch of this call, the wait for the scalars, the rule
        (nodes.py:286-299), and the launch of the graph captured for the resulting count.  `nd` is the caller's
        LpNodeCallDesc with the sigma / schedule / mailbox / rule fields filled in.  Returns (out, n_eff), or None when the
        call is not a steady-state replay (nothing enqueued: the caller takes its ordinary path)."""
        cap0 = self._last_cap
        if cap0 is None or cap0.tail is None or self.model_dtype is not None or not x.is_cuda:
            return None
        self.img_dim_size = len(x.shape)
        self.latent_image, self.noise = latent_image, noise
        self.audio_indicator = self.current_times_audio = self.audio_correction = None
        self._noise_regenerated = self._noise_is_zero(noise)
        if not self._same_call(cap0, x, sigma, latent_mask, current_times, cap0.ident[4], model_options, seed):
            return None
        table = cap0.node_table
        if table is None or table[3] != len(self._graphs) or table[4] is not model_options:
            nd.valid_word = self._rng_state(x.device).data_ptr() + 32
            table = self._node_table(cap0, nd.n_steps, model_options, nd)
        stream = self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        out = torch.empty_like(x)
        k0 = cap0.k0_desc
        k0.x, k0.noise = x.data_ptr(), noise.data_ptr()
        k0.t_ve, k0.t_abt, k0.t_rsig = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr()
        k0.t_model = (current_times[2] if cap0.flow else ve).data_ptr()
        k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        off = 0
        if self.rng == "torch":
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
        if getattr(nd, "_lp_table", None) is not table:     # (the table changes when a new count has been captured)
            nd._lp_table = table
            nd.replace, nd.exec_by_count, nd.n_counts = table[2], table[0], len(table[1])
            nd.valid_word = self._rng_state(x.device).data_ptr() + 32
            nd.full_exec_by_count, nd.full_binding_by_count = table[5], table[6]
        rc = self._lib.lp_node_call(ctypes.byref(nd), stream)
        if rc != _cabi.LP_OK:
            # A failed call may have left a speculated, self-voided run in the queue.  The library tries to restore the word
            # the captured lp_finalize checks; do not rely on it: restore it from here as well (an ordinary torch write in
            # stream order) and forget every capture of this engine, so nothing replays against half-published state.
            try:
                self._rng_state(x.device)[4] = 1
            except Exception:
                pass
            self._forget_captures()
            _cabi.check(rc, "lp_node_call")
        n_eff = nd.n_eff
        if not nd.launched:
            # no capture for this count yet: the ordinary path captures it (and re-enqueues the replace step, which reads
            # the same untouched x and publishes the same generator state)
            cap0.node_table = None
            return self(x, latent_image, noise, sigma, latent_mask, current_times, model_options, seed, n_steps=n_eff), n_eff
        cap = table[1][n_eff]
        if self.rng == "torch" and cap.launches:
            self._generator(x.device).set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        return out, n_eff

    def _forget_captures(self):
        """Drop every captured sigma call (after a failed native call: the next call takes the full path again)."""
        self._last_cap = None
        for cap in self._graphs.values():
            cap.alive = False
            cap.node_table = None
            cap.siblings = {}
        self._graphs.clear()

    def _node_table(self, cap0, n_max, model_options, nd=None):
        """hipGraphExec_t of the tail graph captured for every inner-step count 0 .. n_max of this call shape (NULL where
        none exists yet), as the array lp_node_call indexes; holds the captures alive.  With the caller's node descriptor
        `nd`, also the whole-call graphs whose first node is the replace launch with the sigma algebra folded in
        (`_sigma_root`): a speculated call is then one hipGraphLaunch."""
        caps = []
        for n in range(int(n_max) + 1):
            cap = cap0 if cap0.ident[4] == n else cap0.siblings.get(n)
            if cap is None or not cap.alive:
                cap = self._graphs.get(cap0.key[:2] + (n,) + cap0.key[3:])
                if cap is not None and (cap.tail is None or cap.model_options is not model_options or cap.ws is not cap0.ws):
                    cap = None
                if cap is not None:
                    cap0.siblings[n] = cap
            caps.append(cap if (cap is not None and cap.tail is not None) else None)
        arr = (ctypes.c_void_p * len(caps))(*[(c.tail.graph_exec if c is not None else None) for c in caps])
        full = full_b = None
        if nd is not None and os.environ.get("LANPAINT_AMD_NODE_ONE_LAUNCH", "1") != "0":
            roots = [self._sigma_root(c, cap0, nd) if c is not None else None for c in caps]
            if any(r is not None for r in roots):
                full = (ctypes.c_void_p * len(caps))(*[(r[1] if r is not None else None) for r in roots])
                full_b = (ctypes.POINTER(_cabi.LpGraphBinding) * len(caps))(
                    *[(ctypes.pointer(r[2]) if r is not None else ctypes.POINTER(_cabi.LpGraphBinding)()) for r in roots])
        cap0.node_table = (arr, caps, ctypes.pointer(cap0.k0_desc), len(self._graphs), model_options, full, full_b)
        return cap0.node_table

    def _sigma_root(self, cap, cap0, nd):
        """The copy of `cap`'s graph whose node 0 is the replace launch WITH the sigma algebra (LP_PH_SIGMA), made once per
        capture, or None where the call does not qualify (lp_node_call's own conditions for folding the algebra: bit-packed
        mask, a fused replace form, no correction tensor, no early-stop reset) or the runtime refuses."""
        if cap.sigma_root is not None:
            return cap.sigma_root or None
        cap.sigma_root = False
        k0 = cap0.k0_desc
        ok = (cap.binding is not None and k0 is not None
              and k0.phases == (_cabi.LP_PH_REPLACE | _cabi.LP_PH_EMIT | _cabi.LP_PH_COEFFS) and (k0.flags & _cabi.LP_FL_MASK_BITS)
              and not k0.corr_el and not k0.es_reset and k0.replace_kind != _cabi.LP_REPLACE_KNOWN
              and bool(nd.is_flow) == bool(k0.flags & _cabi.LP_FL_FLOW) and nd.rows == k0.rows and nd.fold_sigma)
        if not ok:
            return None
        try:
            raw_graph = int(cap.graph.raw_cuda_graph())
        except Exception:
            return None
        d = _cabi.LpStepDesc.from_buffer_copy(k0)
        d.phases = k0.phases | _cabi.LP_PH_SIGMA
        d.io_valid = 0
        d.sg_sigma, d.sg_schedule, d.sg_schedule_len, d.sg_times_out = nd.sigma, nd.schedule, nd.schedule_len, nd.times_out
        d.sg_scalars_out, d.sg_seq_out, d.sg_seq, d.sg_valid_out = nd.scalars_out, nd.seq_out, 0, nd.valid_word
        d.sg_n_steps, d.sg_early_stop, d.sg_total_steps, d.sg_guess = nd.n_steps, nd.early_stop, nd.total_steps, 0
        d.sg_min_step_frac = nd.min_step_frac
        g, e, b = ctypes.c_void_p(), ctypes.c_void_p(), _cabi.LpGraphBinding()
        if self._lib.lp_graph_clone_sigma_root(raw_graph, ctypes.byref(d), ctypes.byref(g), ctypes.byref(e), ctypes.byref(b)) != _cabi.LP_OK:
            return None
        cap.sigma_root = (g.value, e.value, b)
        return cap.sigma_root

    # ------------------------------------------------------------------ hipGraph replay of one sigma call
    def _same_call(self, cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed):
        """Identity pre-check of the steady state (a sampler calls the engine once per sigma with the same
        latent_image / mask / options objects): everything `_graph_eligible` and the graph key look at, without
        building the key.  Any miss falls through to the full path."""
        i = cap.ident
        y, nz = self.latent_image, self.noise
        ve, abt, ft = current_times
        f32 = torch.float32
        return (i[0] is y and i[1] is latent_mask and i[2] is model_options and i[3] == x.shape and i[4] == n_steps
                and i[5] == seed and i[6] is self.rng and self.graph and not self._noise_regenerated
                and i[7] == y.data_ptr() and i[8] == latent_mask.data_ptr()
                and i[9] is getattr(latent_mask, "_lp_bits", None) and i[10] is getattr(latent_mask, "_lp_u8", None)
                and i[11] == x.device and i[12] == sigma.numel()
                and self._times_ok(cap, current_times)
                and self.audio_indicator is None and self.audio_correction is None
                and not (self.early_stop_threshold > 0.0 and self.early_stop_patience > 0)
                and x.dtype == f32 and sigma.dtype == f32
                and nz.dtype == f32 and nz.shape == x.shape and x.is_contiguous() and sigma.is_contiguous()
                and nz.is_contiguous()
                and x.device.index == torch.cuda.current_device() and i[16] == self._override_state()
                and i[17] == self._hyper_key() and (x.data_ptr() & 15) == 0 and (nz.data_ptr() & 15) == 0
                and not (isinstance(model_options, dict) and "lanpaint_semantic_stop" in model_options))

    @staticmethod
    def _times_ok(cap, current_times):
        """The time tensors fit the capture (size, fp32, dense).  A caller that hands the same tensor objects call after
        call -- KSamplerX0Inpaint alternates between two sets -- is only checked once per set."""
        i, f32 = cap.ident, torch.float32
        ve, abt, ft = current_times
        for seen in cap.times_seen:                     # identity, never tensor ==
            if seen[0] is ve and seen[1] is abt and seen[2] is ft:
                return True
        ok = (i[13] == ve.numel() and i[14] == abt.numel() and i[15] == ft.numel() and ve.dtype == f32 and abt.dtype == f32
              and ft.dtype == f32 and ve.is_contiguous() and abt.is_contiguous() and ft.is_contiguous())
        if ok:
            cap.times_seen = (cap.times_seen + ((ve, abt, ft),))[-2:]
        return ok

    def _hyper_key(self):
        """The public hyper-parameters a captured launch bakes in (the reference reads them on every call)."""
        return (self.chara_lamb, self.chara_beta, self.step_size, self.min_step_frac)

    def _override_state(self):
        """Which of the three overridable methods are not this module's own (on the instance or on its class -- compared with
        the functions as DEFINED here, so patching the base class itself counts too), plus the model-type switches."""
        d, t, o = self.__dict__, type(self), self._OWN_METHODS
        return ("langevin_dynamics" in d or t.langevin_dynamics is not o[0], "score_model" in d or t.score_model is not o[1],
                "prepare_step_size" in d or t.prepare_step_size is not o[2], self.IS_FLUX, self.IS_FLOW, self.model_dtype)

    def _graph_eligible(self, x, model_options, sigma, current_times):
        if not self.graph or self._graph_blocked or callable(self.rng) or self._noise_regenerated:
            return False         # (regenerated noise is a fresh tensor per call: nothing stable to bake into a graph)
        rows = x.shape[0] if x.ndim else 1
        if any(t.numel() not in (1, rows) for t in (sigma, *current_times)):
            return False         # per-element times: the general path, eager only
        if self.audio_indicator is not None or self.audio_correction is not None:
            # AV packs replay only on the two-row table (LP_FL_AV: per-row time pairs + a 0/1 indicator), whose per-call inputs --
            # the interleaved times, the correction tensor -- live in workspace buffers the prologue refreshes; the reference-
            # shaped per-element form builds fresh full-size tensors per call.  (A gated early stop rides along since round 5: a
            # stopped launch re-emits every element with its own stream's scale; rows with different audio shares keep the host
            # stopper -- the device-side threshold takes ONE share, see __call__.)
            if (self.audio_indicator is None or self.current_times_audio is None
                    or any(t.numel() not in (1, rows) for t in self.current_times_audio)
                    or os.environ.get("LANPAINT_AMD_AV_TABLE", "1") == "0"):
                return False
            packed = self._indicator_pack(self.audio_indicator, x.shape)
            if packed is None or (self._es_opts is not None and not packed[2]):
                return False
        if self._es_opts is not None and (not self._es_opts["device"] or self.rng not in ("torch", "philox")):
            return False         # a custom distance_fn / a sharded batch keeps the stopper on the host; a gated loop
                                 # redoes its tentative half-step from a counter-based in-kernel generator only
        if self._overridden("langevin_dynamics") or self._overridden("score_model") or \
                self._overridden("prepare_step_size"):
            return False
        return x.dtype == torch.float32 and x.numel() > 0

    def _auto_ready(self, x, latent_mask, model_options):
        """graph="auto": has this job been seen running eagerly, with a backbone cheap enough on the host for the loop to
        be launch bound?  The first call with a new (latent_image, mask, model_options, shape) starts the record (and
        runs eagerly: its plain loop times the backbone calls); later calls of the same job ask it."""
        a = self._auto
        if a is not None and a[0][0]() is self.latent_image and a[0][1]() is latent_mask and a[0][2] == id(model_options) \
                and a[0][3] == x.shape and a[0][4] == x.device:
            if self._es_opts is not None:
                return False             # (the inner early stop is captured on request only: graph=True)
            return a[1] >= 1 and 1e6 * a[2] < self.AUTO_MAX_BACKBONE_HOST_US
        # [signature, eager calls seen, cheapest per-call mean of the backbone's host time so far (s)]
        # (weak references / an id: the record must not keep a finished job's tensors and options alive; a recycled id
        # only means one more eager call before the capture, whose key checks the dict by identity anyway)
        self._auto = [(weakref.ref(self.latent_image), weakref.ref(latent_mask), id(model_options), x.shape, x.device), 0,
                      float("inf")]
        return False

    def _auto_capture(self, key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """graph="auto": capture, but never let a backbone that cannot be captured take the call down, and take nothing
        on trust -- the first replay is compared with eager launches of the same call before the capture is used."""
        import warnings
        dev = x.device
        saved = (self._iterations_run, torch.cuda.get_rng_state(dev), self._es_opts, self._torch_consumed, self._philox_offset)

        def give_up(why):
            self._capturing = None
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
            self._iterations_run, _, self._es_opts, self._torch_consumed, self._philox_offset = saved
            torch.cuda.set_rng_state(saved[1], dev)
            self._graph_blocked = True
            cap = self._graphs.pop(key, None)
            if cap is not None:
                cap.alive = False
            self._last_cap = None
            warnings.warn("lanpaint_amd: graph='auto' stays with eager launches for this engine: " + why)
            return None

        try:
            cap = self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
        except Exception as e:                       # e.g. a host sync inside the backbone while the stream is capturing
            return give_up("the backbone could not be captured (%s: %s)" % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
        if cap is None or self.rng != "torch":
            return cap
        # one replay and one eager run of this very call on copies of x, from the same generator state: they must agree
        # bit for bit (same kernels, same noise stream), or something in the backbone does not survive capture
        try:
            xa, xb = x.clone(), x.clone()
            out_a = self._run_capture(cap, xa, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            torch.cuda.set_rng_state(saved[1], dev)
            self._last_cap = None
            out_b = self.LanPaint(xb, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            same = bool(torch.equal(out_a, out_b)) and bool(torch.equal(xa, xb))
        except Exception as e:
            return give_up("checking the capture against eager launches failed (%s)" % type(e).__name__)
        if not same:
            return give_up("a replayed sigma call does not reproduce the eager one (the backbone keeps state the graph does not see)")
        self._iterations_run, _, _, self._torch_consumed, self._philox_offset = saved
        torch.cuda.set_rng_state(saved[1], dev)
        return cap

    def _call_graphed(self, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """One sigma call with its think loop replayed as a hipGraph.  Only the part BETWEEN the replace step
        and the finalise is captured (N x [backbone, fused step] + the final backbone call): the prologue
        (lp_coeffs, replace) and the epilogue (lp_finalize) are ordinary launches that read / write the
        caller's tensors directly, so nothing of x / noise / out has to be staged through static buffers."""
        m_c, _ = _compact_mask(latent_mask, x.shape, x.device)
        # every pointer the captured launches bake in is part of the key (y and the mask; the tensors stay the caller's)
        key = (tuple(x.shape), x.device.index, int(n_steps), bool(IS_FLUX), bool(IS_FLOW), self.latent_image.data_ptr(),
               latent_mask.data_ptr(), m_c.data_ptr() if m_c is not None else 0, int(sigma.numel()),
               tuple(int(t.numel()) for t in current_times), id(model_options), seed, self.rng, self._hyper_key(),
               self.model_dtype, None if self._es_opts is None else (self._es_opts["threshold"], self._es_opts["patience_eff"],
                                                                      self._es_opts["trace"] is not None),
               # (AV: the captured launches bake the address of the indicator's bit-packed copy -- that address, not the tensor's
               # id(), which another tensor of the same shape can recycle)
               None if self.audio_indicator is None else (self._indicator_pack(self.audio_indicator, x.shape)[0].data_ptr(),
                                                          self.audio_correction is not None))
        cap = self._graphs.get(key)
        if cap is not None and cap.model_options is not model_options:
            del self._graphs[key]        # another dict at a recycled id(): the captured backbone calls used the old one
            cap.alive = False
            cap = None
        if cap is None:
            capture = self._auto_capture if self.graph == "auto" else self._capture
            cap = capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            if cap is None:          # not capturable after all (see _capture): the eager path
                return self.LanPaint(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
            while len(self._graphs) > self.MAX_GRAPHS:       # bound the static memory held by stale captures
                self._graphs.popitem(last=False)[1].alive = False
        else:
            self._graphs.move_to_end(key)
        return self._run_capture(cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)

    def _run_capture(self, cap, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW):
        """Replay `cap` for this call's tensors."""
        srcs = (x, sigma, current_times[0], current_times[1], current_times[2], self.noise)
        fast = cap.fast and all(t.dtype == torch.float32 and t.is_contiguous() for t in srcs) and self.noise.shape == x.shape \
            and (x.data_ptr() & 15) == 0 and (self.noise.data_ptr() & 15) == 0
        if not fast and cap.binding is not None:
            # the capture holds its own replace launch (node 0), which only takes dense fp32 16-byte-aligned caller
            # tensors: this call's do not qualify, so it runs as eager launches
            self._last_cap = None
            return self.LanPaint(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW)
        self._iterations_run += cap.ran
        self.last_inner_steps = cap.ran
        if fast:
            if not isinstance(model_options, dict) or "lanpaint_semantic_stop" not in model_options:
                cap.ident = (self.latent_image, latent_mask, model_options, x.shape, n_steps, seed, self.rng,
                             self.latent_image.data_ptr(), latent_mask.data_ptr(), getattr(latent_mask, "_lp_bits", None),
                             getattr(latent_mask, "_lp_u8", None), x.device, sigma.numel(), current_times[0].numel(),
                             current_times[1].numel(), current_times[2].numel(), self._override_state(), self._hyper_key())
                self._last_cap = cap
            return self._replay_fast(cap, x, sigma, current_times)
        self._last_cap = None
        st = self._prologue(x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW, ws=cap.ws,
                            ds=cap.es["ds"] if cap.es is not None else None)
        cap.graph.replay()
        if self.rng == "torch":        # the replayed launches consumed this much of the generator's stream
            gen = self._generator(x.device)
            gen.set_offset(gen.get_offset() + cap.launches)
            self._torch_consumed += cap.launches
        out = self._epilogue(st, cap.final, rng_bump=(cap.counter, cap.launches) if self.rng == "philox" else None,
                             in_graph=cap.final_in_graph)
        if cap.es is not None and st.es is not None:
            self._es_after_replay(cap, st.es["seq"], x.device)
        return out

    def _replay_fast(self, cap, x, sigma, current_times):
        """Steady-state replay: the two launches around the graph (replace + coefficient table, lp_finalize) reuse
        the descriptors snapshotted at capture; only the caller's pointers (x, noise, sigma, times, out) change.
        With the raw hipGraphExec_t the whole sequence is ONE trip through the FFI (lp_replay_call)."""
        lib, stream = self._lib, self._stream(x.device)
        ve, abt = current_times[0], current_times[1]
        t_src = current_times[2] if cap.flow else ve
        out = torch.empty_like(x)
        k0, f = cap.k0_desc, cap.f_desc
        k0.x, k0.noise = x.data_ptr(), self.noise.data_ptr()
        # the replace launch also rebuilds the coefficient table from this call's sigma / times (LP_PH_COEFFS)
        k0.t_ve, k0.t_abt, k0.t_rsig, k0.t_model = ve.data_ptr(), abt.data_ptr(), sigma.data_ptr(), t_src.data_ptr()
        if cap.final_in_graph:         # the captured lp_finalize reads these two through the table the replace publishes
            k0.io_table_val[0], k0.io_table_val[1] = k0.x, out.data_ptr()
        else:
            f.x_dst, f.out = k0.x, out.data_ptr()
        if self.rng == "torch":        # generator state in (published by the replace launch), state out
            gen = self._generator(x.device)
            off = gen.get_offset()
            k0.rng_state_val[0], k0.rng_state_val[1] = off, gen.initial_seed()
            gen.set_offset(off + cap.launches)
            self._torch_consumed += cap.launches
        seq = 0
        if cap.es is not None:         # the replace launch resets the device-side stopper for this call
            seq = k0.es_seq_base = cap.es["ds"].next_seq()
        if cap.raw_exec is not None:
            _cabi.check(lib.lp_replay_call(ctypes.byref(cap.call), stream), "lp_replay_call")
        else:
            _cabi.check(lib.lp_step(ctypes.byref(k0), stream), "lp_step")
            cap.graph.replay()
            if not cap.final_in_graph:
                _cabi.check(lib.lp_finalize(ctypes.byref(f), stream), "lp_finalize")
        if cap.es is not None:
            self._es_after_replay(cap, seq, x.device)
        return out

    def _es_after_replay(self, cap, seq, dev):
        """The replayed loop decides its own length: remember what to collect.  With the trace requested, or with
        torch's generator to be left exactly where the reference leaves it, collect it now -- ONE host read per
        sigma call; otherwise at the next call / when iterations_run is read."""
        es = dict(cap.es, trace=self._es_opts["trace"] if self._es_opts is not None else None,
                  tags=self._es_opts["tags"] if self._es_opts is not None else (None, None, None))
        inc = self._randn_policy(dev, cap.ws.x_t.numel())[1] if self.rng == "torch" else 0
        self._es_pending = (cap.es["ds"], seq, cap.n_steps, es, dev, inc)
        if inc or es["trace"] is not None:
            self._es_resolve()

    def _rng_state(self, dev):
        """Device u64[4] read by captured launches.  rng="philox": [0] = launch-sequence base, ONE per device and
        bumped by every replay so the streams of different captures never overlap.  rng="torch": (generator
        offset, seed) published by the replace launch of each call.  [2], [3]: the I/O table of the call in flight
        (address of the sampler latent x, address of `out`), published by the replace launch for the captured
        lp_finalize.  [4]: the word that voids a captured lp_finalize when 0 (io_table word 2)."""
        state = self._rng_counters.get(dev)
        if state is None:
            state = self._rng_counters[dev] = torch.zeros(8, dtype=torch.int64, device=dev)
            state[4] = 1          # [4]: "this sigma call is valid" -- 0 voids a captured lp_finalize (a speculated call, lp_node_call)
        return state

    _capture_sentinels = {}
    _capture_warned = set()

    @classmethod
    def _warm_capture_state(cls, dev):
        """Once per device, before this process's first capture through the engine: a trivial capture OUTSIDE inference mode whose
        graph object is then KEPT for the life of the process.  torch allocates the device generator's graph-capture state (its
        seed / offset tensors) when the FIRST graph registers with the generator, updates it in place at every `capture_begin`,
        and frees it again when the LAST registered graph dies.  Allocated under torch.inference_mode() -- how ComfyUI runs its
        nodes -- the state is inference tensors, and the first capture attempted outside inference mode while any graph is still
        alive dies inside capture_begin on that in-place update, leaving the generator in its capturing state (every later
        torch.randn of the process then raises "Offset increment outside graph capture").  Found by the property test of the
        capture state machine (tests/test_gpu_state_machine.py) -- twice: a warm-up capture that was freed again only moved the
        hazard to the next moment no graph was alive.  With one sentinel graph registered from normal mode and never freed, the
        state stays allocated as normal tensors, which either mode may update."""
        if cls._capture_sentinels.get(dev.index) is not None:
            return
        try:
            with torch.inference_mode(False), torch.no_grad():
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    torch.zeros(1, device=dev)
                torch.cuda.current_stream(dev).wait_stream(s)
                cls._capture_sentinels[dev.index] = g          # recorded on SUCCESS only: a failed warm-up is tried again
        except Exception as e:     # (a torch build that refuses: the engine's own captures will say why)
            if dev.index not in cls._capture_warned:
                cls._capture_warned.add(dev.index)
                import warnings
                warnings.warn("lanpaint_amd: the capture warm-up on %s failed (%s: %s); it is retried before the next capture"
                              % (dev, type(e).__name__, str(e).splitlines()[0] if str(e) else ""))

    def _capture(self, key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                 replace_in_graph=None):
        """Capture one sigma call.  `replace_in_graph` (default: on, LANPAINT_AMD_REPLACE_IN_GRAPH=0 turns it off): the
        replace launch is captured too, as the FIRST node, and every replay refreshes that node's arguments (the caller's
        x / noise / sigma / times, this call's out, generator state) with hipGraphExecKernelNodeSetParams -- the whole
        sigma call is then ONE hipGraphLaunch with nothing eager in front of it (lp_replay_call, replace_binding).  Needs the
        raw graph handles and a call the steady-state path takes (dense fp32 tensors, a fusable replace form); anything
        else is captured the round-2 way, with the replace launch outside the graph."""
        dev = x.device
        self._warm_capture_state(dev)
        if replace_in_graph is None:
            replace_in_graph = os.environ.get("LANPAINT_AMD_REPLACE_IN_GRAPH", "1") != "0"
        raw_ok = self.rng in ("philox", "torch") and os.environ.get("LANPAINT_AMD_RAW_GRAPH", "1") != "0"
        replace_in_graph = bool(replace_in_graph and raw_ok)
        counter = self._rng_state(dev)
        cap = _CapturedCall(counter)
        if replace_in_graph:
            try:
                cap.graph = torch.cuda.CUDAGraph(keep_graph=True)  # the hipGraph_t stays: node 0 has to be found in it
            except (TypeError, RuntimeError):                      # a torch build without `keep_graph`: the round-2 layout
                replace_in_graph = False
        # captures that differ only in the step count (KSamplerX0Inpaint's n_eff ramp) share ONE workspace: sigma
        # calls are serialised on the stream, and the n_steps-independent replace launch can then be enqueued
        # before the count is known (begin_call / finish_call)
        ws_key = (tuple(x.shape), dev.index, self.model_dtype)
        cap.ws = self._static_ws.get(ws_key)
        if cap.ws is None:
            if len(self._static_ws) >= self.MAX_GRAPHS:
                self._static_ws.pop(next(iter(self._static_ws)))
            cap.ws = self._static_ws[ws_key] = _Workspace(x.detach().to(torch.float32).contiguous(), static_io=True,
                                                          model_dtype=self.model_dtype)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        it0 = self._iterations_run
        rng_state = torch.cuda.get_rng_state(dev)   # warm-up + capture must not consume the user's torch stream
        gen = self._generator(dev)
        off0, own0 = gen.get_offset(), self._torch_consumed
        es_user = self._es_opts
        if es_user is not None:                    # the warm-up below is not the caller's run: keep it out of their trace
            self._es_opts = dict(es_user, trace=None)
            # without a trace to fill nobody needs the verdict of the loop's LAST iteration: its launch closes the loop
            # itself (LP_FL_ES_CLOSE) and the closing decision kernel -- one more graph node -- is not captured
            self._es_close = es_user["trace"] is None
        with torch.cuda.stream(side):              # one complete eager call on the side stream: lazy inits
            xw = x.detach().clone()
            st = self._prologue(xw, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                ws=cap.ws)
            self._epilogue(st, self._think_and_final_model(st, model_options, seed))
            # the state the captured launches start from.  With the replace launch inside the graph it is only
            # described here (descriptor snapshot) and enqueued as the first captured launch below.
            st = self._prologue(xw, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                ws=cap.ws, defer_launch=replace_in_graph)
            if replace_in_graph and not (st.k0_desc is not None and st.replace_kind_static and st.xc is st.input_x
                                         and (st.xc.data_ptr() & 15) == 0 and (self.noise.data_ptr() & 15) == 0):
                replace_in_graph = False           # not a call the steady-state path takes: the replace stays outside
                self._launch_step_desc(st.k0_desc, st.stream)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # did the warm-up (backbone included) draw from torch's generator?  Then only torch's own replay() keeps
        # the captured Philox offsets moving and the graph must not be launched behind its back.
        torch_rng_used = (gen.get_offset() - off0) != (self._torch_consumed - own0)
        if torch_rng_used and self.rng == "torch":
            # the engine's in-kernel draws and the backbone's own draws would have to interleave inside the graph
            # exactly as they do eagerly: not representable with one published offset -> this engine stays eager
            torch.cuda.set_rng_state(rng_state, dev)
            self._iterations_run = it0
            self._graph_blocked = True
            self._es_opts = es_user
            return None
        if torch_rng_used and replace_in_graph:
            # (the graph has to go through torch's replay(), which knows nothing of node arguments: capture again the
            # round-2 way)
            torch.cuda.set_rng_state(rng_state, dev)
            self._iterations_run = it0
            self._es_opts = es_user
            return self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX, IS_FLOW,
                                 replace_in_graph=False)
        self._iterations_run = it0
        self._capturing, self._cap_offset = counter, 0
        # No cyclic garbage collection while the stream is capturing: a collection that happens to run inside the captured region
        # finalises whatever unreachable objects the process holds -- among them other engines' captures, whose destructors destroy
        # hipGraph / hipGraphExec handles and free their memory pools.  Those calls are illegal on a capturing thread; the error
        # surfaces in a C++ destructor and aborts the process ("Fatal Python error: Aborted ... Garbage-collecting", met once in
        # the 400-sequence property test, which leaves hundreds of dead engines behind).  torch collects BEFORE a capture; it
        # cannot stop the collector from firing during one.  (Reference counting still frees what drops to zero: tensors go back
        # to the caching allocator, which knows about captures.)
        _gc_hold()
        try:
            # thread_local: a live RCCL communicator's watchdog thread issues HIP calls of its own;
            # in the default "global" mode those would invalidate this thread's capture
            f = _cabi.LpFinalDesc()
            with torch.cuda.graph(cap.graph, stream=side, capture_error_mode="thread_local"):
                if replace_in_graph:
                    self._launch_step_desc(st.k0_desc, self._stream(dev))
                cap.final = self._think_and_final_model(st, model_options, seed)
                dense_ok = self._fill_final_desc(f, st, cap.final, st.out)
                if self.rng == "philox":
                    f.rng_bump_ptr, f.rng_bump = counter.data_ptr(), self._cap_offset
                if dense_ok and st.out is not None:
                    # the finalise is the last node of the graph: it takes the caller's x and this call's `out`
                    # from the table the replace launch of the same call publishes
                    f.io_table = counter.data_ptr() + 16
                    _cabi.check(self._lib.lp_finalize(ctypes.byref(f), self._stream(dev)), "lp_finalize")
                    cap.final_in_graph = True
        finally:
            self._capturing = None
            _gc_release()
        cap.launches = self._cap_offset
        self._es_opts = es_user
        torch.cuda.set_rng_state(rng_state, dev)
        cap.ran = self._iterations_run - it0
        self._iterations_run = it0
        cap.keep = st                              # descriptor-side tensors referenced by the baked launches
        # descriptors of the launches whose arguments change from call to call, for the steady-state replay path
        cap.rows, cap.flow = st.rows, st.flow
        cap.hyper = _cabi.LpHyper.from_buffer_copy(self._hyper)
        cap.k0_desc = st.k0_desc
        cap.f_desc = f
        cap.model_options = model_options
        cap.es, cap.n_steps = st.es, st.n_steps
        cap.fast = bool(dense_ok and st.k0_desc is not None and st.replace_kind_static and st.xc is st.input_x)
        if replace_in_graph:
            ok = cap.fast and cap.final_in_graph
            if ok:
                try:
                    cap.graph.instantiate()
                    raw_graph = int(cap.graph.raw_cuda_graph())
                    cap.raw_exec = int(cap.graph.raw_cuda_graph_exec()) or None
                    b = _cabi.LpGraphBinding()
                    ok = cap.raw_exec is not None and self._lib.lp_graph_bind_replace(
                        raw_graph, ctypes.byref(cap.k0_desc), ctypes.byref(b)) == _cabi.LP_OK
                    if ok:
                        cap.binding = b
                        tg, te = ctypes.c_void_p(), ctypes.c_void_p()
                        if self._lib.lp_graph_clone_tail(raw_graph, ctypes.byref(tg), ctypes.byref(te)) == _cabi.LP_OK:
                            cap.tail_handles = (tg.value, te.value)
                except Exception:
                    ok = False
            if not ok:       # not a steady-state call after all, or this runtime does not give the handles: round-2 layout
                return self._capture(key, x, sigma, latent_mask, current_times, n_steps, model_options, seed, IS_FLUX,
                                     IS_FLOW, replace_in_graph=False)
        if cap.fast:
            if cap.binding is None and raw_ok and not torch_rng_used:
                try:
                    cap.raw_exec = int(cap.graph.raw_cuda_graph_exec()) or None
                except Exception:
                    cap.raw_exec = None
            c = cap.call = _cabi.LpCallDesc()          # hyper = NULL: no separate lp_coeffs launch, the replace does it
            c.replace = ctypes.pointer(cap.k0_desc)
            c.final = None if cap.final_in_graph else ctypes.pointer(f)
            c.rows, c.coef_table, c.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.raw_exec
            if cap.binding is not None:
                c.replace_binding = ctypes.pointer(cap.binding)
                if cap.tail_handles is not None:
                    t = cap.tail = _cabi.LpCallDesc()  # the graph minus node 0: its replace launch went ahead (begin_call)
                    t.rows, t.coef_table, t.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.tail_handles[1]
            elif cap.raw_exec is not None and cap.final_in_graph:
                t = cap.tail = _cabi.LpCallDesc()      # the graph alone: its replace launch went ahead (begin_call)
                t.rows, t.coef_table, t.graph_exec = st.rows, cap.ws.coef.data_ptr(), cap.raw_exec
        cap.key = key
        self._graphs[key] = cap
        return cap
