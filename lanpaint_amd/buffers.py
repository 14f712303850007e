"""Device buffers and per-call records of the engine: the workspace a sigma call runs in, the device-side early-stop state, what
one call carries from its prologue to its epilogue, and a captured sigma call (hipGraph + the descriptors its replay patches)."""
from __future__ import annotations

import weakref

import torch

from . import _cabi
from ._util import tensor_version

class _Workspace:
    """Device buffers reused across sigma calls of one engine (torch-owned).  static_io: also owns the
    backbone-input / final-x buffers (a captured call bakes their addresses)."""

    def __init__(self, like: torch.Tensor, static_io: bool = False, model_dtype=None):
        self.shape, self.device = tuple(like.shape), like.device
        self.x_t = torch.empty_like(like)
        self.C = torch.empty_like(like)
        self.coef = torch.empty((like.shape[0], _cabi.LP_COEF_STRIDE), dtype=torch.float32, device=like.device)
        self.coef_av = None      # lazily: [2 * rows][LP_COEF_STRIDE], two time sets per row (AV packs, LP_FL_AV)
        self.av_times = None     # lazily: [4][2 * rows] interleaved (VE, abt, replace sigma, model time) inputs of that table
        self.x0s = []            # lazily: rotating buffers for LangevinState.x0 (early stop only)
        self.static_io = static_io
        if static_io:
            self.x_final = torch.empty_like(like)
            self.x_in = self.x_final if model_dtype is None else torch.empty_like(like, dtype=model_dtype)

    def matches(self, like):
        return self.shape == tuple(like.shape) and self.device == like.device


class _DeviceStop:
    """Buffers of the inner early stop evaluated on the device (LP_FL_ES): the lp_es_state, three rotating x0s
    buffers, the accumulator sets the blocks add their sums into and the pinned-host mailbox the trace records go to."""

    def __init__(self, like: torch.Tensor, n_steps: int):
        dev = like.device
        self.shape, self.device, self.n_cap = tuple(like.shape), dev, max(8, int(n_steps))
        self.x0s = [torch.empty_like(like) for _ in range(3)]
        self.x_te = torch.empty_like(like)      # gated loops: the state after the tentative half-step (lp_step_desc.es_xte)
        init = _cabi.LpEsState()
        init.cur_slot = init.anchor_slot = -1
        for k in range(3):
            init.x0s_buf[k] = self.x0s[k].data_ptr()
        raw = torch.frombuffer(bytearray(bytes(init) * 2), dtype=torch.uint8)     # two slots (folded gated loops ping-pong)
        self.state = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
        self.state.copy_(raw)
        # the accumulator sets the blocks of an early-stop launch add their sums into (LP_ES_ACC_DOUBLES)
        self.partials = torch.zeros(_cabi.LP_ES_ACC_DOUBLES, dtype=torch.float64, device=dev)
        self.mailbox = torch.zeros(_cabi.LP_ES_TRACE0 + 8 * self.n_cap, dtype=torch.float64).pin_memory()
        self.f64 = self.mailbox.numpy()
        self.i64 = self.mailbox.view(torch.int64).numpy()
        self.seq_base = 0
        self.seen_total = 0         # of the device's running iteration count, what the engine has accounted already
        self.ring = None            # (weakref(mask), version, ring tensor | None, bit-packed ring | None)

    def matches(self, like, n_steps):
        return self.shape == tuple(like.shape) and self.device == like.device and n_steps <= self.n_cap

    def next_seq(self):
        self.seq_base += 2 * _cabi.LP_ES_SEQ_DONE
        return self.seq_base

    def ring_for(self, key, mask):
        """Mask-edge ring weight (earlystop.py:32-49; 4-D latents only) of the dense fp32 `mask`, computed once per
        mask tensor object `key` and version."""
        c = self.ring
        ver = (tensor_version(key), key.data_ptr())
        # (no version counter -- inference mode --: recomputed on every call; always into the SAME buffers when the shape allows,
        # because captured early-stop launches bake the ring's address)
        if c is None or c[0]() is not key or c[1] != ver or ver[0] == -1:
            ring = None
            if mask.dim() == 4:
                old = c[2] if c is not None else None
                ring = old if (old is not None and old.shape == mask.shape and old.device == mask.device) else torch.empty_like(mask)
                b, ch, h, w = mask.shape
                with torch.cuda.device(mask.device):
                    _cabi.check(_cabi.load().lp_boundary_ring(mask.data_ptr(), ring.data_ptr(), b * ch, h, w,
                                                              torch.cuda.current_stream(mask.device).cuda_stream),
                                "lp_boundary_ring")
            bits = None
            if ring is not None:       # the bit-packed form the hard-mask kernels read (LP_FL_ES_RING_BITS): ring pixels are inpaint
                # pixels, so with a binary mask the weight (1 - m) on them is exactly 1 and the ring IS a bit per element
                old_bits = c[3] if c is not None else None
                n_bytes = _cabi.mask_bits_bytes(ring.numel())
                bits = old_bits if (old_bits is not None and old_bits.numel() == n_bytes and old_bits.device == ring.device) \
                    else torch.empty(n_bytes, dtype=torch.uint8, device=ring.device)
                with torch.cuda.device(mask.device):
                    _cabi.check(_cabi.load().lp_pack_mask(ring.data_ptr(), ring.numel(), 0, bits.data_ptr(), None,
                                                          torch.cuda.current_stream(mask.device).cuda_stream), "lp_pack_mask")
            self.ring = c = (weakref.ref(key), ver, ring, bits)
        return c[2]

    def ring_bits(self):
        """Bit-packed form of the ring `ring_for` returned last (None for latents without a ring)."""
        return self.ring[3] if self.ring is not None else None

    def wait(self, seq, device):
        """Block until the mailbox sequence word reaches `seq` (spin briefly, then sleep on the stream)."""
        i64 = self.i64
        for _ in range(20000):
            if i64[0] >= seq:
                return
        torch.cuda.current_stream(device).synchronize()
        if i64[0] < seq:
            raise RuntimeError("early-stop mailbox was not written (expected sequence %d, found %d)" % (seq, int(i64[0])))


class _CallState:
    """Everything one sigma call carries from its prologue to its loop and epilogue."""
    __slots__ = ("input_x", "xc", "shape", "n_el", "rows", "flow", "ws", "stream", "sigma", "y", "m", "m_c", "m_flag", "abt",
                 "current_times", "base_flags", "keep", "t_model", "sigma_model", "compat", "n_steps", "x_final", "x_in",
                 "xin_flag", "k0_desc", "replace_kind_static", "out", "es")


class _CapturedCall:
    """The think loop + final backbone call of one sigma call captured as a hipGraph, with the workspace
    whose addresses it bakes in and the device-side Philox counter its launches read."""

    def __init__(self, counter):
        self.graph = torch.cuda.CUDAGraph()
        self.counter = counter
        self.ws = None
        self.final = None        # the backbone's final output object (static tensors)
        self.ran = 0
        self.launches = 0
        self.keep = None
        self.fast = False        # steady-state replay may reuse the snapshotted descriptors
        self.rows, self.flow, self.hyper, self.k0_desc, self.f_desc = 0, False, None, None, None
        self.call = None         # lp_call_desc: the whole enqueue sequence of a replay in one C call
        self.raw_exec = None     # hipGraphExec_t, when launching it without torch's replay() is equivalent
        self.ident = None        # what the caller passed last time (identity pre-check of the next call)
        self.final_in_graph = False   # lp_finalize is a node of the graph (reads x / out through the I/O table)
        self.es = None                # early stop evaluated on the device inside the graph (LP_FL_ES_GATED): options + buffers
        self.n_steps = 0
        self.key = None               # its key in the engine's graph table (siblings differ in the step count only)
        self.tail = None              # lp_call_desc that launches the graph alone (the replace went ahead, begin_call)
        self.model_options = None     # the dict the captured backbone calls were made with (kept alive: its id is in the key)
        self.alive = True             # still in the engine's graph table
        self.siblings = {}            # n_steps -> the capture of the same call shape for that count (finish_call)
        self.times_seen = ()          # the (VE sigma, abt, flow t) tuples that passed the identity pre-check
        self.node_table = None        # (exec array by inner-step count, captures, -, graphs seen, options): lp_node_call's table
        self.binding = None           # lp_graph_binding: the replace launch is node 0 of the graph (ONE hipGraphLaunch per call)
        self.tail_handles = None      # (hipGraph_t, hipGraphExec_t) of the same graph without node 0 (begin_call / finish_call)
        self.sigma_root = None        # (hipGraph_t, hipGraphExec_t, LpGraphBinding) of the same graph with the sigma algebra folded
                                      # into node 0 (lp_graph_clone_sigma_root): the node path's ONE launch per call; False = tried, no

    def __del__(self):
        sr, self.sigma_root = self.sigma_root, None
        if sr:
            try:
                _cabi.load().lp_graph_release(sr[0], sr[1])
            except Exception:
                pass
        h, self.tail_handles = self.tail_handles, None
        if h is not None:
            try:
                _cabi.load().lp_graph_release(h[0], h[1])
            except Exception:
                pass
