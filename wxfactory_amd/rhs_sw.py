"""Shallow-water RHS on cubed-sphere panels: host-side mirror of the reference's callable.

Same contract as reference wx_factory/rhs/rhs_sw.py:38-56 (`RhsShallowWater.__call__`): takes a
state of any shape whose size matches `(3, H, H, n^2)`, returns the right-hand side in that shape
and dtype (float64 / complex128).  One plan per cube panel; the exchange of rhs_sw.py:103-150
(vector message + scalar message through Ineighbor_alltoall) is one 3-variable edge line per
neighbour, rotated and flipped by the pack kernel.
"""
import ctypes
from typing import Dict, Optional, Sequence

import numpy
import torch

from . import _lib
from ._lib import DfrOps, SwMetric, check
from .exchange import PanelExchange  # noqa: F401
from .panel_rhs import PanelRhs, _ptr_array

_DTYPES = {torch.float64: _lib.WX_F64, torch.complex128: _lib.WX_C128}
_TOPO = ("hsurf", "dzdx1", "dzdx2", "hsurf_itf_i", "hsurf_itf_j")


class SwPlan:
    def __init__(self, n: int, H: int, panel: int, ops: Dict[str, numpy.ndarray], metric: Dict[str, torch.Tensor],
                 dtype: torch.dtype = torch.float64, dual: bool = False, on_panel_edge=(True, True, True, True)):
        self.lib = _lib.load()
        if dtype not in _DTYPES:
            raise TypeError(f"dtype must be float64 or complex128, not {dtype}")
        # complex128 storage can run true complex arithmetic (WX_C128) or first-order dual-number
        # arithmetic (WX_DUAL128: same complex-step JVP to O(eps^2), much cheaper); see include/wxhip.h
        self.dual = bool(dual) and dtype == torch.complex128
        wx_dtype = _lib.WX_DUAL128 if self.dual else _DTYPES[dtype]
        self.n, self.H, self.panel, self.dtype = n, H, panel, dtype
        self.shape = (3, H, H, n * n)
        self._ops, self._metric = ops, metric
        self.on_panel_edge = tuple(bool(x) for x in on_panel_edge)
        self._keep = []
        o = DfrOps()
        for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter"):
            a = numpy.ascontiguousarray(ops[k], dtype=numpy.float64)
            self._keep.append(a)
            setattr(o, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        pts, ii, jj = (H, H, n * n), (H, H + 2, 2 * n), (H + 2, H, 2 * n)
        expect = {k: pts for k in _lib.SW_METRIC_FIELDS}
        for k in ("sqrtG_itf_i", "H_contra_11_itf_i", "H_contra_21_itf_i", "hsurf_itf_i"):
            expect[k] = ii
        for k in ("sqrtG_itf_j", "H_contra_12_itf_j", "H_contra_22_itf_j", "hsurf_itf_j"):
            expect[k] = jj
        expect["boundary_sn"] = expect["boundary_we"] = (H * n,)
        self.device = metric["sqrtG"].device
        m = SwMetric()
        for k in _lib.SW_METRIC_FIELDS:
            t = metric.get(k)
            if t is None:
                if k not in _TOPO:
                    raise ValueError(f"metric[{k!r}] is required")
                setattr(m, k, None)
                continue
            if tuple(t.shape) != expect[k] or t.dtype != torch.float64 or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"metric[{k!r}]: need contiguous float64 {expect[k]} on {self.device}, "
                                 f"got {t.dtype} {tuple(t.shape)} on {t.device}")
            self._keep.append(t)
            setattr(m, k, t.data_ptr())
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            flags = (ctypes.c_int * 4)(*[int(x) for x in self.on_panel_edge])
            check(self.lib.wx_sw_plan_create_tile(ctypes.byref(self._h), n, H, wx_dtype, panel, flags, ctypes.byref(o),
                                                  ctypes.byref(m)), "wx_sw_plan_create_tile")
        self.edge_count = int(self.lib.wx_sw_edge_count(self._h))

    def twin(self, dtype, dual: bool = False):
        return SwPlan(self.n, self.H, self.panel, self._ops, self._metric, dtype=dtype, dual=dual,
                      on_panel_edge=self.on_panel_edge)

    def _check_q(self, q):
        if q.dtype != self.dtype or q.numel() != 3 * self.H * self.H * self.n**2 or not q.is_contiguous() \
                or q.device != self.device:
            raise ValueError(f"state must be a contiguous {self.dtype} tensor of {self.shape} on {self.device}")

    def extrap_pack(self, q, send_ptrs: Optional[Sequence[int]]):
        self._check_q(q)
        arr = _ptr_array(send_ptrs)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_extrap_pack(self._h, q.data_ptr(), arr, st), "wx_sw_extrap_pack")

    def rhs(self, q, halo_ptrs: Optional[Sequence[int]], out, region: int = _lib.WX_REGION_ALL):
        self._check_q(q)
        self._check_q(out)
        arr = _ptr_array(halo_ptrs)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_rhs(self._h, q.data_ptr(), arr, out.data_ptr(), region, st), "wx_sw_rhs")

    def rhs_axpy(self, q, halo_ptrs, y, out, a, b, c, region: int = _lib.WX_REGION_ALL, z=None, d: float = 0.0):
        """out = a*y + b*q + c*R(q) in the RHS launch (y may be None)."""
        if z is not None:
            raise NotImplementedError("the shallow-water kernel fuses one extra array (y) only")
        self._check_q(q)
        self._check_q(out)
        if y is not None:
            self._check_q(y)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_rhs_axpy(self._h, q.data_ptr(), _ptr_array(halo_ptrs), y.data_ptr() if y is not None else None,
                                      out.data_ptr(), a, b, c, region, st), "wx_sw_rhs_axpy")

    def extrap_pack_ring(self, q, send_ptrs):
        """The tile-edge lines alone (direct form): wx_sw_extrap_pack_ring."""
        self._check_q(q)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_extrap_pack_ring(self._h, q.data_ptr(), _ptr_array(send_ptrs), st), "wx_sw_extrap_pack_ring")

    def rhs_direct(self, q, halo_ptrs, out, region=_lib.WX_REGION_ALL, y=None, coef=None):
        """R(q) - or coef = (a, b, c): a*y + b*q + c*R(q) - with no interface buffer (wx_sw_rhs_direct)."""
        self._check_q(q)
        self._check_q(out)
        a, b, c = coef[:3] if coef is not None else (0.0, 0.0, 1.0)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_rhs_direct(self._h, q.data_ptr(), _ptr_array(halo_ptrs), y.data_ptr() if y is not None else None,
                                        out.data_ptr(), a, b, c, 0 if coef is None else 1, region, st), "wx_sw_rhs_direct")

    def reserve(self):
        """Setup-time allocation of the second interface slot (stage pipeline): wx_sw_plan_reserve."""
        if getattr(self, "_reserved", False):
            return
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("SwPlan.reserve: the stage pipeline's buffers do not exist yet and a stream capture is in "
                               "progress - call reserve() (or RhsShallowWater.reserve()) before it")
        with torch.cuda.device(self.device):
            check(self.lib.wx_sw_plan_reserve(self._h, _lib.WX_RESERVE_STAGE), "wx_sw_plan_reserve")
        self._reserved = True

    def extrap_pack_slot(self, q, send_ptrs, slot: int):
        self._check_q(q)
        if slot == 1:
            self.reserve()
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_extrap_pack_slot(self._h, q.data_ptr(), _ptr_array(send_ptrs), slot, st), "wx_sw_extrap_pack_slot")

    def stage(self, q, halo_ptrs, y, out, a, b, c, region, itf_in: int, next_send, prepare_next: bool):
        """out = a*y + b*q + c*R(q) reading faces from slot itf_in; with prepare_next also the faces of `out` into the
        other slot / next_send (wx_sw_stage)."""
        self._check_q(q)
        self._check_q(out)
        if y is not None:
            self._check_q(y)
        if prepare_next or itf_in == 1:
            self.reserve()
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_stage(self._h, q.data_ptr(), _ptr_array(halo_ptrs), y.data_ptr() if y is not None else None,
                                   out.data_ptr(), a, b, c, region, itf_in, _ptr_array(next_send), int(bool(prepare_next)), st),
              "wx_sw_stage")

    def close(self):
        if self._h:
            self.lib.wx_sw_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SwBatch:
    """All panels of a rank in one launch per phase (wx_sw_batch_*): the edge buffers are the
    exchange's persistent send/halo slots, the states are slices of one stacked tensor."""

    def __init__(self, plans: Dict[int, SwPlan], exchange: PanelExchange, exchange2: PanelExchange = None):
        """exchange2: the second edge-buffer set of the stage pipeline (wx_sw_batch_create_pipelined; the plans are
        reserved here)."""
        self.lib = _lib.load()
        self.panels = sorted(plans)
        n = len(self.panels)
        first = plans[self.panels[0]]
        self.device, self.dtype = first.device, first.dtype
        self.stride = 3 * first.H * first.H * first.n**2
        handles = (ctypes.c_void_p * n)(*[plans[p]._h for p in self.panels])

        def edges(ex):
            send = ((ctypes.c_void_p * 4) * n)()
            halo = ((ctypes.c_void_p * 4) * n)()
            for i, p in enumerate(self.panels):
                for e in range(4):
                    send[i][e] = ex.send_view(p, e).data_ptr()
                    halo[i][e] = ex.halo_view(p, e).data_ptr()
            return send, halo

        send, halo = edges(exchange)
        self._keep = (plans, exchange, exchange2)
        self._h = ctypes.c_void_p()
        self.pipelined = exchange2 is not None
        with torch.cuda.device(self.device):
            if self.pipelined:
                for pl in plans.values():
                    pl.reserve()
                send2, halo2 = edges(exchange2)
                check(self.lib.wx_sw_batch_create_pipelined(ctypes.byref(self._h), handles, n, send, halo, send2, halo2),
                      "wx_sw_batch_create_pipelined")
            else:
                check(self.lib.wx_sw_batch_create(ctypes.byref(self._h), handles, n, send, halo), "wx_sw_batch_create")
        # True: every tile's neighbours are in this batch - rhs_direct forms the tile-edge lines itself, no ring pack in front
        pulls = ctypes.c_int(0)
        check(self.lib.wx_sw_batch_direct_pulls(self._h, ctypes.byref(pulls)), "wx_sw_batch_direct_pulls")
        self.pulls = bool(pulls.value)

    def extrap_pack_ring(self, q):
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_batch_extrap_pack_ring(self._h, q.data_ptr(), self.stride, st), "wx_sw_batch_extrap_pack_ring")

    def rhs_direct(self, q, out, region, y=None, coef=None):
        a, b, c = coef[:3] if coef is not None else (0.0, 0.0, 1.0)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_batch_rhs_direct(self._h, q.data_ptr(), y.data_ptr() if y is not None else None, out.data_ptr(),
                                              self.stride, a, b, c, 0 if coef is None else 1, region, st), "wx_sw_batch_rhs_direct")

    def extrap_pack_slot(self, q, slot: int):
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_batch_extrap_pack_slot(self._h, q.data_ptr(), self.stride, slot, st), "wx_sw_batch_extrap_pack_slot")

    def stage(self, q, y, out, a, b, c, region, itf_in: int, prepare_next: bool):
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_batch_stage(self._h, q.data_ptr(), y.data_ptr() if y is not None else None, out.data_ptr(),
                                         self.stride, a, b, c, region, itf_in, int(bool(prepare_next)), st), "wx_sw_batch_stage")

    def extrap_pack(self, q):
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_sw_batch_extrap_pack(self._h, q.data_ptr(), self.stride, st), "wx_sw_batch_extrap_pack")

    def rhs(self, q, out, region, y=None, coef=None):
        st = torch.cuda.current_stream(self.device).cuda_stream
        if coef is None:
            check(self.lib.wx_sw_batch_rhs(self._h, q.data_ptr(), out.data_ptr(), self.stride, region, st), "wx_sw_batch_rhs")
        else:
            check(self.lib.wx_sw_batch_rhs_axpy(self._h, q.data_ptr(), y.data_ptr() if y is not None else None,
                                                out.data_ptr(), self.stride, coef[0], coef[1], coef[2], region, st),
                  "wx_sw_batch_rhs_axpy")

    def close(self):
        if self._h:
            self.lib.wx_sw_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RhsShallowWater(PanelRhs):
    """R(Q) for the panels this rank owns; same calling convention as RhsEuler3D
    (contract of rhs/rhs_sw.py:38-56).  A stacked, contiguous state of several panels is evaluated
    with ONE launch per phase for all of them (SwBatch)."""

    batched = True
    overlapped_entry = "wx_sw_rhs_overlapped"

    @property
    def supports_pipeline(self) -> bool:
        """The stage pipeline (wx_sw_stage: the stage's kernel extrapolates its own output) pays where an evaluation takes two
        launches; where the direct form is taken a fused stage is one launch already, and faster (S7: 49 against 54 us).  The
        direct form is only taken by the batched evaluation of several local tiles (_run_batched): a rank with one tile - the
        six-GPU decomposition - evaluates in two launches and keeps the pipeline."""
        return not (self._use_direct(torch.float64) and self.batched and len(self.panels) > 1)
    # the direct form (no interface buffer: ring-only pack - none when SwBatch.pulls - then ONE launch; wx_sw_rhs_direct): bit-identical to the two-kernel
    # form.  None = automatic: taken for float64 states at n = 8, where it is measured ahead since its face stage is spread
    # over all threads and its independent loads are issued before the first barrier (S7: 49 against 53-58 us,
    # profiles/r05_sw_s7_ab.txt); True / False: forced
    direct = None

    def _use_direct(self, dtype) -> bool:
        if self.direct is not None:
            return bool(self.direct)
        return dtype == torch.float64 and self.panel_shape is not None and self.panel_shape[3] == 64

    def _pipe_state(self, dtype):
        """The stage pipeline's ping-pong state of one dtype: slot in use, the tensor whose faces are prepared, the two
        edge-buffer sets (the second created here: setup time)."""
        if not hasattr(self, "_pipe"):
            self._pipe = {}
        st = self._pipe.get(dtype)
        if st is None:
            st = self._pipe[dtype] = {"slot": 0, "ready": None, "ex": [self.exchange_for(dtype), None], "batch": None}
            st["ex"][1] = self.new_exchange(self.edge_count * (2 if dtype.is_complex else 1))
        return st

    def reserve(self, stage: bool = True, dtype=torch.float64):
        """Allocate now what the stage pipeline needs (second interface slots, second edge-buffer set, the pipelined
        batch), so that its first call can sit inside a HIP-graph capture."""
        if not stage:
            return
        st = self._pipe_state(dtype)
        plans = self.plans_for(dtype)
        for pl in plans.values():
            pl.reserve()
        if self.batched and len(self.panels) > 1 and st["batch"] is None:
            st["batch"] = SwBatch(plans, st["ex"][0], st["ex"][1])

    def invalidate_faces(self):
        """Forget the faces the last stage prepared (call after modifying a state that stage() returned)."""
        for st in getattr(self, "_pipe", {}).values():
            st["ready"] = None

    def stage(self, Q: torch.Tensor, Y, a: float, b: float, c: float) -> torch.Tensor:
        """One explicit Runge-Kutta stage  a*Y + b*Q + c*R(Q)  on stacked states with the stage pipeline (wx_sw_stage /
        wx_sw_batch_stage): the kernel that produces the result also extrapolates it to the element faces and packs its
        edge lines, so that the NEXT call whose Q is that very tensor starts at the exchange - at the benchmark's S7 size
        the extrapolation launch is a fifth of an evaluation.  Same contract as RhsEuler3D.stage: the prepared faces are
        used only for the tensor the previous stage returned, unmodified (torch's version counter; invalidate_faces())."""
        import weakref

        np_ = len(self.panels)
        dtype = Q.dtype
        plans = self.plans_for(dtype)
        st = self._pipe_state(dtype)
        cur = st["slot"]
        ex, exn = st["ex"][cur], st["ex"][1 - cur]
        if not self.panels:   # a rank that owns no tile: the exchange of this stage, and the same slot flip as the others
            if ex.needs_comm:
                ex.start(on_compute=True)
                ex.wait()
            st["slot"] = 1 - cur
            return torch.empty_like(Q)
        Qs = Q.reshape((np_,) + tuple(self.panel_shape))
        Ys = Y.reshape((np_,) + tuple(self.panel_shape)) if Y is not None else None
        out = torch.empty_like(Qs)
        last = st["ready"][0]() if st["ready"] is not None else None
        reuse = last is Q and st["ready"][1:] == (Q.data_ptr(), Q._version, Q.numel())
        use_batch = self.batched and np_ > 1 and Q.is_contiguous() and (Y is None or Y.is_contiguous())
        if use_batch:
            if st["batch"] is None:
                st["batch"] = SwBatch(plans, st["ex"][0], st["ex"][1])
            bt = st["batch"]
            if not reuse:
                bt.extrap_pack_slot(Qs, cur)
            self._phases(ex, lambda region: bt.stage(Qs, Ys, out, a, b, c, region, cur, True))
        else:
            if not reuse:
                for i, p in enumerate(self.panels):
                    plans[p].extrap_pack_slot(Qs[i], ex.send_views(p), cur)
            self._exchange_and_launch(ex, lambda i, p, halo, region: plans[p].stage(
                Qs[i], halo, Ys[i] if Ys is not None else None, out[i], a, b, c, region, cur, exn.send_views(p), True))
        res = out.reshape(Q.shape)
        st["slot"] = 1 - cur
        st["ready"] = (weakref.ref(res), res.data_ptr(), res._version, res.numel())
        return res

    def _run(self, qs, ys, coef, dtype, zs=None):
        if (self.batched and zs is None and isinstance(qs, torch.Tensor) and len(self.panels) > 1
                and qs.is_contiguous() and qs.numel() == len(self.panels) * 3 * self.panel_shape[1] ** 2 * self.panel_shape[3]
                and (ys is None or (isinstance(ys, torch.Tensor) and ys.is_contiguous() and ys.numel() == qs.numel()))):
            return self._run_batched(qs, ys, coef)
        return super()._run(qs, ys, coef, dtype, zs)

    def _run_batched(self, q, y=None, coef=None):
        dt = q.dtype
        plans, ex = self.plans_for(dt), self.exchange_for(dt)
        if not hasattr(self, "_batches"):
            self._batches = {}
        if dt not in self._batches:
            self._batches[dt] = SwBatch(plans, ex)
        b = self._batches[dt]
        out = torch.empty_like(q)
        if self._use_direct(dt):
            if not b.pulls:
                b.extrap_pack_ring(q)
            self._phases(ex, lambda region: b.rhs_direct(q, out, region, y, coef))
            return out
        b.extrap_pack(q)
        self._phases(ex, lambda region: b.rhs(q, out, region, y, coef))
        return out
