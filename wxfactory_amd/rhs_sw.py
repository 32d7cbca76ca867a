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

    def __init__(self, plans: Dict[int, SwPlan], exchange: PanelExchange):
        self.lib = _lib.load()
        self.panels = sorted(plans)
        n = len(self.panels)
        first = plans[self.panels[0]]
        self.device, self.dtype = first.device, first.dtype
        self.stride = 3 * first.H * first.H * first.n**2
        handles = (ctypes.c_void_p * n)(*[plans[p]._h for p in self.panels])
        send = ((ctypes.c_void_p * 4) * n)()
        halo = ((ctypes.c_void_p * 4) * n)()
        for i, p in enumerate(self.panels):
            for e in range(4):
                send[i][e] = exchange.send_view(p, e).data_ptr()
                halo[i][e] = exchange.halo_view(p, e).data_ptr()
        self._keep = (plans, exchange)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.wx_sw_batch_create(ctypes.byref(self._h), handles, n, send, halo), "wx_sw_batch_create")

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
        b.extrap_pack(q)
        self._phases(ex, lambda region: b.rhs(q, out, region, y, coef))
        return out
