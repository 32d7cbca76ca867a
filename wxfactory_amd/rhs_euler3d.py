"""3-D Euler RHS on cubed-sphere panels: host-side mirror of the reference's RHS object.

Same contract as reference wx_factory/rhs/rhs.py:75-122 (`RHS.__call__`) with
`RHSDirecFluxReconstruction_mpi` (rhs/rhs_dfr.py:48-313): `rhs(Q) -> R`, same shape and dtype
as Q (float64 or complex128), collective over the ranks, result is fresh storage.  The eight
phases of the reference collapse into two HIP kernels per panel (see csrc/euler3d.hip); the
phase timers of rhs.py:88-118 are kept as four buckets (extrap+pack, exchange, interior, boundary).
"""
import ctypes
import math
import os
import weakref
from typing import Dict, Optional, Sequence

import numpy
import torch

from . import _lib
from ._lib import DfrOps, Euler3DMetric, check
from .exchange import PanelExchange  # noqa: F401
from .panel_rhs import PanelRhs, _ptr_array

_DTYPES = {torch.float64: _lib.WX_F64, torch.complex128: _lib.WX_C128}


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def column_metric_slabs(metric: Dict[str, torch.Tensor], n: int, H: int, V: int, rtol: float = 1e-9):
    """The metric arrays of a column-invariant geometry as one slab per column: {field: tensor}, or None when some array
    differs between levels by more than rtol x its largest entry.  (A real difference - topography, a deep atmosphere -
    is of order one.  What the reference's arrays show on a shallow atmosphere without topography is rounding: 1e-14 of
    the components that do not vanish, and the whole value - 1e-12 absolute at the benchmark's resolution, 2e-10 of the
    array's largest entry - of the Christoffel symbols that vanish analytically and hold second-difference noise.)
    Shapes: point fields (.., V, H, H, n^3) -> (.., H, H, n^2), taken at the lowest level of the lowest element;
    lateral interface fields (.., V, H[+2], H[+2], 2 n^2) -> (.., H[+2], H[+2], 2, n): the n values along the face's
    horizontal direction; horizontal interface fields (.., V + 2, H, H, 2 n^2) -> (.., H, H, n^2): the faces between the
    elements of a column (the two padding layers are not part of the comparison: no kernel reads them)."""
    n2 = n * n
    out = {}

    def same(full, slab_b):
        scale = float(full.abs().max())
        return scale == 0.0 or float((full - slab_b).abs().max()) <= rtol * scale

    for k in ("sqrtG", "h_contra", "christoffel", "inv_dzdeta"):
        t = metric[k]
        lead = t.shape[:-4]
        v = t.reshape(*lead, V, H, H, n, n2)
        slab = v[..., 0, :, :, 0, :]                       # (.., H, H, n2)
        if not same(v, slab[..., None, :, :, None, :]):
            return None
        out[k] = slab.contiguous()
    for k, hh in (("sqrtG_itf_i", (H, H + 2)), ("h_contra_itf_i", (H, H + 2)), ("sqrtG_itf_j", (H + 2, H)),
                  ("h_contra_itf_j", (H + 2, H))):
        t = metric[k]
        lead = t.shape[:-4]
        v = t.reshape(*lead, V, hh[0], hh[1], 2, n, n)     # (.., level, row, col, side, a = vertical index, b)
        slab = v[..., 0, :, :, :, 0, :]                    # (.., row, col, side, b)
        if not same(v, slab[..., None, :, :, :, None, :]):
            return None
        out[k] = slab.contiguous()
    for k in ("sqrtG_itf_k", "h_contra_itf_k"):
        t = metric[k]
        lead = t.shape[:-4]
        v = t.reshape(*lead, V + 2, H, H, 2, n2)[..., 1: V + 1, :, :, :, :]
        slab = v[..., 0, :, :, 0, :]                       # (.., H, H, n2)
        if not same(v, slab[..., None, :, :, None, :]):
            return None
        out[k] = slab.contiguous()
    return out


class LaunchEvents:
    """Opt-in instrumentation of the two launches of an evaluation (bench.py's live kernel timing): while `on`, every
    extrapolation / pack launch and every fused-kernel launch of Euler3DPlan and Euler3DBatch is bracketed by a pair of HIP
    events on the launch stream.  `rhs`: (start, end, region, tiles in the launch); `pack`: (start, end).  Off: one attribute
    test per launch."""

    def __init__(self):
        self.on = False
        self.rhs, self.pack = [], []

    def clear(self):
        self.rhs, self.pack = [], []

    def around(self, launch, kind, region=0, tiles=1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        if kind == "rhs":
            self.rhs.append((a, b, region, tiles))
        else:
            self.pack.append((a, b))


LAUNCH_EVENTS = LaunchEvents()


class Euler3DPlan:
    """One tile (= one cube panel).  Owns the native plan; borrows the metric tensors
    (kept alive here) exactly as the reference's pde module borrows NumPy/CuPy buffers."""

    def __init__(self, n: int, H: int, V: int, case_number: int, panel: int, ops: Dict[str, numpy.ndarray],
                 metric: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float64, dual: bool = False,
                 on_panel_edge=(True, True, True, True), column_metric=False):
        """`column_metric` (float64 and dual plans): False; True - the metric is column-invariant (a shallow atmosphere without
        topography: every metric array takes the same values on all levels, which is what the reference's arrays hold
        to rounding for config/dcmip31.ini and its own RHS benchmark), checked here, ValueError otherwise; or "auto" -
        taken when the check passes.  Whole-tile launches then read the metric as one (n x n) slab per column and field
        (`column_metric_slabs`) instead of V n of them."""
        self.lib = _lib.load()
        self.faces_epoch = 0  # bumped by every call that rewrites an interface buffer (pipeline validity)
        if dtype not in _DTYPES:
            raise TypeError(f"dtype must be float64 or complex128, not {dtype}")
        # complex128 storage can run true complex arithmetic (WX_C128) or first-order dual-number
        # arithmetic (WX_DUAL128: same complex-step JVP to O(eps^2), much cheaper); see include/wxhip.h
        self.dual = bool(dual) and dtype == torch.complex128
        wx_dtype = _lib.WX_DUAL128 if self.dual else _DTYPES[dtype]
        self.n, self.H, self.V, self.case_number, self.panel, self.dtype = n, H, V, case_number, panel, dtype
        self.shape = (5, V, H, H, n**3)
        self._ops, self._metric = ops, metric
        self.on_panel_edge = tuple(bool(x) for x in on_panel_edge)  # k x k tiles per panel: interior edges False
        self._keep = []
        o = DfrOps()
        for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter"):
            a = numpy.ascontiguousarray(ops[k], dtype=numpy.float64)
            self._keep.append(a)
            setattr(o, k, _dptr(a))
        m = Euler3DMetric()
        n2, n3 = n * n, n**3
        expect = {
            "sqrtG": (V, H, H, n3), "h_contra": (3, 3, V, H, H, n3), "christoffel": (3, 9, V, H, H, n3),
            "inv_dzdeta": (V, H, H, n3),
            "sqrtG_itf_i": (V, H, H + 2, 2 * n2), "sqrtG_itf_j": (V, H + 2, H, 2 * n2), "sqrtG_itf_k": (V + 2, H, H, 2 * n2),
            "h_contra_itf_i": (3, 3, V, H, H + 2, 2 * n2), "h_contra_itf_j": (3, 3, V, H + 2, H, 2 * n2),
            "h_contra_itf_k": (3, 3, V + 2, H, H, 2 * n2),
            "damp_coef": (V, H, H, n3), "damp_uref": (3, V, H, H, n3),
            "boundary_sn": (H * n,), "boundary_we": (H * n,),
        }
        self.device = metric["sqrtG"].device
        for k in _lib.EULER3D_METRIC_FIELDS:
            t = metric.get(k)
            if t is None:
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
            check(self.lib.wx_euler3d_plan_create_tile(ctypes.byref(self._h), n, H, V, case_number, wx_dtype, panel, flags,
                                                       ctypes.byref(o), ctypes.byref(m)), "wx_euler3d_plan_create_tile")
        self.edge_count = int(self.lib.wx_euler3d_edge_count(self._h))
        self.column_metric = False
        if column_metric:
            self.enable_column_metric(column_metric)

    def enable_column_metric(self, mode="auto", slabs=None) -> bool:
        """Give the plan the column form of its metric (see __init__): mode True - required, ValueError when the arrays differ
        between levels; "auto" - taken when the check passes.  `slabs`: the result of column_metric_slabs when the caller has
        it already (the twins of a plan share one set).  Returns whether the plan now runs the column kernels."""
        if self.column_metric or not (self.dtype == torch.float64 or self.dual):
            return self.column_metric
        if slabs is None:
            slabs = column_metric_slabs(self._metric, self.n, self.H, self.V)
        if slabs is None:
            if mode is True:
                raise ValueError("column_metric=True, but the metric arrays are not the same on all levels")
            return False
        cm = Euler3DMetric()
        for k in _lib.EULER3D_METRIC_FIELDS:
            setattr(cm, k, slabs[k].data_ptr() if k in slabs else None)
        self._keep.append(slabs)
        self._column_slabs = slabs
        check(self.lib.wx_euler3d_plan_set_column_metric(self._h, ctypes.byref(cm)), "wx_euler3d_plan_set_column_metric")
        self.column_metric = True
        return True

    axpy_two = True  # rhs_axpy takes a second array (z, d)

    @property
    def one_kernel(self) -> bool:
        """True when this plan evaluates in the low-order one-kernel form (csrc/euler3d_brick.h; include/wxhip.h)."""
        return int(self.lib.wx_euler3d_plan_one_kernel(self._h)) == 1

    def set_one_kernel(self, on: bool):
        """Setup time (before any Euler3DBatch over this plan is made): choose the two-kernel form (False) for a low-order plan."""
        check(self.lib.wx_euler3d_plan_set_one_kernel(self._h, 1 if on else 0), "wx_euler3d_plan_set_one_kernel")

    def twin(self, dtype, dual: bool = False):
        """Plan of another dtype over the same (borrowed) metric tensors."""
        t = Euler3DPlan(self.n, self.H, self.V, self.case_number, self.panel, self._ops, self._metric, dtype=dtype,
                        dual=dual, on_panel_edge=self.on_panel_edge)
        if self.column_metric:
            t.enable_column_metric("auto", slabs=self._column_slabs)   # (the same slabs: no second check, no second copy)
        # the form chosen for this plan (set_one_kernel) goes with the twin, where the twin's dtype offers it
        if (dtype == torch.float64 or dual) and self.n <= 4 and self.dtype == torch.float64 and t.one_kernel != self.one_kernel:
            t.set_one_kernel(self.one_kernel)
        return t

    def _check_q(self, q):
        if q.dtype != self.dtype or q.numel() != 5 * self.V * self.H * self.H * self.n**3 or not q.is_contiguous() \
                or q.device != self.device:
            raise ValueError(f"state must be a contiguous {self.dtype} tensor of {self.shape} on {self.device}")

    @property
    def bytes_per_point(self) -> float:
        """Compulsory HBM bytes per point of one RHS launch on this plan (after plan-time specialisation)."""
        return float(self.lib.wx_euler3d_bytes_per_point(self._h))

    def extrap_pack(self, q: torch.Tensor, send_ptrs: Optional[Sequence[int]]):
        self._check_q(q)
        self.faces_epoch += 1
        arr = _ptr_array(send_ptrs)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if LAUNCH_EVENTS.on:
            return LAUNCH_EVENTS.around(lambda: check(self.lib.wx_euler3d_extrap_pack(self._h, q.data_ptr(), arr, st),
                                                      "wx_euler3d_extrap_pack"), "pack")
        check(self.lib.wx_euler3d_extrap_pack(self._h, q.data_ptr(), arr, st), "wx_euler3d_extrap_pack")

    def rhs(self, q: torch.Tensor, halo_ptrs: Optional[Sequence[int]], out: torch.Tensor, region: int = _lib.WX_REGION_ALL):
        self._check_q(q)
        self._check_q(out)
        arr = _ptr_array(halo_ptrs)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if LAUNCH_EVENTS.on:
            return LAUNCH_EVENTS.around(lambda: check(self.lib.wx_euler3d_rhs(self._h, q.data_ptr(), arr, out.data_ptr(), region, st),
                                                      "wx_euler3d_rhs"), "rhs", region, 1)
        check(self.lib.wx_euler3d_rhs(self._h, q.data_ptr(), arr, out.data_ptr(), region, st), "wx_euler3d_rhs")

    def rhs_axpy(self, q, halo_ptrs, y, out, a: float, b: float, c: float, region: int = _lib.WX_REGION_ALL,
                 z=None, d: float = 0.0):
        """out = a*y + b*q + c*R(q) [+ d*z] in the same launch (y, z may be None)."""
        self._check_q(q)
        self._check_q(out)
        for t in (y, z):
            if t is not None:
                self._check_q(t)
        arr = _ptr_array(halo_ptrs)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_rhs_axpy2(self._h, q.data_ptr(), arr, y.data_ptr() if y is not None else None,
                                            z.data_ptr() if z is not None else None, out.data_ptr(), a, b, c, d, region,
                                            st), "wx_euler3d_rhs_axpy2")

    def shifted_extrap_pack(self, q, v, eps: float, send):
        """Phase 1-2 on the state q + eps*v formed on load (float64 plans)."""
        self._check_real(q)
        self._check_real(v)
        self.faces_epoch += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_shifted_extrap_pack(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(send), st),
              "wx_euler3d_shifted_extrap_pack")

    def shifted_rhs_axpy(self, q, v, eps: float, halo, y, out, a, b, c, region=_lib.WX_REGION_ALL, z=None, d: float = 0.0):
        """out = a*y + b*(q + eps v) + c*R(q + eps v) + d*z, the shifted state never materialised."""
        for t in (q, v, out, y, z):
            if t is not None:
                self._check_real(t)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_shifted_rhs_axpy2(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(halo),
                                                    y.data_ptr() if y is not None else None,
                                                    z.data_ptr() if z is not None else None, out.data_ptr(), a, b, c, d,
                                                    region, st), "wx_euler3d_shifted_rhs_axpy2")

    def reserve(self, what: int):
        """Setup-time allocation of the second interface slot (_lib.WX_RESERVE_STAGE) / the face-value cache of the prepared
        JVP (_lib.WX_RESERVE_JVP): wx_euler3d_plan_reserve.  The evaluation entry points never allocate; the host wrappers
        below reserve on first use, which must therefore not sit inside a stream capture - or call this before."""
        if (self.lib.wx_euler3d_plan_reserved(self._h) & what) == what:
            return
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("Euler3DPlan.reserve: this plan's buffers for the stage pipeline / prepared JVP do not exist "
                               "yet and a stream capture is in progress - call reserve() (or RhsEuler3D.reserve()) before it")
        with torch.cuda.device(self.device):
            check(self.lib.wx_euler3d_plan_reserve(self._h, what), "wx_euler3d_plan_reserve")

    def extrap_pack_slot(self, q, send, slot: int):
        self._check_q(q)
        if slot == 1:
            self.reserve(_lib.WX_RESERVE_STAGE)
        self.faces_epoch += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_extrap_pack_slot(self._h, q.data_ptr(), _ptr_array(send), slot, st),
              "wx_euler3d_extrap_pack_slot")

    def set_exp_filter(self, filter_matrix):
        """The nodal 1-D exponential filter a stage with prepare_next = 2 applies to its output."""
        F = numpy.ascontiguousarray(filter_matrix, dtype=numpy.float64)
        if F.shape != (self.n, self.n):
            raise ValueError(f"filter matrix must be {self.n} x {self.n}")
        check(self.lib.wx_euler3d_set_exp_filter(self._h, F.ctypes.data_as(ctypes.POINTER(ctypes.c_double))),
              "wx_euler3d_set_exp_filter")

    def stage(self, q, halo, y, z, out, a, b, c, d, region, itf_in: int, next_send, prepare_next, nan_flag: int = 0):
        """out = a*y + b*q + c*R(q) + d*z reading faces from slot itf_in; with prepare_next also the faces of
        `out` into the other slot / next_send (stage pipeline); prepare_next = 2 filters `out` first and raises the
        device flag at address nan_flag (0: none) on a NaN."""
        self._check_q(q)
        self._check_q(out)
        for t in (y, z):
            if t is not None:
                self._check_q(t)
        self.faces_epoch += int(bool(prepare_next))
        if prepare_next or itf_in == 1:
            self.reserve(_lib.WX_RESERVE_STAGE)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_stage(self._h, q.data_ptr(), _ptr_array(halo), y.data_ptr() if y is not None else None,
                                        z.data_ptr() if z is not None else None, out.data_ptr(), a, b, c, d, region,
                                        itf_in, _ptr_array(next_send), int(prepare_next), nan_flag or None, st),
              "wx_euler3d_stage")

    def _check_real(self, t):
        if t.dtype != torch.float64 or t.numel() != 5 * self.V * self.H * self.H * self.n**3 or not t.is_contiguous() \
                or t.device != self.device:
            raise ValueError(f"need a contiguous float64 tensor of {self.shape} on {self.device}")

    def jvp_extrap_pack(self, q, v, eps: float, send):
        """dual plans only: phase 1-2 on the dual state (q, eps*v) formed on load from two real arrays."""
        self._check_real(q)
        self._check_real(v)
        self.faces_epoch += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_jvp_extrap_pack(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(send), st),
              "wx_euler3d_jvp_extrap_pack")

    def jvp(self, q, v, eps: float, halo, out, scale: float, region: int = _lib.WX_REGION_ALL):
        """dual plans only: out (real) = scale * Im R(q + i eps v)."""
        for t in (q, v, out):
            self._check_real(t)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_jvp(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(halo), out.data_ptr(), scale,
                                      region, st), "wx_euler3d_jvp")

    def jvp_prepare(self, q, send_val):
        """dual plans: cache the face values of the linearisation state q (and pack its value edge messages)."""
        self._check_real(q)
        self.reserve(_lib.WX_RESERVE_JVP)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_jvp_prepare(self._h, q.data_ptr(), _ptr_array(send_val), st), "wx_euler3d_jvp_prepare")

    def jvp_tangent_pack(self, q, v, eps: float, send_tan, fix=None):
        """fix = (rows, h_ptr, s_ptr, partials_ptr): v is first corrected IN PLACE, v -= h[0] s[0] rows[0] [+ the second row],
        its squared norm left as jvp_workgroups() partial sums (wx_euler3d_jvp_tangent_extrap_pack_fix)."""
        self._check_real(q)
        self._check_real(v)
        self.faces_epoch += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        if fix is not None:
            rows, h_ptr, s_ptr, part_ptr = fix
            for r in rows:
                self._check_real(r)
            check(self.lib.wx_euler3d_jvp_tangent_extrap_pack_fix(
                self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(send_tan), rows[0].data_ptr(),
                rows[1].data_ptr() if len(rows) > 1 else None, h_ptr, s_ptr or None, part_ptr, st),
                "wx_euler3d_jvp_tangent_extrap_pack_fix")
            return
        check(self.lib.wx_euler3d_jvp_tangent_extrap_pack(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(send_tan), st),
              "wx_euler3d_jvp_tangent_extrap_pack")

    def jvp_workgroups(self, region: int = _lib.WX_REGION_ALL) -> int:
        """workgroups of one jvp_prepared launch over `region` (= pairs of partial products it writes when asked to)"""
        return int(self.lib.wx_euler3d_jvp_workgroups(self._h, region))

    def jvp_prepared(self, q, v, eps: float, halo_val, halo_tan, out, scale: float, region: int = _lib.WX_REGION_ALL,
                     z=None, z_scale: int = 0, z_coef: int = 0, rows=None, partials: int = 0):
        """z (a real tensor like out; z_scale / z_coef: DEVICE addresses of one double each, z_scale 0 = 1):
        out = *z_scale * (scale * Im R) + *z_coef * z formed in the product's own store; rows (one or two tensors like out)
        with `partials` (a device address, 2 * jvp_workgroups(region) doubles): the products <row, out> as partial sums"""
        for t in (q, v, out):
            self._check_real(t)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if z is not None:
            self._check_real(z)
            r0 = rows[0].data_ptr() if rows else None
            r1 = rows[1].data_ptr() if rows and len(rows) > 1 else None
            check(self.lib.wx_euler3d_jvp_prepared_axpy(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(halo_val),
                                                        _ptr_array(halo_tan), out.data_ptr(), scale, z.data_ptr(),
                                                        z_scale or None, z_coef, r0, r1, (partials or None) if rows else None,
                                                        region, st), "wx_euler3d_jvp_prepared_axpy")
            return
        check(self.lib.wx_euler3d_jvp_prepared(self._h, q.data_ptr(), v.data_ptr(), eps, _ptr_array(halo_val),
                                               _ptr_array(halo_tan), out.data_ptr(), scale, region, st),
              "wx_euler3d_jvp_prepared")

    def close(self):
        if self._h:
            self.lib.wx_euler3d_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Euler3DBatch:
    """All tiles of a rank in one launch per phase (wx_euler3d_batch_*): the edge buffers are the exchange's
    persistent send / halo slots, the states are slices of one stacked tensor."""

    def __init__(self, plans: Dict[int, "Euler3DPlan"], exchange: PanelExchange):
        self.lib = _lib.load()
        self.panels = sorted(plans)
        n = len(self.panels)
        first = plans[self.panels[0]]
        self.device, self.dtype = first.device, first.dtype
        self.stride = 5 * first.V * first.H * first.H * first.n**3
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
            check(self.lib.wx_euler3d_batch_create(ctypes.byref(self._h), handles, n, send, halo), "wx_euler3d_batch_create")

    @property
    def pulls(self) -> bool:
        """One-kernel form with every neighbour tile in the batch: no pack launch, R(Q) is one launch (wx_euler3d_batch_pulls)."""
        return int(self.lib.wx_euler3d_batch_pulls(self._h)) == 1

    def extrap_pack(self, q, v=None, eps: float = 0.0):
        """Phase 1-2 of all tiles; with v: on q + eps v (float64 batch) or on the dual state (q, eps v) formed from
        the two real arrays (dual batch)."""
        for pl in self._keep[0].values():
            pl.faces_epoch += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        launch = lambda: check(self.lib.wx_euler3d_batch_extrap_pack(self._h, q.data_ptr(), v.data_ptr() if v is not None else None,  # noqa: E731
                                                                     eps, self.stride, st), "wx_euler3d_batch_extrap_pack")
        if LAUNCH_EVENTS.on:
            return LAUNCH_EVENTS.around(launch, "pack")
        launch()

    def rhs(self, q, out, region, y=None, z=None, coef=None, v=None, eps: float = 0.0):
        st = torch.cuda.current_stream(self.device).cuda_stream
        a, b, c, d = coef if coef is not None else (0.0, 0.0, 1.0, 0.0)
        launch = lambda: check(self.lib.wx_euler3d_batch_rhs_axpy2(  # noqa: E731
            self._h, q.data_ptr(), v.data_ptr() if v is not None else None, eps, y.data_ptr() if y is not None else None,
            z.data_ptr() if z is not None else None, out.data_ptr(), self.stride, 0 if coef is None else 1, a, b, c, d, region, st),
            "wx_euler3d_batch_rhs_axpy2")
        if LAUNCH_EVENTS.on:
            return LAUNCH_EVENTS.around(launch, "rhs", region, len(self.panels))
        launch()

    def jvp(self, q, v, eps: float, out, scale: float, region):
        """dual batches: out (real) = scale * Im R(q + i eps v) for all tiles."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_batch_jvp(self._h, q.data_ptr(), v.data_ptr(), eps, out.data_ptr(), scale, self.stride,
                                            region, st), "wx_euler3d_batch_jvp")

    def fgmres_vector(self, q, rq, V, J: int, n: int, eps: float, half_dt_over_eps: float, R, T, K, ld: int, coef, vn, flag, work):
        """float64 batches on a rank that owns the whole sphere: Krylov vector of fgmres (step J of the lagged Gram-Schmidt) with
        the finite-difference Rosenbrock operator in front, one host call, no host round trip (wx_euler3d_batch_fgmres_vector)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_batch_fgmres_vector(self._h, q.data_ptr(), rq.data_ptr(), V.data_ptr(), V.stride(0), J, n, eps,
                                                      half_dt_over_eps, R.data_ptr(), T.data_ptr(), K.data_ptr(), ld,
                                                      coef.data_ptr(), vn.data_ptr(), flag.data_ptr(), work.data_ptr(), self.stride,
                                                      st), "wx_euler3d_batch_fgmres_vector")

    def kiops_vector(self, q, V, j: int, n: int, p: int, iop: int, eps: float, scale: float, uflip, hcol, aw, work):
        """dual batches on a rank that owns the whole sphere: Krylov vector j of KIOPS from one host call
        (wx_euler3d_batch_kiops_vector)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_batch_kiops_vector(self._h, q.data_ptr(), V.data_ptr(), V.stride(0), j, n, p, iop, eps, scale,
                                                     uflip.data_ptr(), hcol.data_ptr(), aw.data_ptr(), work.data_ptr(),
                                                     self.stride, st), "wx_euler3d_batch_kiops_vector")

    def pmex_vector(self, q, V, j: int, n: int, p: int, eps: float, scale: float, uflip, LT, Linv, tol: float, hcol_ptr: int,
                    own_ptr: int, aw, work, mmax: int):
        """... and Krylov vector j of PMEX (wx_euler3d_batch_pmex_vector)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_euler3d_batch_pmex_vector(self._h, q.data_ptr(), V.data_ptr(), V.stride(0), j, n, p, eps, scale,
                                                    uflip.data_ptr(), LT.data_ptr(), Linv.data_ptr(), LT.shape[1], tol,
                                                    hcol_ptr, own_ptr, aw.data_ptr(), work.data_ptr(), mmax, self.stride, st),
              "wx_euler3d_batch_pmex_vector")

    def close(self):
        if self._h:
            self.lib.wx_euler3d_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RhsEuler3D(PanelRhs):
    """R(Q) for the panels this rank owns (all six on one GPU, or one per GPU on six).

    `plans` : {panel: Euler3DPlan}.  Call with {panel: Q}, with a single tensor when the rank owns
    one panel, or with the panels stacked along a leading axis; returns the same structure of
    freshly allocated R, shaped like the input (the contract of rhs/rhs.py:75-122)."""

    supports_jvp = True
    supports_pipeline = True
    overlapped_entry = "wx_euler3d_rhs_overlapped"

    def __init__(self, plans, *args, column_metric=False, **kw):
        """column_metric: False (default: every kernel reads the metric arrays as they were handed over); "auto" - plans whose
        metric is the same on all levels to rounding (a shallow atmosphere without topography: config/dcmip31.ini, the
        reference's RHS benchmark) switch to the column form (Euler3DPlan.enable_column_metric), the others stay as they
        are; True - required of every plan."""
        super().__init__(plans, *args, **kw)
        if column_metric:
            for pl in plans.values():
                pl.enable_column_metric(column_metric)
    batched = True  # stacked states of several SMALL tiles: one launch per phase for all of them (Euler3DBatch)
    batch_max_points = 4_000_000  # per tile.  Whole E7 panels (14.7 M points) are faster launched one by one (7.07 vs 7.16 ms
    #                               per sphere, 7.01 vs 7.31 ms with the INTERIOR / BOUNDARY split); the 3.7 M-point tiles of the
    #                               24-tile layout are not (7.13 vs 7.08 ms, and 7.39 vs 7.14 ms with the split: one ring launch
    #                               for all local tiles instead of one of 928 workgroups per tile) - tools/tilebench.py

    def _run(self, qs, ys, coef, dtype, zs=None):
        np_ = len(self.panels)
        stacked = lambda t: (isinstance(t, torch.Tensor) and t.is_contiguous() and self.panel_shape is not None  # noqa: E731
                             and t.numel() == np_ * math.prod(self.panel_shape))
        if (self.batched and np_ > 1 and math.prod(self.panel_shape or (0,)) // 5 <= self.batch_max_points and stacked(qs) and (ys is None or (stacked(ys) and ys.dtype == qs.dtype))
                and (zs is None or (stacked(zs) and zs.dtype == qs.dtype)) and not self.timed):
            return self._run_batched(qs, ys, zs, coef)
        return super()._run(qs, ys, coef, dtype, zs)

    # tiles of this many points and more take the PREPARED per-tile Jacobian-vector product in a Krylov solve even where an
    # evaluation would share launches: face values cached once per solve, tangent-only extrapolation, the matrix-core JVP
    # kernel - against the batched unprepared product 0.382 / 0.374 ms at 0.46 M points per tile (equal), 0.591 / 0.628 at
    # 0.82 M, 1.25 / 1.36 at 1.8 M, 2.46 / 2.77 at 3.7 M (tools: a V = 1, 2 sweep of the E7 tile, round 4)
    jvp_prepare_min_points = 600_000

    def _small_tiles(self) -> bool:
        return (self.batched and len(self.panels) > 1 and self.panel_shape is not None and not self.timed
                and math.prod(self.panel_shape) // 5 <= self.batch_max_points)

    def _batch_for(self, key, plans, ex) -> Euler3DBatch:
        if not hasattr(self, "_batches"):
            self._batches = {}
        if key not in self._batches:
            self._batches[key] = Euler3DBatch(plans, ex)
        return self._batches[key]

    def _batched_phases(self, ex, launch):
        """launch(region) enqueues one kernel for all tiles."""
        self._phases(ex, launch)

    def _run_batched(self, q, y, z, coef):
        dt = q.dtype
        plans, ex = self.plans_for(dt), self.exchange_for(dt)
        b = self._batch_for(dt, plans, ex)
        out = torch.empty_like(q)
        b.extrap_pack(q)
        self._batched_phases(ex, lambda region: b.rhs(q, out, region, y, z, coef))
        return out

    def reserve(self, stage: bool = False, jvp: bool = False, dtype=torch.float64):
        """Allocate now what the stage pipeline (`stage`: second interface slot and edge-buffer set of `dtype`) and the
        prepared complex-step JVP (`jvp`: face-value cache, value / tangent exchanges) need, so that their FIRST call can
        already sit inside a HIP-graph capture (the evaluation entry points of the C ABI never allocate)."""
        if stage:
            for pl in self.plans_for(dtype).values():
                pl.reserve(_lib.WX_RESERVE_STAGE)
            st = self.__dict__.setdefault("_pipe", {}).setdefault(dtype, {"slot": 0, "ready": None, "ex": [self.exchange_for(dtype), None]})
            if st["ex"][1] is None:
                st["ex"][1] = self.new_exchange(self.edge_count * (2 if dtype.is_complex else 1))
        if jvp:
            for pl in self._jvp_plans().values():
                pl.reserve(_lib.WX_RESERVE_JVP)
            if getattr(self, "_ex_val", None) is None:
                self._ex_val, self._ex_tan = self.new_exchange(self.edge_count), self.new_exchange(self.edge_count)

    def set_exp_filter(self, filter_matrix):
        """Give every plan (of every dtype in use) the nodal 1-D exponential filter for filtered stages."""
        self._exp_filter = numpy.ascontiguousarray(filter_matrix, dtype=numpy.float64)
        for plans in self._plans.values():
            for pl in plans.values():
                pl.set_exp_filter(self._exp_filter)

    def stage(self, Q: torch.Tensor, Y, a: float, b: float, c: float, filtered: bool = False, nan_flag=None) -> torch.Tensor:
        """One explicit Runge-Kutta stage  a*Y + b*Q + c*R(Q)  on stacked states with the stage pipeline:
        the kernel that produces the result also extrapolates it to the element faces, so the NEXT call
        whose Q is that result starts without the extrapolation pass.  Two sets of interface / edge buffers
        alternate.  CONTRACT: the prepared faces are used only for the very tensor the previous stage returned
        (same object, still alive, same storage) while nobody has called invalidate_faces() and no other call of
        this object has rewritten an interface buffer.  Whoever modifies that tensor in between must call
        invalidate_faces() - as StepLoop does after a filter that is not fused into the stage.  torch's version
        counter is kept as a second line of defence for in-place torch operations, nothing more; a write through a
        raw pointer is invisible to it.  check_faces = True (or WXHIP_PIPE_CHECK=1) turns every reuse into a
        verification: the faces are extrapolated again and the edge messages compared with the prepared ones."""
        np_ = len(self.panels)
        dtype = Q.dtype
        fresh = dtype not in self._plans
        plans = self.plans_for(dtype)
        if filtered:
            if getattr(self, "_exp_filter", None) is None:
                raise RuntimeError("stage(filtered=True) needs set_exp_filter first")
            if fresh:
                for pl in plans.values():
                    pl.set_exp_filter(self._exp_filter)
        if not hasattr(self, "_pipe"):
            self._pipe = {}
        st = self._pipe.setdefault(dtype, {"slot": 0, "ready": None, "ex": [self.exchange_for(dtype), None]})
        if st["ex"][1] is None:
            st["ex"][1] = self.new_exchange(self.edge_count * (2 if dtype.is_complex else 1))
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
        last = st["ready"][0]() if st["ready"] is not None else None  # alive => its storage was not recycled
        epochs = lambda: tuple(plans[p].faces_epoch for p in self.panels)  # noqa: E731
        reuse = last is Q and st["ready"][1:] == (Q.data_ptr(), Q._version, Q.numel(), epochs())
        if reuse and (self.check_faces or os.environ.get("WXHIP_PIPE_CHECK") == "1"):
            prepared = ex.send_buf.clone()
            for i, p in enumerate(self.panels):
                plans[p].extrap_pack_slot(Qs[i], ex.send_views(p), cur)
            if not torch.equal(prepared, ex.send_buf):
                raise RuntimeError("stage pipeline: the prepared faces do not belong to this state - it was modified "
                                   "after the stage that produced it without a call of invalidate_faces()")
        elif not reuse:
            for i, p in enumerate(self.panels):
                plans[p].extrap_pack_slot(Qs[i], ex.send_views(p), cur)

        def launch(i, p, halo, region):
            plans[p].stage(Qs[i], halo, Ys[i] if Ys is not None else None, None, out[i], a, b, c, 0.0, region, cur,
                           exn.send_views(p), 2 if filtered else 1, nan_flag.ptr() if nan_flag is not None else 0)

        self._exchange_and_launch(ex, launch)
        res = out.reshape(Q.shape)
        st["slot"] = 1 - cur
        st["ready"] = (weakref.ref(res), res.data_ptr(), res._version, res.numel(), epochs())
        return res

    check_faces = False  # debug: verify prepared faces on every reuse (see stage)

    def invalidate_faces(self):
        """Forget the faces the last stage prepared: the next stage() extrapolates its state itself.  To be called
        by any code that modifies a state returned by stage() before handing it back."""
        for st in getattr(self, "_pipe", {}).values():
            st["ready"] = None

    supports_shift = True

    def shifted_axpy(self, Q: torch.Tensor, v: torch.Tensor, eps: float, Y, a: float, b: float, c: float, Z=None,
                     d: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a*Y + b*(Q + eps v) + c*R(Q + eps v) + d*Z for stacked real states; Q + eps v is formed inside the
        kernels (the finite-difference Jacobian products of solvers/matvec.py:62-66, 76-88 in two launches per
        panel and no pass over the state besides them)."""
        np_ = len(self.panels)
        if not self.panels:   # a rank that owns no tile only takes part in the exchange
            ex = self.exchange_for(torch.float64)
            if ex.needs_comm:
                ex.start(on_compute=True)
                ex.wait()
            return torch.empty_like(Q)
        shp = (np_,) + tuple(self.panel_shape)
        Qs, vs = Q.reshape(shp), v.reshape(shp)
        Ys = Y.reshape(shp) if Y is not None else None
        Zs = Z.reshape(shp) if Z is not None else None
        plans, ex = self.plans_for(torch.float64), self.exchange_for(torch.float64)
        if out is not None and out.is_contiguous() and out.numel() == Qs.numel() and out.dtype == Qs.dtype \
                and out.data_ptr() not in (Q.data_ptr(), v.data_ptr()):
            out = out.view(shp)   # the caller's storage (a row of a Krylov basis): no copy afterwards
        else:
            out = torch.empty_like(Qs)
        if self._small_tiles() and all(t is None or t.is_contiguous() for t in (Q, v, Y, Z)):
            bt = self._batch_for(torch.float64, plans, ex)
            bt.extrap_pack(Qs, vs, eps)
            self._batched_phases(ex, lambda region: bt.rhs(Qs, out, region, Ys, Zs, (a, b, c, d), vs, eps))
            return out.reshape(Q.shape)
        for i, p in enumerate(self.panels):
            plans[p].shifted_extrap_pack(Qs[i], vs[i], eps, ex.send_views(p))

        def launch(i, p, halo, region):
            plans[p].shifted_rhs_axpy(Qs[i], vs[i], eps, halo, Ys[i] if Ys is not None else None, out[i], a, b, c, region,
                                      Zs[i] if Zs is not None else None, d)

        self._exchange_and_launch(ex, launch)
        return out.reshape(Q.shape)

    def kiops_vector_fn(self, Q: torch.Tensor, eps: float, scale: float):
        """A callable (V, j, n, p, iop, uflip, hcol, aw, work) that builds Krylov vector j of KIOPS for the complex-step
        Jacobian at Q with one host call, or None when that shortcut does not apply (several ranks, large tiles)."""
        if not (self._small_tiles() and self.world == 1 and Q.is_contiguous() and Q.dtype == torch.float64
                and os.environ.get("WXHIP_JVP_LEAN") != "0" and os.environ.get("WXHIP_KIOPS_VECTOR", "1") != "0"):
            return None
        if "jvp" not in self._plans:
            self._plans["jvp"] = {p: pl.twin(torch.complex128, dual=True) for p, pl in self.plans.items()}
        plans = self._plans["jvp"]
        ex = self.exchange_for(torch.complex128)
        if ex.needs_comm:
            return None
        bt = self._batch_for("jvp", plans, ex)
        for pl in plans.values():
            pl.faces_epoch += 1

        def build(V, j, n, p, iop, uflip, hcol, aw, work):
            bt.kiops_vector(Q, V, j, n, p, iop, eps, scale, uflip, hcol, aw, work)

        def build_pmex(V, j, n, p, uflip, LT, Linv, tol, hcol_ptr, own_ptr, aw, work, mmax):
            bt.pmex_vector(Q, V, j, n, p, eps, scale, uflip, LT, Linv, tol, hcol_ptr, own_ptr, aw, work, mmax)

        build.pmex = build_pmex
        return build

    def fgmres_vector_fn(self, Q: torch.Tensor, Rq: torch.Tensor, dt: float, eps: float):
        """A callable (V, J, n, R, T, K, ld, coef, vn, flag, work) that builds a Krylov vector of fgmres for the finite-difference
        Rosenbrock operator v - dt/2 (R(Q + eps v) - R(Q)) / eps (solvers/matvec.py:76-88) from one host call and with no host
        round trip, or None when that does not apply (several ranks, large tiles, another dtype)."""
        if not (self._small_tiles() and self.world == 1 and isinstance(Q, torch.Tensor) and Q.is_cuda and Q.is_contiguous()
                and Q.dtype == torch.float64 and Rq.is_contiguous() and getattr(self, "fused_shift", True)
                and os.environ.get("WXHIP_FGMRES_VECTOR", "1") != "0"):
            return None
        plans, ex = self.plans_for(torch.float64), self.exchange_for(torch.float64)
        if ex.needs_comm:
            return None
        bt = self._batch_for(torch.float64, plans, ex)
        Rq = Rq.reshape(Q.shape)
        c = 0.5 * dt / eps

        def build(V, J, n, R, T, K, ld, coef, vn, flag, work):
            for pl in plans.values():
                pl.faces_epoch += 1
            bt.fgmres_vector(Q, Rq, V, J, n, eps, c, R, T, K, ld, coef, vn, flag, work)

        return build

    def jvp_fuses_store(self, Q) -> bool:
        """True when a product about Q takes the prepared per-tile kernels, whose store can form a x + b z (jvp(out=, z=))."""
        return bool(self.panels) and self._jvp_is_prepared(Q) and Q.dtype == torch.float64

    def jvp_partials_capacity(self) -> int:
        """doubles that hold the pairs of partial products of one product's launches, whichever way it is split"""
        plans = self._jvp_plans()
        return 2 * sum(max(plans[p].jvp_workgroups(_lib.WX_REGION_ALL),
                           plans[p].jvp_workgroups(_lib.WX_REGION_INTERIOR) + plans[p].jvp_workgroups(_lib.WX_REGION_BOUNDARY))
                       for p in self.panels)

    jvp_supports_fix = True   # jvp(fix=): the tangent extrapolation can correct its input in place (KIOPS' deferred stage)

    def jvp_fix_capacity(self) -> int:
        """doubles that hold the partial squared norms of one product's tangent-extrapolation launches (jvp(fix=))"""
        plans = self._jvp_plans()
        return sum(plans[p].jvp_workgroups(_lib.WX_REGION_ALL) for p in self.panels)

    def _jvp_plans(self):
        if "jvp" not in self._plans:
            self._plans["jvp"] = {p: pl.twin(torch.complex128, dual=True) for p, pl in self.plans.items()}
        return self._plans["jvp"]

    def jvp_prepare(self, Q: torch.Tensor):
        """Declare Q the linearisation state of the Jacobian-vector products to come (every matvec of one FGMRES / KIOPS
        solve): its face values are extrapolated and exchanged ONCE and cached; jvp(Q, v, ...) then extrapolates and
        exchanges only tangents (wx_euler3d_jvp_prepare).  Holds until jvp_release(), another jvp_prepare(), or a jvp()
        with a different or modified Q (then the unprepared path runs).  Large tiles only (small ones take the batched
        launches); collective like an evaluation."""
        dev = torch.device(self.device) if self.device is not None else Q.device
        can = 1   # a rank that owns no tile goes along with the others
        if self.panels:
            big = math.prod(self.panel_shape) // 5 >= self.jvp_prepare_min_points if self.panel_shape is not None else False
            can = int((big or not self._small_tiles()) and Q.device.type == dev.type and Q.dtype == torch.float64 and Q.is_contiguous())
        if self.world > 1:   # one decision for all ranks: the value exchange below is collective
            from . import reduce as _reduce

            flag = torch.tensor([can], dtype=torch.int32, device=dev)
            can = int(_reduce.allreduce(flag, self.reduce_group, "min").item())
        if not can:
            self._jvp_lin = None
            return False
        if getattr(self, "_ex_val", None) is None:
            self._ex_val, self._ex_tan = self.new_exchange(self.edge_count), self.new_exchange(self.edge_count)
        ex = self._ex_val
        if self.panels:
            Qs = Q.reshape((len(self.panels),) + tuple(self.panel_shape))
            plans = self._jvp_plans()
            for i, p in enumerate(self.panels):
                plans[p].jvp_prepare(Qs[i], ex.send_views(p))
        ex.start(on_compute=True)
        ex.wait()
        self._jvp_lin = (weakref.ref(Q), Q.data_ptr(), Q._version) if self.panels else ("idle",)
        return True

    def jvp_release(self):
        self._jvp_lin = None

    def _jvp_is_prepared(self, Q) -> bool:
        lin = getattr(self, "_jvp_lin", None)
        if lin is None:
            return False
        if lin[0] == "idle":
            return True
        return lin[0]() is Q and lin[1:] == (Q.data_ptr(), Q._version)

    def jvp(self, Q: torch.Tensor, v: torch.Tensor, eps: float, scale: float, out=None, z=None, z_scale: int = 0,
            z_coef: int = 0, rows=None, partials=None, fix=None) -> torch.Tensor:
        """scale * Im R(Q + i eps v) for stacked real Q, v -> real tensor shaped like Q.  The dual state is
        formed inside the kernels and only the tangent is stored: the complex-step JVP of
        solvers/matvec.py:56-61 without a complex array in HBM.
        PREPARED products only (the caller checks jvp_fuses_store): `out` = a contiguous tensor of Q's size to write into;
        `z` (like out) with the device addresses z_scale / z_coef: out = *z_scale * product + *z_coef * z in the same store;
        `rows` (one or two tensors like out) with `partials` (a float64 device tensor of jvp_partials_capacity() doubles): the
        launches also leave the products <row, out> as pairs of partial sums, self.jvp_partials_written of them.
        `fix` = dict(rows=[one or two tensors like v], h=device address of their coefficients, s=device address of the rows'
        scales or 0, part=a float64 device tensor of jvp_fix_capacity() doubles, between=callable(count)): v - a contiguous
        tensor the caller owns - is first CORRECTED IN PLACE by the tangent-extrapolation launches (v -= h[k] s[k] rows[k]),
        which leave |v|^2 as `count` partial sums in `part`; between(count) then runs on the same stream (the caller completes
        the norm there: KIOPS' deferred orthogonalisation stage) before the product's own launches."""
        np_ = len(self.panels)
        prepared = self._jvp_is_prepared(Q)
        if self.world > 1:
            # Several ranks: which exchange a product uses (tangents only, or dual faces) must be the same decision on
            # every rank, and a rank cannot see that ANOTHER rank's state was modified.  So the declaration is binding:
            # between jvp_prepare(Q) and jvp_release() - both collective - every product linearises about that Q, and a
            # different or modified state is an error here instead of a silent fall-back (which would deadlock the ranks)
            declared = getattr(self, "_jvp_lin", None) is not None
            if declared and not prepared:
                raise RuntimeError("jvp: the state differs from the one declared by jvp_prepare(); over several ranks call "
                                   "jvp_release() or jvp_prepare(Q) again on all ranks first")
            prepared = declared
        if not self.panels:   # a rank that owns no tile only takes part in the exchange of the product
            ex = self._ex_tan if prepared else self.exchange_for(torch.complex128)
            if ex.needs_comm:
                ex.start(on_compute=True)
                ex.wait()
            return torch.empty_like(Q)
        Qs = Q.reshape((np_,) + tuple(self.panel_shape))
        vs = v.reshape((np_,) + tuple(self.panel_shape))
        plans = self._jvp_plans()
        if prepared and not v.is_contiguous():
            v = v.contiguous()
            vs = v.reshape((np_,) + tuple(self.panel_shape))
        if prepared:
            exv, ext = self._ex_val, self._ex_tan
            out = torch.empty_like(Qs) if out is None else out.reshape(Qs.shape)
            zs = None if z is None else z.reshape(Qs.shape)
            rs = None if not rows or zs is None else [r.reshape(Qs.shape) for r in rows]
            written = [0]

            def extra(i, p, region):
                if zs is None:
                    return ()   # (the plain call: stand-in plans of the CPU tests know no more)
                if rs is None:
                    return (zs[i], z_scale, z_coef)
                at = partials.data_ptr() + 16 * written[0]
                written[0] += plans[p].jvp_workgroups(region)
                assert 2 * written[0] <= partials.numel()
                return (zs[i], z_scale, z_coef, [r[i] for r in rs], at)

            if fix is not None:
                frows = [r.reshape(Qs.shape) for r in fix["rows"]]
                at = 0
                for i, p in enumerate(self.panels):
                    plans[p].jvp_tangent_pack(Qs[i], vs[i], eps, ext.send_views(p),
                                              ([r[i] for r in frows], fix["h"], fix["s"], fix["part"].data_ptr() + 8 * at))
                    at += plans[p].jvp_workgroups(_lib.WX_REGION_ALL)
                assert at <= fix["part"].numel()
                fix["between"](at)
            else:
                for i, p in enumerate(self.panels):
                    plans[p].jvp_tangent_pack(Qs[i], vs[i], eps, ext.send_views(p))
            self._exchange_and_launch(ext, lambda i, p, halo, region: plans[p].jvp_prepared(
                Qs[i], vs[i], eps, exv.halo_views(p) if halo is not None else None, halo, out[i], scale, region,
                *extra(i, p, region)))
            self.jvp_partials_written = written[0]
            return out.reshape(Q.shape)
        if out is not None or z is not None or fix is not None:
            raise RuntimeError("jvp(out=, z=, fix=): only the prepared product stores into a caller's buffer (jvp_fuses_store)")
        ex = self.exchange_for(torch.complex128)
        out = torch.empty_like(Qs)
        if self._small_tiles() and Q.is_contiguous() and v.is_contiguous() and Q.dtype == torch.float64 \
                and os.environ.get("WXHIP_JVP_LEAN") != "0":  # (the batch always runs the JVP kernel)
            bt = self._batch_for("jvp", plans, ex)
            bt.extrap_pack(Qs, vs, eps)
            self._batched_phases(ex, lambda region: bt.jvp(Qs, vs, eps, out, scale, region))
            return out.reshape(Q.shape)
        for i, p in enumerate(self.panels):
            plans[p].jvp_extrap_pack(Qs[i], vs[i], eps, ex.send_views(p))
        self._exchange_and_launch(ex, lambda i, p, halo, region: plans[p].jvp(Qs[i], vs[i], eps, halo, out[i], scale, region))
        return out.reshape(Q.shape)
