"""Per-step filters and the NaN check of the explicit time loop, on the device.

Host mirror of the reference pieces `Simulation.step` runs after every integrator step
(simulation/simulation.py:147-155):
  * DFROperators.make_filter / apply_filter_3d / apply_filters   geometry/operators.py:101-119, 208-261
  * Simulation._check_for_nan                                     simulation/simulation.py:399-408
The arithmetic runs in libwxhip.so (csrc/filters.hip); there is no CPU fallback.
"""
import ctypes
from typing import Optional, Sequence

import numpy
import torch

from . import _lib
from ._lib import check

_DT = {torch.float64: _lib.WX_F64, torch.complex128: _lib.WX_C128}


def make_filter(alpha: float, order: int, cutoff: float, solution_points) -> numpy.ndarray:
    """Nodal form V diag(sigma) V^-1 of the exponential modal filter (Warburton eqn 5.16):
    sigma_m = exp(-alpha ((m/(n-1) - cutoff) / (1 - cutoff))^order) above the cutoff, 1 below
    (operators.py:208-233).  Setup-time, n x n."""
    pts = numpy.asarray(solution_points, dtype=numpy.float64)
    n = len(pts)
    if n < 2:
        raise ValueError("the 3-D filter needs degree > 1")  # operators.py:105-108 disables it instead
    modes = numpy.arange(n) / (n - 1)
    sigma = numpy.ones(n)
    hi = modes > cutoff
    sigma[hi] = numpy.exp(-alpha * ((modes[hi] - cutoff) / (1.0 - cutoff)) ** order)
    vander = numpy.polynomial.legendre.legvander(pts, n - 1)
    return vander @ numpy.diag(sigma) @ numpy.linalg.inv(vander)


class NanFlag:
    """Device-side NaN flag shared by the filter kernel and wx_check_nan; `raise_if_set` is the
    collective part of Simulation._check_for_nan (Allreduce MAX, then ValueError("NaN") on every rank)."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.group = group

    def ptr(self) -> int:
        return self.flag.data_ptr()

    def check(self, Q: torch.Tensor):
        """Scan a state that no filter has just scanned."""
        if Q.dtype not in _DT or not Q.is_contiguous() or Q.device != self.device:
            raise TypeError("state must be a contiguous float64/complex128 tensor on the flag's device")
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.load().wx_check_nan(Q.data_ptr(), Q.numel(), _DT[Q.dtype], self.ptr(), st), "wx_check_nan")

    def raise_if_set(self):
        """simulation.py:399-408: Allreduce MAX of the flag over `group` (a torch.distributed group, or the library's
        communicator - reduce.py), then ValueError("NaN") on every rank."""
        from . import reduce as _reduce

        bad = int(_reduce.allreduce(self.flag, self.group, "max").item())
        self.flag.zero_()
        if bad > 0:
            raise ValueError("NaN")


class ExpFilter3D:
    """`Q = operators.apply_filters(Q, geom, metric, dt)` for CubedSphere3D (the exponential filter is the only
    filter that geometry has, operators.py:101-119).  `sqrtG[i]` is metric.sqrtG_new of the i-th local
    panel/tile, element-blocked (V, H, H, n^3) on the device."""

    def __init__(self, filter_matrix: numpy.ndarray, sqrtG: Sequence[torch.Tensor], nan_flag: Optional[NanFlag] = None):
        self.lib = _lib.load()
        F = numpy.ascontiguousarray(filter_matrix, dtype=numpy.float64)
        if F.ndim != 2 or F.shape[0] != F.shape[1]:
            raise ValueError("filter matrix must be square (n x n)")
        self.n = F.shape[0]
        self.matrix = F
        self.sqrtG = [s.contiguous() for s in sqrtG]
        for s in self.sqrtG:
            if s.dtype != torch.float64 or not s.is_cuda or s.numel() % self.n**3:
                raise TypeError("sqrtG must be float64 device tensors of whole elements")
        self.nan_flag = nan_flag
        # small tiles (launch-bound): one launch for all of them, on a stacked copy of sqrtG
        self.sqrtG_all = None
        if len(self.sqrtG) > 1 and sum(s.numel() for s in self.sqrtG) <= 12_000_000 \
                and all(s.numel() == self.sqrtG[0].numel() for s in self.sqrtG):
            self.sqrtG_all = torch.stack([s.reshape(-1) for s in self.sqrtG]).contiguous()
        self._h = ctypes.c_void_p()
        check(self.lib.wx_expfilter_create(ctypes.byref(self._h), self.n,
                                           F.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), "wx_expfilter_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.lib.wx_expfilter_destroy(h)

    def __call__(self, Q: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Q: stacked local panels (len(sqrtG), nvar, V, H, H, n^3) (or one panel without the leading axis);
        returns the filtered state in fresh storage, or in `out` (which may be Q itself)."""
        if Q.dtype not in _DT or not Q.is_contiguous():
            raise TypeError("state must be a contiguous float64/complex128 tensor")
        np_ = len(self.sqrtG)
        per = Q.numel() // np_
        if per * np_ != Q.numel():
            raise ValueError("state does not hold a whole number of panels")
        if out is None:
            out = torch.empty_like(Q)
        elif out.shape != Q.shape or out.dtype != Q.dtype or not out.is_contiguous():
            raise TypeError("out must match the state")
        qf, of = Q.reshape(np_, per), out.reshape(np_, per)
        st = torch.cuda.current_stream(Q.device).cuda_stream
        flag = self.nan_flag.ptr() if self.nan_flag is not None else None
        if self.sqrtG_all is not None and per % self.sqrtG[0].numel() == 0:
            sg0 = self.sqrtG[0]
            check(self.lib.wx_expfilter_apply_stacked(self._h, qf.data_ptr(), of.data_ptr(), self.sqrtG_all.data_ptr(),
                                                      per // sg0.numel(), sg0.numel() // self.n**3, np_, _DT[Q.dtype], flag,
                                                      st), "wx_expfilter_apply_stacked")
            torch.autograd.graph.increment_version(out)
            return out
        for i, sg in enumerate(self.sqrtG):
            nvar = per // sg.numel()
            if nvar * sg.numel() != per:
                raise ValueError("state and sqrtG sizes do not match")
            check(self.lib.wx_expfilter_apply(self._h, qf[i].data_ptr(), of[i].data_ptr(), sg.data_ptr(), nvar,
                                              sg.numel() // self.n**3, _DT[Q.dtype], flag, st), "wx_expfilter_apply")
        # the kernel wrote through a raw pointer: tell torch, so that version-based caches (the stage
        # pipeline's "faces of this tensor are ready" check) see an in-place filter
        torch.autograd.graph.increment_version(out)
        return out


def sponge_2d(Q: torch.Tensor, beta: torch.Tensor, dt: float, idx_rho_w: int = 2) -> torch.Tensor:
    """Cartesian2D Rayleigh sponge of apply_filters (operators.py:242-253): rho_w *= 1/(1 + beta dt), in place.
    (The reference's own loop unpacks `geom.X1.shape` into two values and raises on the element-blocked
    layout, so this follows its formula and cannot be pinned by running it.)"""
    row = Q[idx_rho_w]
    if not row.is_contiguous() or beta.numel() != row.numel() or beta.dtype != torch.float64:
        raise TypeError("beta must be float64 with the shape of one field")
    st = torch.cuda.current_stream(Q.device).cuda_stream
    check(_lib.load().wx_cart2d_sponge(row.data_ptr(), beta.contiguous().data_ptr(), float(dt), row.numel(),
                                       _DT[Q.dtype], st), "wx_cart2d_sponge")
    torch.autograd.graph.increment_version(Q)
    return Q


def sponge_profile(X3: numpy.ndarray, z1: float, zscale: float, tscale: float) -> numpy.ndarray:
    """beta of operators.py:122-139: (1/tscale) sin^2(pi/2 (z - zs)/(z1 - zs)) above zs = z1 - zscale."""
    zs = z1 - zscale
    beta = numpy.zeros_like(X3)
    m = X3 >= zs
    beta[m] = (1.0 / tscale) * numpy.sin((0.5 * numpy.pi) * (X3[m] - zs) / (z1 - zs)) ** 2
    return beta
