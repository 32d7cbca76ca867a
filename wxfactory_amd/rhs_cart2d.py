"""2-D Cartesian Euler RHS (the reference's plumbing case) as one HIP launch.

Same contract as `RHSDirecFluxReconstruction` with `PDEEulerCartesian`
(reference rhs/rhs_dfr.py:8-45, pde/pde_euler_cartesian.py:8-48): rhs(Q) for Q of shape
(4, nz, nx, n^2), float64 or complex128; fresh storage.
"""
import ctypes

import numpy
import torch

from . import _lib
from ._lib import DfrOps, check

_DTYPES = {torch.float64: _lib.WX_F64, torch.complex128: _lib.WX_C128}


class RhsCart2D:
    """`dtype` is the plan built at construction; a state of the other dtype (the complex step of matvec_fun,
    solvers/matvec.py:56-61, on a float64 run) gets its plan on first use, as the reference's RHS takes either."""

    def __init__(self, n, num_elem_x1, num_elem_x3, dx1, dx3, ops, device, dtype=torch.float64):
        self.lib = _lib.load()
        self.n, self.nx, self.nz, self.dtype, self.device = n, num_elem_x1, num_elem_x3, dtype, torch.device(device)
        self.dx1, self.dx3 = float(dx1), float(dx3)
        self.shape = (4, num_elem_x3, num_elem_x1, n * n)
        self._ops = DfrOps()
        self._keep = []
        for k in ("extrap_neg", "extrap_pos", "diff_solpt", "correction", "highfilter"):
            a = numpy.ascontiguousarray(ops[k], dtype=numpy.float64)
            self._keep.append(a)
            setattr(self._ops, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        self._plans = {}
        self._plan(dtype)

    def _plan(self, dtype):
        h = self._plans.get(dtype)
        if h is None:
            if dtype not in _DTYPES:
                raise TypeError(f"dtype must be float64 or complex128, not {dtype}")
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                check(self.lib.wx_cart2d_plan_create(ctypes.byref(h), self.n, self.nx, self.nz, self.dx1, self.dx3,
                                                     _DTYPES[dtype], ctypes.byref(self._ops)), "wx_cart2d_plan_create")
            self._plans[dtype] = h
        return h

    def __call__(self, q: torch.Tensor) -> torch.Tensor:
        if q.dtype not in _DTYPES or q.numel() != 4 * self.nz * self.nx * self.n**2 or not q.is_contiguous() \
                or q.device != self.device:
            raise ValueError(f"state must be a contiguous float64 or complex128 tensor of {self.shape} on {self.device}")
        out = torch.empty_like(q)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.wx_cart2d_rhs(self._plan(q.dtype), q.data_ptr(), out.data_ptr(), st), "wx_cart2d_rhs")
        return out

    full = __call__

    def close(self):
        for h in self._plans.values():
            if h:
                self.lib.wx_cart2d_plan_destroy(h)
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
