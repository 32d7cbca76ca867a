"""Matrix-free Jacobian-vector products through the RHS kernels.

Mirrors reference wx_factory/solvers/matvec.py:36-88: same names, argument order, epsilons and
formulas; states are torch tensors on the GPU (any of the structures PanelRhs accepts, flattened
results like the reference's `.flatten()`).
  matvec_fun(..., method="complex")  dt * Im R(Q + i eps v) / eps,  eps = sqrt(eps_f64)   (:56-61)
  matvec_fun(..., method="fd")       dt * (R(Q + eps v) - R(Q)) / eps, eps = sqrt(eps_f32) (:62-66)
  matvec_rat                         v - 1/2 dt (R(Q + eps v) - R(Q)) / eps               (:76-88)
"""
import math
from typing import Callable

import torch

EPS_COMPLEX = math.sqrt(torch.finfo(torch.float64).eps)
EPS_FD = math.sqrt(torch.finfo(torch.float32).eps)


def _can_shift(rhs_handle, Q) -> bool:
    return bool(getattr(rhs_handle, "supports_shift", False) and getattr(rhs_handle, "fused_shift", True)
                and isinstance(Q, torch.Tensor) and Q.dtype == torch.float64 and Q.is_contiguous() and rhs_handle.panels)


def matvec_fun(vec: torch.Tensor, dt: float, Q: torch.Tensor, rhs: torch.Tensor, rhs_handle: Callable,
               method: str = "complex") -> torch.Tensor:
    if method == "complex" and getattr(rhs_handle, "supports_jvp", False) and getattr(rhs_handle, "fused_jvp", True) \
            and isinstance(Q, torch.Tensor) and Q.dtype == torch.float64 and Q.is_contiguous() \
            and (rhs_handle.panels or getattr(rhs_handle, "world", 1) > 1):   # (a rank without tiles joins the exchanges)
        # dual-number kernels: (Q, eps v) formed on load, dt/eps * tangent stored - no complex arrays
        return rhs_handle.jvp(Q, vec.reshape(Q.shape).contiguous(), EPS_COMPLEX, dt / EPS_COMPLEX).flatten()
    if method == "complex":
        Qvec = torch.complex(Q, EPS_COMPLEX * vec.reshape(Q.shape))
        jac = dt * (rhs_handle(Qvec).imag / EPS_COMPLEX)
    elif _can_shift(rhs_handle, Q):
        # Q + eps v formed on load, dt/eps * (R(Q + eps v) - R(Q)) formed in the store: two launches per panel
        jac = rhs_handle.shifted_axpy(Q, vec.reshape(Q.shape).contiguous(), EPS_FD, rhs.reshape(Q.shape).contiguous(),
                                      -dt / EPS_FD, 0.0, dt / EPS_FD)
    elif getattr(rhs_handle, "supports_axpy2", False) and isinstance(Q, torch.Tensor) and Q.dtype == torch.float64:
        # fused store: dt/eps * R(Q + eps v) - dt/eps * R(Q) in the RHS launch itself
        Qvec = torch.add(Q, vec.reshape(Q.shape), alpha=EPS_FD)
        jac = rhs_handle.axpy(Qvec, None, 0.0, 0.0, dt / EPS_FD, zs=rhs.reshape(Q.shape), d=-dt / EPS_FD)
    else:
        Qvec = Q + EPS_FD * vec.reshape(Q.shape)
        jac = dt * (rhs_handle(Qvec) - rhs) / EPS_FD
    return jac.flatten()


def matvec_rat(vec: torch.Tensor, dt: float, Q: torch.Tensor, rhs: torch.Tensor, rhs_handle: Callable,
               scale: float = 1.0, out=None) -> torch.Tensor:
    """scale != 1: matvec_rat(vec / scale) * scale - what fgmres' lagged normalisation asks for (solvers/fgmres.py:172:
    `A(Z / v_norm) * v_norm`; not the same as matvec_rat(vec) for a finite-difference operator) - with the two
    scalings folded into the shift and the store coefficients of the fused kernels instead of two passes over the vector."""
    if _can_shift(rhs_handle, Q):
        v = vec.reshape(Q.shape).contiguous()
        c = 0.5 * dt / EPS_FD * scale
        return rhs_handle.shifted_axpy(Q, v, EPS_FD / scale, v, 1.0, 0.0, -c, rhs.reshape(Q.shape).contiguous(), c,
                                       out=out).flatten()
    if scale != 1.0:
        return matvec_rat(vec / scale, dt, Q, rhs, rhs_handle) * scale
    if getattr(rhs_handle, "supports_axpy2", False) and isinstance(Q, torch.Tensor) and Q.dtype == torch.float64:
        # v - dt/(2 eps) (R(Q + eps v) - R(Q)) formed in the RHS kernel's store
        v = vec.reshape(Q.shape)
        Qvec = torch.add(Q, v, alpha=EPS_FD)
        c = 0.5 * dt / EPS_FD
        return rhs_handle.axpy(Qvec, v, 1.0, 0.0, -c, zs=rhs.reshape(Q.shape), d=c).flatten()
    Qvec = Q + EPS_FD * vec.reshape(Q.shape)
    jac = dt * (rhs_handle(Qvec) - rhs) / EPS_FD
    return vec.flatten() - 0.5 * jac.flatten()


class MatvecOp:
    """solvers/matvec.py:7-28"""

    def __init__(self, matvec: Callable, dtype, shape):
        self.matvec, self.dtype, self.shape = matvec, dtype, tuple(shape)
        self.size = math.prod(self.shape)

    def __call__(self, vec):
        return self.matvec(vec)


class ComplexStepOperator:
    """v -> matvec_fun(v, dt, Q, rhs, rhs_handle, "complex") as an object, so that a solver can ask it for more than a
    product: `kiops_vector` (when the RHS offers it) builds a whole Krylov vector of KIOPS - JVP, augmented update,
    orthogonalisation, normalisation - from one host call at launch-bound sizes."""

    def __init__(self, dt: float, Q: torch.Tensor, rhs: torch.Tensor, rhs_handle: Callable, method: str = "complex"):
        self.dt, self.Q, self.rhs, self.rhs_handle, self.method = dt, Q, rhs, rhs_handle, method
        self.kiops_vector = None
        # EXACTLY linear in v: the dual-number kernels carry the tangent to first order, with no truncation term.  (A
        # finite difference has one that is quadratic in v, and the true complex step one of order eps^2 |v|^2: a solver
        # that feeds such an operator un-normalised vectors changes its results - solvers.kiops `lazy`.)
        self.linear = bool(method == "complex" and getattr(rhs_handle, "supports_jvp", False)
                           and getattr(rhs_handle, "fused_jvp", True))
        prep = getattr(rhs_handle, "jvp_prepare", None)
        if method == "complex" and prep is not None and getattr(rhs_handle, "fused_jvp", True) and isinstance(Q, torch.Tensor):
            prep(Q)   # every product of this operator linearises about Q: cache its face values once
        fn = getattr(rhs_handle, "kiops_vector_fn", None)
        if method == "complex" and fn is not None and getattr(rhs_handle, "fused_jvp", True) and Q.is_cuda:
            self.kiops_vector = fn(Q, EPS_COMPLEX, dt / EPS_COMPLEX)
        # ... and the same for a vector of PMEX (wx_euler3d_batch_pmex_vector)
        self.pmex_vector = getattr(self.kiops_vector, "pmex", None)

    def __call__(self, vec: torch.Tensor) -> torch.Tensor:
        return matvec_fun(vec, self.dt, self.Q, self.rhs, self.rhs_handle, self.method)

    def fold_ready(self) -> bool:
        """True when axpy_into(..., rows=, fix=) is available: the product runs on the prepared per-tile kernels, whose
        tangent-extrapolation launch can first correct its input vector in place (KIOPS on long vectors: the subtraction and
        the norm of the previous vector's orthogonalisation without a sweep of their own)."""
        h = self.rhs_handle
        fuses = getattr(h, "jvp_fuses_store", None)
        return bool(self.method == "complex" and fuses is not None and getattr(h, "supports_jvp", False)
                    and getattr(h, "fused_jvp", True) and getattr(h, "jvp_supports_fix", False)
                    and hasattr(h, "jvp_partials_capacity") and isinstance(self.Q, torch.Tensor) and self.Q.is_contiguous()
                    and fuses(self.Q))

    def axpy_into(self, vec: torch.Tensor, out: torch.Tensor, z: torch.Tensor, z_scale: int, z_coef: int, rows=None, fix=None):
        """out = *z_scale * (A vec) + *z_coef * z formed in the product's own store (the two coefficients: device addresses of
        one double each, z_scale 0 = 1) - KIOPS' V[j] = A V[j-1] + u a (solvers/kiops.py:170-176) without a sweep of its own.
        False (nothing done) when this product does not run on the kernels that offer the store; True when it is done;
        with `rows` (one or two contiguous vectors like out): (partials, count) - the launches have left the products
        <row, out> as `count` pairs of partial sums in the device tensor `partials` (wx_kiops_long_a_finish sums them).
        `fix` (only with rows, only when fold_ready()): dict(rows=, h=, s=, between=) - `vec` is corrected in place first,
        vec -= h[k] s[k] fix_rows[k], and between(partials_tensor, count) is called on the stream between that and the
        product (RhsEuler3D.jvp(fix=))."""
        h = self.rhs_handle
        fuses = getattr(h, "jvp_fuses_store", None)
        if (self.method != "complex" or fuses is None or not getattr(h, "supports_jvp", False) or not getattr(h, "fused_jvp", True)
                or not isinstance(self.Q, torch.Tensor) or not self.Q.is_contiguous()
                or not vec.is_contiguous() or not out.is_contiguous() or not z.is_contiguous() or not fuses(self.Q)):
            return False
        if rows and len(rows) <= 2 and all(r.is_contiguous() for r in rows) and hasattr(h, "jvp_partials_capacity"):
            part = getattr(h, "_jvp_partials_buf", None)   # (kept by the RHS object: one address from solve to solve)
            if part is None or part.device != out.device or part.numel() < h.jvp_partials_capacity():
                part = h._jvp_partials_buf = torch.empty(h.jvp_partials_capacity(), dtype=torch.float64, device=out.device)
            jfix = None
            if fix is not None:
                fpart = getattr(h, "_jvp_fix_buf", None)
                if fpart is None or fpart.device != out.device or fpart.numel() < h.jvp_fix_capacity():
                    fpart = h._jvp_fix_buf = torch.empty(h.jvp_fix_capacity(), dtype=torch.float64, device=out.device)
                jfix = dict(rows=fix["rows"], h=fix["h"], s=fix["s"], part=fpart,
                            between=lambda count: fix["between"](fpart, count))
            h.jvp(self.Q, vec.reshape(self.Q.shape), EPS_COMPLEX, self.dt / EPS_COMPLEX, out=out, z=z, z_scale=z_scale,
                  z_coef=z_coef, rows=rows, partials=part, **({"fix": jfix} if jfix is not None else {}))
            return part, h.jvp_partials_written
        if fix is not None:
            raise RuntimeError("axpy_into(fix=) needs rows= and an operator that is fold_ready()")
        h.jvp(self.Q, vec.reshape(self.Q.shape), EPS_COMPLEX, self.dt / EPS_COMPLEX, out=out, z=z, z_scale=z_scale, z_coef=z_coef)
        return True


class MatvecOpRat(MatvecOp):
    """solvers/matvec.py:71-73"""

    def __init__(self, dt, Q, rhs_vec, rhs_handle):
        super().__init__(lambda vec: matvec_rat(vec, dt, Q, rhs_vec, rhs_handle), Q.dtype, Q.shape)
        # A(vec / s) * s without passes over the vector for the scalings (used by fgmres' lagged normalisation)
        self.scaled = lambda vec, s, out=None: matvec_rat(vec, dt, Q, rhs_vec, rhs_handle, scale=s, out=out)
        # ... and a whole Krylov vector of fgmres - this operator applied with the scale read from device memory, the products, the
        # Gram-Schmidt step, the update of the rows - from one host call with no host round trip (RhsEuler3D.fgmres_vector_fn)
        fn = getattr(rhs_handle, "fgmres_vector_fn", None)
        self.fgmres_vector = fn(Q, rhs_vec, dt, EPS_FD) if fn is not None and _can_shift(rhs_handle, Q) else None


class MatvecOpBasic(MatvecOp):
    def __init__(self, dt, Q, rhs_vec, rhs_handle, method="complex"):
        super().__init__(lambda vec: matvec_fun(vec, dt, Q, rhs_vec, rhs_handle, method), Q.dtype, Q.shape)
