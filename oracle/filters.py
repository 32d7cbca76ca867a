"""TEST INFRASTRUCTURE - CPU restatement (NumPy) of the per-step filters of the explicit loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path is the HIP library and never routes through here.

Follows the reference (read as text, never copied):
  make_filter        geometry/operators.py:208-233  exponential modal filter (Warburton eqn 5.16), nodal form
  apply_filter_3d    geometry/operators.py:114-119, 257-261  ((sqrtG*Q) @ (Fx Fy Fz)) * inv_sqrtG
  check_for_nan      simulation/simulation.py:399-408
Pinned by tests/golden/filters_*.npz (values produced by the reference's own DFROperators.apply_filters).
"""
import numpy
from numpy.polynomial.legendre import legvander


def make_filter(alpha: float, order: int, cutoff: float, solution_points: numpy.ndarray) -> numpy.ndarray:
    n = len(solution_points)
    modes = numpy.arange(n) / (n - 1)
    residual = numpy.ones_like(modes)
    hi = modes > cutoff
    residual[hi] = numpy.exp(-alpha * ((modes[hi] - cutoff) / (1 - cutoff)) ** order)
    vander = legvander(solution_points, n - 1)
    return vander @ numpy.diag(residual) @ numpy.linalg.inv(vander)


def apply_filter_3d(Q: numpy.ndarray, sqrtG: numpy.ndarray, F: numpy.ndarray) -> numpy.ndarray:
    """Q (nvar, V, H, H, n^3) element-blocked, point p = (kl*n + jl)*n + il; F (n, n) nodal 1-D filter.
    The dense operator of the reference is kron(I,F)^T kron(I,F,I)^T kron(F,I)^T: F along each local axis."""
    n = F.shape[0]
    s = (sqrtG * Q).reshape(Q.shape[:-1] + (n, n, n))
    s = numpy.einsum("ic,...abc->...abi", F, s)
    s = numpy.einsum("jb,...abi->...aji", F, s)
    s = numpy.einsum("ka,...aji->...kji", F, s)
    return s.reshape(Q.shape) * (1.0 / sqrtG)


def has_nan(Q: numpy.ndarray) -> bool:
    return bool(numpy.any(numpy.isnan(Q)))
