"""Distributed vector reductions and flexible GMRES on GPU-resident vectors.

global_norm / global_dotprod mirror reference wx_factory/solvers/global_operations.py:14-36 with the
MPI allreduce replaced by torch.distributed (RCCL).  fgmres keeps the reference's signature and return
tuple (solvers/fgmres.py:97-276): restarted flexible GMRES, Givens-updated residual, same
stagnation/convergence flags.  Orthogonalisation is classical Gram-Schmidt applied twice on the
device (two small fused reductions per Krylov vector) instead of the reference's lagged 1-sync
variant - the same Krylov iterates up to rounding.
"""
import math
from time import time
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def _allreduce(t: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def global_dotprod(a: torch.Tensor, b: torch.Tensor, group=None) -> torch.Tensor:
    return _allreduce(torch.dot(a, b).reshape(1), group)[0]


def global_norm(a: torch.Tensor, group=None) -> torch.Tensor:
    if a.dim() != 1:
        raise ValueError("This function only accept a vector (1 dimension tensor)")
    return torch.sqrt(global_dotprod(a, a, group))


def global_inf_norm(a: torch.Tensor, group=None) -> torch.Tensor:
    m = a.abs().max().reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return m[0]


def _rotg(a: float, b: float):
    """solvers/fgmres.py:75-94"""
    if b == 0.0:
        return 1.0, 0.0, a
    if a == 0.0:
        return 0.0, 1.0, b
    scl = min(abs(a), abs(b))
    sigma = math.copysign(1.0, a) if abs(a) > abs(b) else math.copysign(1.0, b)
    r = sigma * (scl * math.sqrt((a / scl) ** 2 + (b / scl) ** 2))
    return a / r, b / r, r


def fgmres(A: Callable, b: torch.Tensor, x0: Optional[torch.Tensor] = None, tol: float = 1e-5, restart: int = 20,
           maxiter: Optional[int] = None, preconditioner: Optional[Callable] = None, verbose: int = 0, group=None
           ) -> Tuple[torch.Tensor, float, float, int, int, List[Tuple[float, float, float]]]:
    """Solve A x = b.  Returns (x, norm_r, norm_b, num_iter, flag, residuals) like the reference."""
    if b.numel() <= restart:
        raise ValueError("The b vector should be longer than the number of restart")
    t0 = time()
    M = preconditioner if preconditioner is not None else (lambda v: v)
    n = b.numel()
    if maxiter is None:
        maxiter = n * 10
    x = torch.zeros_like(b) if x0 is None else x0.clone()
    norm_b = float(global_norm(b, group))
    if norm_b == 0.0:
        return torch.zeros_like(b), 0.0, 0.0, 0, 0, [(0.0, time() - t0, 0.0)]
    tol_abs = tol * norm_b
    r = b - A(x)
    norm_r = float(global_norm(r, group))
    residuals = [(norm_r / norm_b, time() - t0, 0.0)]
    niter = 0
    V = torch.empty((restart + 1, n), dtype=b.dtype, device=b.device)
    Z = torch.empty((restart, n), dtype=b.dtype, device=b.device)
    for _outer in range(maxiter):
        H = [[0.0] * (restart + 1) for _ in range(restart)]  # H[j][i] = h_{i,j}
        cs, sn = [], []
        g = [0.0] * (restart + 1)
        g[0] = norm_r
        V[0] = r / norm_r
        k = 0
        for j in range(restart):
            niter += 1
            Z[j] = M(V[j])
            w = A(Z[j])
            # classical Gram-Schmidt, twice
            h = _allreduce(V[: j + 1] @ w, group)
            w = w - h @ V[: j + 1]
            h2 = _allreduce(V[: j + 1] @ w, group)
            w = w - h2 @ V[: j + 1]
            h = h + h2
            hn = float(global_norm(w, group))
            hj = h.tolist() + [hn]
            if hn != 0.0:
                V[j + 1] = w / hn
            for i in range(j):  # previous rotations
                t = cs[i] * hj[i] + sn[i] * hj[i + 1]
                hj[i + 1] = -sn[i] * hj[i] + cs[i] * hj[i + 1]
                hj[i] = t
            c, s, rr = _rotg(hj[j], hj[j + 1])
            cs.append(c)
            sn.append(s)
            hj[j], hj[j + 1] = rr, 0.0
            g[j + 1] = -s * g[j]
            g[j] = c * g[j]
            H[j][: j + 2] = hj
            k = j + 1
            norm_r = abs(g[j + 1])
            if j < restart - 1:
                residuals.append((norm_r / norm_b, time() - t0, 0.0))
                if norm_r < tol_abs:
                    break
            if hn == 0.0:
                break
        # back substitution on the k x k upper-triangular system
        y = [0.0] * k
        for i in range(k - 1, -1, -1):
            acc = g[i]
            for l in range(i + 1, k):
                acc -= H[l][i] * y[l]
            y[i] = acc / H[i][i]
        update = torch.as_tensor(y, dtype=b.dtype, device=b.device) @ Z[:k]
        x = x + update
        r = b - A(x)
        norm_r = float(global_norm(r, group))
        residuals.append((norm_r / norm_b, time() - t0, 0.0))
        if verbose > 0:
            print(f"res: {norm_r/norm_b:.2e} (iter {niter})", flush=True)
        nz = x != 0
        if bool(nz.any()):
            change = float(global_inf_norm((update[nz] / x[nz]), group))
            if change < 1e-12:
                return x, norm_r, norm_b, niter, -1, residuals
        if norm_r < tol_abs:
            return x, norm_r, norm_b, niter, 0, residuals
    return x, norm_r, norm_b, niter, (0 if norm_r < tol_abs else -1), residuals
