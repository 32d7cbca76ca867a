"""Distributed vector reductions and flexible GMRES on GPU-resident vectors.

global_norm / global_dotprod mirror reference wx_factory/solvers/global_operations.py:14-36 with the
MPI allreduce replaced by torch.distributed (RCCL).  fgmres keeps the reference's signature and return
tuple (solvers/fgmres.py:97-276): restarted flexible GMRES, Givens-updated residual, same
stagnation/convergence flags.  Orthogonalisation is classical Gram-Schmidt on the device with a
second pass only after heavy cancellation (one sweep over the basis for the dot products, one for
the update, one host read per Krylov vector) instead of the reference's lagged 1-sync variant -
the same Krylov iterates up to rounding.
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


class _Basis:
    """Gram-Schmidt passes over the rows of a basis matrix V (m, n) against a vector w: on the GPU the two
    single-pass kernels of csrc/krylov.hip (no temporaries of Krylov-vector size), elsewhere torch expressions."""

    def __init__(self, V: torch.Tensor):
        self.V = V
        self.gpu = V.is_cuda and V.dtype == torch.float64 and V.is_contiguous()
        if self.gpu:
            from . import _lib

            self.lib = _lib.load()
            self.check = _lib.check
            self.work = torch.empty(int(self.lib.wx_multi_dot_workspace(V.shape[0])), dtype=torch.float64, device=V.device)

    def dots(self, lo: int, hi: int, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """<V[k, :len(w)], w> for lo <= k < hi (device tensor; written to `out` when given)."""
        V = self.V
        if not (self.gpu and w.is_contiguous() and (out is None or out.is_contiguous())):
            Vw = V[lo:hi] if w.numel() == V.shape[1] else V[lo:hi, : w.numel()]
            return torch.mv(Vw, w, out=out) if out is not None else Vw @ w
        if out is None:
            out = torch.empty(hi - lo, dtype=torch.float64, device=V.device)
        st = torch.cuda.current_stream(V.device).cuda_stream
        self.check(self.lib.wx_multi_dot(V[lo].data_ptr(), V.stride(0), hi - lo, w.data_ptr(), w.numel(), out.data_ptr(),
                                         self.work.data_ptr(), st), "wx_multi_dot")
        return out

    def subtract(self, w: torch.Tensor, lo: int, hi: int, h: torch.Tensor) -> torch.Tensor:
        """w -= sum_k h[k - lo] V[k], in place."""
        V = self.V
        if not (self.gpu and w.is_contiguous() and h.is_cuda):
            w -= h.to(w.device, w.dtype) @ V[lo:hi]
            return w
        st = torch.cuda.current_stream(V.device).cuda_stream
        self.check(self.lib.wx_multi_axpy(w.data_ptr(), V[lo].data_ptr(), V.stride(0), hi - lo, h.contiguous().data_ptr(),
                                          w.numel(), st), "wx_multi_axpy")
        return w


_REORTH = 0.1  # fgmres: re-orthogonalise when |w - V V^T w| < _REORTH |w| (orthogonality kept to ~1e-15 / _REORTH)


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
    Z = torch.empty((restart, n), dtype=b.dtype, device=b.device) if preconditioner is not None else V  # Z[j] = V[j]
    basis = _Basis(V)
    for _outer in range(maxiter):
        H = [[0.0] * (restart + 1) for _ in range(restart)]  # H[j][i] = h_{i,j}
        cs, sn = [], []
        g = [0.0] * (restart + 1)
        g[0] = norm_r
        V[0] = r / norm_r
        k = 0
        for j in range(restart):
            niter += 1
            if preconditioner is not None:
                Z[j] = M(V[j])
            w = A(Z[j])
            # classical Gram-Schmidt, one sweep for the dots and one for the update; a second pass only when the
            # first one cancelled more than a factor 1/_REORTH of w (then its rounding errors, relative to what is
            # left, threaten the orthogonality of the basis: Daniel, Gragg, Kaufman & Stewart 1976).  The decision
            # uses all-reduced numbers, so every rank takes the same one.
            ww = global_dotprod(w, w, group)
            h = _allreduce(basis.dots(0, j + 1, w), group)
            w = basis.subtract(w, 0, j + 1, h)
            hn2 = global_dotprod(w, w, group)
            torch.div(w, torch.sqrt(hn2), out=V[j + 1])  # (enqueued before the host read below; void if hn == 0)
            hj = torch.cat((h, ww.reshape(1), hn2.reshape(1))).tolist()  # the iteration's one host read
            hn2_h, ww_h = hj.pop(), hj.pop()
            if hn2_h < _REORTH * _REORTH * ww_h:
                h2 = _allreduce(basis.dots(0, j + 1, w), group)
                w = basis.subtract(w, 0, j + 1, h2)
                hn2 = global_dotprod(w, w, group)
                torch.div(w, torch.sqrt(hn2), out=V[j + 1])
                h2 = torch.cat((h2, hn2.reshape(1))).tolist()
                hn2_h = h2.pop()
                hj = [a + b for a, b in zip(hj, h2)]
            hn = math.sqrt(hn2_h)
            hj.append(hn)
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
        x += update
        r = b - A(x)
        norm_r = float(global_norm(r, group))
        residuals.append((norm_r / norm_b, time() - t0, 0.0))
        if verbose > 0:
            print(f"res: {norm_r/norm_b:.2e} (iter {niter})", flush=True)
        # largest relative change of a non-zero component (fgmres.py:263-268), without index arrays
        update.div_(torch.where(x != 0, x, torch.full_like(x, math.inf)))
        if bool((x != 0).any()):
            change = float(global_inf_norm(update, group))
            if change < 1e-12:
                return x, norm_r, norm_b, niter, -1, residuals
        if norm_r < tol_abs:
            return x, norm_r, norm_b, niter, 0, residuals
    return x, norm_r, norm_b, niter, (0 if norm_r < tol_abs else -1), residuals


_blas_threads = None


def _expm(M):
    """scipy.linalg.expm of a small dense matrix with the BLAS thread pool held to one thread: these matrices are
    at most 129 x 129, and waking a many-thread pool for them costs up to 1000x the computation on a busy host."""
    global _blas_threads
    from scipy.linalg import expm

    if _blas_threads is None:
        try:
            from threadpoolctl import ThreadpoolController

            _blas_threads = ThreadpoolController()
        except ImportError:  # pragma: no cover
            _blas_threads = False
    if not _blas_threads:
        return expm(M)
    with _blas_threads.limit(limits=1, user_api="blas"):
        return expm(M)


def _log(x: float) -> float:
    """numpy.log on a float: -inf at 0, nan below, inf at inf (math.log raises instead; the reference's adaptivity
    formulas rely on the IEEE behaviour when an error estimate under- or overflows)."""
    if x != x or x < 0.0:
        return math.nan
    if x == 0.0:
        return -math.inf
    return math.inf if x == math.inf else math.log(x)


def _ceil_clamped(x: float, lo: float, hi: float) -> int:
    """int(ceil(x)) limited to [lo, hi]; +inf goes to hi, -inf and nan to lo: what numpy.ceil followed by the
    reference's max(lo, min(x, hi)) with Python's builtins yields (solvers/kiops.py:275-276)."""
    if x == math.inf:
        return int(hi)
    if x != x or x == -math.inf:
        return int(lo)
    return int(min(hi, max(lo, math.ceil(x))))


def kiops(tau_out, A: Callable, u: torch.Tensor, tol: float = 1e-7, m_init: int = 10, mmin: int = 10, mmax: int = 128,
          iop: int = 2, task1: bool = False, group=None):
    """w(i) = sum_k phi_k(tau_i A) u[k]  by the adaptive Krylov method with incomplete orthogonalisation.

    Same signature, adaptivity rules and `stats` tuple as reference wx_factory/solvers/kiops.py:10-347
    (KIOPS: Gaudreault, Rainwater, Tokman, J. Comput. Phys. 2018; after phipm, Niesen & Wright 2012).
    Layout here: the Krylov basis (the n-long parts and the p augmented components, rows of n+p) and the
    Hessenberg matrix (transposed: a column's entries are contiguous) live on the GPU; building a Krylov
    vector costs one matvec (= one RHS evaluation through `A`), two fused reductions (iop dot products,
    norm) and a few vector kernels, with NO host synchronisation: the host reads the new Hessenberg columns
    once per pass of m vectors and detects a happy breakdown then (vectors built past it are discarded).
    With the vectors split over ranks the augmented components are replicated and enter the products once,
    after the all-reduce.  The (m+1)x(m+1) matrix exponential runs on the host (scipy), as in the reference.
    Returns (w, (steps, rejected, krylov_steps, exps, error_estimate, last_m)).
    """
    import numpy as np

    dev, dtype = u.device, u.dtype
    tau_out = [float(t) for t in tau_out]
    ppo, n = u.shape
    p = ppo - 1
    if p == 0:
        p = 1
        u = torch.cat((u, torch.zeros((1, n), dtype=dtype, device=dev)))
    m = max(mmin, min(m_init, mmax))
    Vd = torch.empty((mmax + 1, n + p), dtype=dtype, device=dev)  # (every row is written before it is read)
    basis = _Basis(Vd)
    Ht = torch.empty((mmax + 1, mmax + 1), dtype=dtype, device=dev)  # Ht[c, r] = H[r, c], written entries only
    nrm2 = torch.empty(1, dtype=dtype, device=dev)
    H = np.zeros((mmax + 1, mmax + 1))
    split = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def products(lo: int, hi: int, j: int, out: torch.Tensor):
        """out[k - lo] = <V[k], V[j]> over the n + p components, lo <= k < hi"""
        if not split:
            return basis.dots(lo, hi, Vd[j], out=out)
        t = _allreduce(basis.dots(lo, hi, Vd[j, :n]), group)
        return torch.addmv(t, Vd[lo:hi, n:], Vd[j, n:], out=out)

    step = krystep = ireject = reject = exps = 0
    sgn = math.copysign(1.0, tau_out[-1])
    tau_now, tau_end = 0.0, abs(tau_out[-1])
    happy = False
    j = 0
    conv = 0.0
    num_steps = len(tau_out)
    w = torch.zeros((num_steps, n), dtype=dtype, device=dev)
    w[0] = u[0]
    normU = float(_allreduce(u[1:].abs().sum(dim=1), group).max()) if ppo > 1 else 0.0
    if ppo > 1 and normU > 0:
        ex = math.ceil(math.log2(normU))
        nu, mu = 2.0 ** (-ex), 2.0 ** ex
    else:
        nu = mu = 1.0
    u_flip_t = (nu * torch.flipud(u[1:])).t()
    shift = torch.diag(torch.ones(p - 1, dtype=dtype, device=dev), 1)
    tau = tau_end
    gamma, gamma_mmax = (0.2, 0.1) if tau_end > 1 else (0.9, 0.6)
    delta = 1.4
    oldm, oldtau, omega = -1, math.nan, math.nan
    orderold = kestold = True
    order, kest = 1.0, 2.0
    l = 0
    beta = 1.0
    while tau_now < tau_end:
        if j == 0:
            va0 = np.zeros(p)
            for k in range(p - 1):
                i = p - k + 1
                va0[k] = (tau_now ** i) / math.factorial(i) * mu
            va0[p - 1] = mu
            Vd[0, :n] = w[l]
            Vd[0, n:] = torch.as_tensor(va0, dtype=dtype, device=dev)
            beta = math.sqrt(float(global_dotprod(Vd[0, :n], Vd[0, :n], group)) + float(va0 @ va0))
            Vd[0] /= beta
        j0 = j
        while j < m:
            j += 1
            torch.addmv(A(Vd[j - 1, :n]), u_flip_t, Vd[j - 1, n:], out=Vd[j, :n])
            torch.mv(shift, Vd[j - 1, n:], out=Vd[j, n:])  # augmented components: up by one, zero at the end
            ilow = max(0, j - iop)
            hcol = Ht[j - 1]  # column j-1 of H: entries ilow..j-1, then the norm at j
            products(ilow, j, j, hcol[ilow:j])
            basis.subtract(Vd[j], ilow, j, hcol[ilow:j])
            products(j, j + 1, j, nrm2)
            torch.sqrt(nrm2, out=hcol[j : j + 1])
            Vd[j] /= hcol[j]
        if j > j0:
            Hh = Ht[j0:j, : j + 1].cpu().numpy()  # the one synchronisation of the pass
            for c in range(j0, j):
                il = max(0, c + 1 - iop)
                H[il : c + 1, c] = Hh[c - j0, il : c + 1]
                if Hh[c - j0, c + 1] < tol:  # happy breakdown at vector c+1: the rest of the pass is void
                    happy = True
                    j = c + 1
                    break
                H[c + 1, c] = Hh[c - j0, c + 1]
                krystep += 1
        H[0, j] = 1.0
        nrm = H[j, j - 1]
        H[j, j - 1] = 0.0
        F = _expm(sgn * tau * H[: j + 1, : j + 1])
        exps += 1
        H[j, j - 1] = nrm
        if happy:
            omega = 0.0
            err = 0.0
            tau_new = min(tau_end - (tau_now + tau), tau)
            m_new = m
            happy = False
        else:
            err = abs(beta * nrm * F[j - 1, j])
            oldomega = omega
            omega = tau_end * err / (tau * tol)
            if m == oldm and tau != oldtau and ireject >= 1:
                o = _log(omega / oldomega) / _log(tau / oldtau) if oldomega > 0 and tau != oldtau else math.nan
                order = max(1.0, o) if o == o and o != math.inf else 1.0
                orderold = False
            elif orderold or ireject == 0:
                orderold = True
                order = j / 4
            else:
                orderold = True
            if m != oldm and tau == oldtau and ireject >= 1:
                kest = max(1.1, (omega / oldomega) ** (1 / (oldm - m)))
                kestold = False
            elif kestold or ireject == 0:
                kestold = True
                kest = 2
            else:
                kestold = True
            remaining = tau_end - tau_now if omega > delta else tau_end - (tau_now + tau)
            same_tau = min(remaining, tau)
            tau_opt = tau * (gamma / omega) ** (1 / order)
            tau_opt = min(remaining, max(tau / 5, min(5 * tau, tau_opt)))
            m_opt = _ceil_clamped(j + _log(omega / gamma) / _log(kest), math.floor(3 / 4 * m), math.ceil(4 / 3 * m))
            m_opt = max(mmin, min(mmax, m_opt))
            if j == mmax:
                if omega > delta:
                    m_new = j
                    tau_new = tau * (gamma_mmax / omega) ** (1 / order)
                    tau_new = min(tau_end - tau_now, max(tau / 5, tau_new))
                else:
                    tau_new = tau_opt
                    m_new = m
            else:
                m_new = m_opt
                tau_new = same_tau
        if omega <= delta:
            reject += ireject
            step += 1
            blown = 0
            next_t = tau_now + tau
            for k in range(l, num_steps):
                if abs(tau_out[k]) < abs(next_t):
                    blown += 1
            if blown != 0:
                w[l + blown] = w[l]
                for k in range(blown):
                    F2 = _expm(sgn * (tau_out[l + k] - tau_now) * H[:j, :j])
                    w[l + k] = torch.as_tensor(beta * F2[:j, 0], dtype=dtype, device=dev) @ Vd[:j, :n]
                l += blown
            w[l] = torch.as_tensor(beta * F[:j, 0], dtype=dtype, device=dev) @ Vd[:j, :n]
            tau_now += tau
            j = 0
            ireject = 0
            conv += err
        else:
            ireject += 1
            H[0, j] = 0.0
        oldtau, tau = tau, tau_new
        oldm, m = m, int(m_new)
    if task1:
        for k in range(num_steps):
            w[k] /= tau_out[k]
    return w, (step, reject, krystep, exps, conv, m)
