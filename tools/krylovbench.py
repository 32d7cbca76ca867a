#!/usr/bin/env python3
"""Krylov vector kernels at the E7 vector length (development tool): wx_multi_dot / wx_multi_axpy against the
torch expressions they replace in fgmres (442 M doubles per vector, 3.5 GB)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd.solvers import _Basis  # noqa: E402

dev = "cuda:0"
n, m = 442_368_000, 12
V = torch.randn((m + 1, n), device=dev, dtype=torch.float64)
w = torch.randn(n, device=dev, dtype=torch.float64)
basis = _Basis(V)


def clock(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for k in (1, 4, 8, 12):
    h = torch.full((k,), 1e-9, device=dev, dtype=torch.float64)
    td, ta = clock(lambda: basis.dots(0, k, w)), clock(lambda: basis.subtract(w, 0, k, h))
    tt, tu = clock(lambda: V[:k] @ w), clock(lambda: w - h @ V[:k])
    read = (k + 1) * n * 8 / 1e9
    print(f"{k:2d} rows: multi_dot {td:6.2f} ms ({read / td:5.2f} TB/s; torch mv {tt:6.2f} ms)   "
          f"multi_axpy {ta:6.2f} ms ({(read + n * 8 / 1e9) / ta:5.2f} TB/s; torch w - h @ V {tu:6.2f} ms)", flush=True)


if "fgmres" in sys.argv[1:]:
    # one restart cycle of FGMRES on the Rosenbrock operator of the E7 sphere (matvec_rat; synthetic metric)
    del V, w, basis
    torch.cuda.empty_cache()
    from wxfactory_amd import synthetic  # noqa: E402
    from wxfactory_amd.matvec import MatvecOpRat  # noqa: E402
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
    from wxfactory_amd.solvers import fgmres  # noqa: E402

    nn, H, Vv = 8, 60, 8
    ops = synthetic.dfr_ops(nn)
    plans = {p: Euler3DPlan(nn, H, Vv, 31, p, ops, synthetic.euler3d_metric(nn, H, Vv, p, dev)) for p in range(6)}
    Q = torch.stack([synthetic.euler3d_state(nn, H, Vv, p, dev) for p in range(6)])
    rhs = RhsEuler3D(plans)
    R = rhs(Q)
    dt = 0.05
    A = MatvecOpRat(dt, Q, R, rhs)
    b = A(Q.flatten()) + R.flatten() * dt
    from wxfactory_amd import solvers  # noqa: E402

    calls = [0]
    orig = solvers._Basis.dots

    def counting(self, lo, hi, w, out=None):
        calls[0] += 1
        return orig(self, lo, hi, w, out=out)

    solvers._Basis.dots = counting
    for label, eta, ortho in (("warm-up", solvers._REORTH, "cgs"), ("cgs, second pass on cancellation", solvers._REORTH, "cgs"),
                              ("cgs, second pass always", 1e30, "cgs"), ("igs warm-up", 0.1, "igs"),
                              ("igs: the reference's one-synchronisation Gram-Schmidt (default)", 0.1, "igs")):
        solvers._REORTH = eta
        calls[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, nr, nb, it, flag, res = fgmres(A, b, x0=Q.flatten(), tol=1e-30, restart=20, maxiter=1, ortho=ortho)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"fgmres restart 20, {label}: {it} iterations, {calls[0]} dot sweeps, {t*1e3:.0f} ms = {t/it*1e3:.1f} ms "
              f"per iteration (incl. 2 residual evaluations), residual {nr/nb:.3e}", flush=True)
