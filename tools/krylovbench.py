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
