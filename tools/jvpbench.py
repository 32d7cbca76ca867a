#!/usr/bin/env python3
"""E7 Jacobian-vector-product micro-benchmark (development tool): whole sphere on one GPU,
matvec_fun (solvers/matvec.py) in its flavours."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.matvec import matvec_fun, matvec_rat  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
n, H, V = 8, 60, int(os.environ.get("V", "8"))
ops = synthetic.dfr_ops(n)
plans = {p: Euler3DPlan(n, H, V, 31, p, ops, synthetic.euler3d_metric(n, H, V, p, dev)) for p in range(6)}
Q = torch.stack([synthetic.euler3d_state(n, H, V, p, dev) for p in range(6)])
v = (torch.rand(Q.shape, device=dev, dtype=torch.float64) - 0.5) * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True) * 1e-3
dt = 30.0


def timeit(label, fn, reps=5):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {t*1e3:8.2f} ms   ({Q.numel()/t/1e9:6.2f} G DOF/s)  |out| {float(out.abs().max()):.4e}", flush=True)
    return out


rhs = RhsEuler3D(plans)
R = timeit("R(Q)  real RHS", lambda: rhs(Q))
ref = timeit("matvec_fun complex: fused dual JVP (wx_euler3d_jvp)", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "complex"))
rhs.jvp_prepare(Q)
pre = timeit("matvec_fun complex: PREPARED (face values cached, tangents only)", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "complex"))
rhs.jvp_release()
print("prepared == unprepared bit for bit:", bool(torch.equal(pre, ref)))
rhs.fused_jvp = False
rhs_d = RhsEuler3D(plans, complex_arith="dual")
rhs_d.fused_jvp = False
b = timeit("matvec_fun complex: torch.complex + dual kernels + .imag", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs_d, "complex"))
c = timeit("matvec_fun complex: torch.complex + complex128 kernels", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "complex"))
d = timeit("matvec_fun fd: shift on load + fused store", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "fd"))
e = timeit("matvec_rat: shift on load + fused store", lambda: matvec_rat(v.flatten(), dt, Q, R, rhs))
rhs.fused_shift = False
d2 = timeit("matvec_fun fd: torch add + fused store (axpy2)", lambda: matvec_fun(v.flatten(), dt, Q, R, rhs, "fd"))
timeit("matvec_rat: torch add + fused store (axpy2)", lambda: matvec_rat(v.flatten(), dt, Q, R, rhs))
print("fd shift-on-load vs materialised rel diff", float((d - d2).abs().max() / d2.abs().max()))
print("fused vs complex128 rel diff", float((ref - c).abs().max() / c.abs().max()), " dual vs complex128", float((b - c).abs().max() / c.abs().max()),
      " fd vs complex", float((d - c).abs().max() / c.abs().max()))
