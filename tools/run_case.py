#!/usr/bin/env python3
"""End-to-end run of a DCMIP case on one GPU from nothing but its parameters (development tool / demo; OUTSIDE the coverage
contract of SURVEY.md section 8 - a driver is out of scope, this is the smallest one that lets the kernels be watched on a real case):
geometry3d -> metric (+ mountain, sponge) -> initial.py state -> SSP-RK3 (pipelined stages) + exponential filter +
NaN flag, with conservation diagnostics.  Mirrors what `Simulation` does for config/dcmip31.ini / dcmip21_rk3.ini
minus configuration parsing and output.

    python tools/run_case.py --case 31 --n 8 --H 60 --V 8 --dt 0.05 --steps 20
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Epi, Ros2, StepLoop, Tvdrk3  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
from wxfactory_amd.synthetic import dfr_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", type=int, default=31, choices=(21, 22, 31))
ap.add_argument("--n", type=int, default=5)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--V", type=int, default=6)
ap.add_argument("--dt", type=float, default=0.25)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--report", type=int, default=10)
ap.add_argument("--integrator", default="tvdrk3", choices=("tvdrk3", "ros2", "epi2"))
ap.add_argument("--tol", type=float, default=1e-7)
ap.add_argument("--filter", type=float, default=None, help="exponential filter strength (default: 1e-3 for cases 21/22, off for 31)")
a = ap.parse_args()

dev = torch.device("cuda", 0)
ztop = 10000.0 if a.case == 31 else 30000.0
topo = topography_for_case(a.case, planet_for_case(a.case)[0])
t0 = time.perf_counter()
plans, sg, q0 = {}, [], []
for p in range(6):
    tile = CubedSphere3DTile(a.n, a.H, a.V, p, ztop, a.case, topo=topo)
    m = metric3d_torch(tile, dev)
    plans[p] = Euler3DPlan(a.n, a.H, a.V, a.case, p, dfr_ops(a.n), m)
    sg.append(m["sqrtG"])
    q0.append(torch.from_numpy(initial_state(tile)).to(dev))
Q = torch.stack(q0)
SG = torch.stack(sg)
print(f"setup {time.perf_counter()-t0:.1f} s: case {a.case}, n={a.n}, {a.H}x{a.H}x{a.V} elements/panel, {Q.numel()/1e6:.1f} M DOF")
w1 = np.polynomial.legendre.leggauss(a.n)[1]
w3 = torch.from_numpy(np.einsum("k,j,i->kji", w1, w1, w1).reshape(-1)).to(dev)
strength = a.filter if a.filter is not None else (1e-3 if a.case != 31 else 0.0)
filt = ExpFilter3D(make_filter(strength, 4, 0.5, np.polynomial.legendre.leggauss(a.n)[0]), sg) if strength > 0 else None
rhs = RhsEuler3D(plans)
stepper = {"tvdrk3": lambda: Tvdrk3(rhs), "ros2": lambda: Ros2(rhs, tol=a.tol, verbose=int(os.environ.get("WX_VERBOSE", "0"))), "epi2": lambda: Epi(2, rhs, tol=a.tol)}[a.integrator]()
loop = StepLoop(stepper, filt, NanFlag(dev), check_every=a.report)


def diag(X):
    mass = float((SG * X[:, 0] * w3).sum())
    return mass, float((X[:, 3] / X[:, 0]).abs().max()), float((X[:, 4] / X[:, 0]).min()), float((X[:, 4] / X[:, 0]).max())


m0, *_ = diag(Q)
torch.cuda.synchronize()
t0, last = time.perf_counter(), 0
for s in range(1, a.steps + 1):
    Q = loop.step(Q, a.dt)
    if s % a.report == 0 or s == a.steps:
        torch.cuda.synchronize()
        m, wmax, tmin, tmax = diag(Q)
        el, n_int = time.perf_counter() - t0, s - last  # this reporting interval only (the first holds the warm-up)
        print(f"step {s:5d}  t={s*a.dt:8.2f} s  mass drift {abs(m-m0)/abs(m0):.2e}  max|w| {wmax:.3e}  theta [{tmin:.3f}, {tmax:.3f}]"
              f"  {el/n_int*1e3:7.2f} ms/step  {Q.numel()*n_int/el/1e9:.2f} G DOF-steps/s", flush=True)
        info = getattr(stepper, "solver_info", None)
        if info:
            print("           last solve:", {k: v for k, v in info.items() if k in ("iterations", "rel_residual", "flag", "time")}
                  if isinstance(info, dict) else info, flush=True)
        t0, last = time.perf_counter(), s
