#!/usr/bin/env python3
"""EPI2 + KIOPS (or, SOLVER=pmex, PMEX) steps at the benchmark's resolution (development tool): n = 8, 60 x 60 elements per panel, V vertical
elements (default 2; V = 8, the whole benchmark sphere, runs with the Krylov basis sized to the free memory), whole sphere on
one GPU.  Time per step, per Krylov vector, and what a bare prepared matvec costs - the overhead of everything around it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Epi  # noqa: E402
from wxfactory_amd.matvec import ComplexStepOperator  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
from wxfactory_amd.synthetic import dfr_ops  # noqa: E402

n, H, V = 8, 60, int(os.environ.get("V", "2"))
dt = float(os.environ.get("DT", "2.0"))
dev = torch.device("cuda", 0)
plans, Q = {}, []
gen = torch.Generator(device=dev).manual_seed(3)
for p in range(6):
    tile = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
    plans[p] = Euler3DPlan(n, H, V, 31, p, dfr_ops(n), metric3d_torch(tile, dev))
    q = torch.from_numpy(initial_state(tile)).to(dev)
    Q.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=dev, dtype=q.dtype) - 0.5)))
Q = torch.stack(Q)
rhs = RhsEuler3D(plans)
R = rhs(Q)
op = ComplexStepOperator(dt, Q, R, rhs)
v = torch.randn(Q.numel(), device=dev, dtype=torch.float64)
for _ in range(3):
    op(v)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    op(v)
torch.cuda.synchronize()
mv = (time.perf_counter() - t0) / 10
rhs.jvp_release()
del op, v
solver = os.environ.get("SOLVER", "kiops")   # or pmex
epi = Epi(2, rhs, tol=1e-7, exponential_solver=solver)
print("exponential solver", solver, flush=True)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q = epi.step(Q, dt)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    info = epi.solver_info
    print(f"step {i}: {t*1e3:8.1f} ms, {info['iterations']} Krylov vectors ({info['substeps']} substeps, {info['rejected']} rejected) "
          f"= {t/info['iterations']*1e3:6.2f} ms per vector; bare prepared matvec {mv*1e3:6.2f} ms -> overhead "
          f"{(t/info['iterations']/mv - 1)*100:5.1f} %", flush=True)
