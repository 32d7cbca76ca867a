#!/usr/bin/env python3
"""Where an EPI2 + KIOPS step at the size of config/dcmip31.ini spends its time (development probe): wall time of
the Krylov passes (synchronised), the host-side matrix exponentials, everything else."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import solvers  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Epi  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402
from wxfactory_amd.synthetic import dfr_ops  # noqa: E402

n, H, V, case = 2, 12, 3, 31
dev = torch.device("cuda", 0)
topo = topography_for_case(case, planet_for_case(case)[0])
plans, Q = {}, []
for p in range(6):
    tile = CubedSphere3DTile(n, H, V, p, 10000.0, case, topo=topo)
    plans[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metric3d_torch(tile, dev))
    Q.append(torch.from_numpy(initial_state(tile)).to(dev))
Q = torch.stack(Q)
rhs = RhsEuler3D(plans)
epi = Epi(2, rhs, tol=1e-7)
for _ in range(4):
    Q = epi.step(Q, 30.0)
torch.cuda.synchronize()
acc = {"pass": 0.0, "expm": 0.0, "vectors": 0}
orig_pass, orig_expm = solvers.KiopsWorkspace.run_pass, solvers._expm


def timed_pass(self, j0, m, build, use_graphs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    orig_pass(self, j0, m, build, use_graphs)
    torch.cuda.synchronize()
    acc["pass"] += time.perf_counter() - t
    acc["vectors"] += m - j0


def timed_expm(M):
    t = time.perf_counter()
    F = orig_expm(M)
    acc["expm"] += time.perf_counter() - t
    return F


solvers.KiopsWorkspace.run_pass, solvers._expm = timed_pass, timed_expm
steps = 6
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    Q = epi.step(Q, 30.0)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"per step: total {tot/steps*1e3:.2f} ms; Krylov passes {acc['pass']/steps*1e3:.2f} ms for {acc['vectors']/steps:.0f} vectors "
      f"({acc['pass']/acc['vectors']*1e6:.1f} us/vector); expm (host) {acc['expm']/steps*1e3:.2f} ms; "
      f"rest {(tot-acc['pass']-acc['expm'])/steps*1e3:.2f} ms; {epi.solver_info}")
