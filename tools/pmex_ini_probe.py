#!/usr/bin/env python3
"""EPI2 + PMEX (the schema's default exponential solver) at the sizes of the shipped 3-D Euler .ini files (development tool):
step time and Krylov vectors; under rocprofv3 --kernel-trace --stats the per-kernel durations."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Epi  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
for label, case, n, H, V, ztop, dt in (("dcmip31.ini", 31, 2, 12, 3, 10000.0, 30.0), ("dcmip21.ini", 21, 3, 3, 4, 30000.0, 25.0)):
    topo = topography_for_case(case, planet_for_case(case)[0])
    plans, q = {}, []
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, ztop, case, topo=topo)
        plans[p] = Euler3DPlan(n, H, V, case, p, synthetic.dfr_ops(n), metric3d_torch(t, dev))
        q.append(torch.from_numpy(initial_state(t)).to(dev))
    Q = torch.stack(q)
    rhs = RhsEuler3D(plans)
    for solver in ("pmex", "kiops"):
        epi, Qs, ts = Epi(2, rhs, tol=1e-7, exponential_solver=solver), Q, []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Qs = epi.step(Qs, dt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        it = int(epi.solver_info["iterations"])
        med = sorted(ts[2:])[2]
        print(f"{label} epi2 + {solver}: step {med*1e3:.2f} ms, {it} Krylov vectors, {med/it*1e6:.1f} us per vector", flush=True)
