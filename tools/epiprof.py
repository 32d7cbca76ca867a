#!/usr/bin/env python3
"""Host-side profile of EPI2 + KIOPS steps at the size of config/dcmip31.ini (development tool)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

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
for _ in range(3):
    Q = epi.step(Q, 30.0)
torch.cuda.synchronize()
if os.environ.get("WX_EXPM_DUMP"):
    import numpy as np
    import scipy.linalg

    _expm = scipy.linalg.expm

    def timed_expm(M):
        t = time.perf_counter()
        F = _expm(M)
        dt = time.perf_counter() - t
        print(f"expm {M.shape} {dt*1e3:.2f} ms, |M|_1 = {np.abs(M).sum(axis=0).max():.3e}, min nonzero |M| = "
              f"{np.abs(M[M != 0]).min():.3e}, subnormal results: {int(((np.abs(F) < 2.3e-308) & (F != 0)).sum())}")
        np.save(os.environ["WX_EXPM_DUMP"], M)
        return F

    scipy.linalg.expm = timed_expm
t0 = time.time()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    Q = epi.step(Q, 30.0)
torch.cuda.synchronize()
pr.disable()
print("ms/step", (time.time() - t0) / 5 * 1e3, epi.solver_info)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
