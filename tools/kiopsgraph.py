#!/usr/bin/env python3
"""EPI2 + KIOPS at the size of config/dcmip31.ini: HIP-graph replay of the Krylov passes against eager launches
(development tool): same states bit for bit, ms per step, us per Krylov vector."""
import os
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
plans, Q0 = {}, []
for p in range(6):
    tile = CubedSphere3DTile(n, H, V, p, 10000.0, case, topo=topo)
    plans[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metric3d_torch(tile, dev))
    Q0.append(torch.from_numpy(initial_state(tile)).to(dev))
Q0 = torch.stack(Q0)
rhs = RhsEuler3D(plans)
res = {}
for graphs in (False, True):
    epi = Epi(2, rhs, tol=1e-7)
    epi.graph_passes = graphs
    Q, ts, its = Q0, [], []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Q = epi.step(Q, 30.0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        its.append(epi.solver_info["iterations"])
    ws = epi._ws if graphs else None
    med = sorted(ts[4:])[len(ts[4:]) // 2]
    print(f"graphs={graphs}: ms/step {['%.2f' % t for t in ts]} median(5..) {med:.2f} ms, vectors {its}, "
          f"us/vector {med * 1e3 / its[-1]:.1f}" + (f", captures {ws.captures}, replays {ws.replays}, keys {sorted(ws.graphs)}" if ws else ""))
    res[graphs] = Q
print("states equal bit for bit:", bool(torch.equal(res[False], res[True])))
