#!/usr/bin/env python3
"""A Rosenbrock-2 + FGMRES step at the sizes the shipped 3-D Euler .ini files configure (development tool): per-iteration cost
against the operator's, i.e. what the host round trip of the one-synchronisation Gram-Schmidt costs at launch-bound sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.integrators import Ros2  # noqa: E402
from wxfactory_amd.matvec import matvec_rat  # noqa: E402
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
    R = rhs(Q)
    v = torch.randn(Q.numel(), device=dev, dtype=torch.float64)
    for _ in range(5):
        matvec_rat(v, dt, Q, R, rhs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        matvec_rat(v, dt, Q, R, rhs)
    torch.cuda.synchronize()
    mv = (time.perf_counter() - t0) / 200
    for ortho, vec, one in (("igs", "1", "1"), ("igs", "1", "0"), ("igs", "0", "1"), ("cgs", "0", "1")):
        os.environ["WXHIP_FGMRES_VECTOR"] = vec   # 1: device passes (one read-back per pass of several vectors), 0: one per vector
        os.environ["WXHIP_FGMRES_ONE_LAUNCH"] = one   # 1: the step's products, algebra and update from one launch, 0: three
        ros, Qs, ts = Ros2(rhs, tol=1e-7, gmres_restart=30, ortho=ortho), Q, []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Qs = ros.step(Qs, dt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        it = ros.solver_info["iterations"]
        med = sorted(ts[1:])[1]
        si = ros.solver_info
        print(f"{label} ros2 + fgmres({ortho}, device passes {'on' if vec == '1' else 'off'}{', one-launch step' if vec == '1' and one == '1' else ''}): step {med*1e3:.2f} ms, {it} iterations, "
              f"{med/it*1e6:.1f} us per iteration; operator {mv*1e6:.1f} us; flag {si['flag']}; passes {si.get('device_passes')}, "
              f"vectors built {si.get('vectors_built')}, wasted {si.get('wasted_vectors')}, redone on the host {si.get('host_redone_steps')}; "
              f"host: enqueue {si.get('enqueue_s', 0)*1e3:.2f} ms, wait {si.get('wait_s', 0)*1e3:.2f} ms, cycle ends {si.get('cycle_end_s', 0)*1e3:.2f} ms",
              flush=True)
