#!/usr/bin/env python3
"""E7 time-stepping micro-benchmark (development tool): SSP-RK3 steps of the whole sphere on one GPU,
fused stage updates vs the literal sequence of torch axpys."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.integrators import Tvdrk3  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
n, H, V = 8, 60, int(os.environ.get("V", "8"))
ops = synthetic.dfr_ops(n)
if os.environ.get("TRUE_METRIC") == "1":  # the benchmark's metric (non-rotating planet: 312 B/point kernels)
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch

    plans = {p: Euler3DPlan(n, H, V, 31, p, ops, metric3d_torch(CubedSphere3DTile(n, H, V, p, 10000.0, 31), dev)) for p in range(6)}
else:
    plans = {p: Euler3DPlan(n, H, V, 31, p, ops, synthetic.euler3d_metric(n, H, V, p, dev)) for p in range(6)}
Q = torch.stack([synthetic.euler3d_state(n, H, V, p, dev) for p in range(6)])
rhs = RhsEuler3D(plans)
dof = Q.numel()
for fused, pipeline in ((False, False), (True, False), (True, True)):
    st = Tvdrk3(rhs, fused=fused, pipeline=pipeline)
    q = Q
    for _ in range(2):
        q = st.step(q, 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        q = st.step(q, 1e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"Tvdrk3 fused={fused!s:5} pipeline={pipeline!s:5}: {dt*1e3:7.2f} ms/step = {dt*1e3/3:6.2f} ms per stage; {dof/dt/1e9:6.2f} G DOF-steps/s; "
          f"finite={bool(torch.isfinite(q).all())}")
