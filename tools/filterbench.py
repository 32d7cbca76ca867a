#!/usr/bin/env python3
"""E7 explicit step with the per-step exponential filter (development tool): filter fused into the last
SSP-RK3 stage's kernel vs a separate filter pass after the step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter  # noqa: E402
from wxfactory_amd.integrators import StepLoop, Tvdrk3  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
n, H, V = 8, 60, 8
ops = synthetic.dfr_ops(n)
metrics = {p: synthetic.euler3d_metric(n, H, V, p, dev) for p in range(6)}
plans = {p: Euler3DPlan(n, H, V, 31, p, ops, metrics[p]) for p in range(6)}
Q0 = torch.stack([synthetic.euler3d_state(n, H, V, p, dev) for p in range(6)])
F = make_filter(1e-3, 4, 0.5, np.polynomial.legendre.leggauss(n)[0])
sg = [metrics[p]["sqrtG"] for p in range(6)]
for fused in (False, True):
    rhs = RhsEuler3D(plans)
    stepper = Tvdrk3(rhs)
    filt = ExpFilter3D(F, sg)
    loop = StepLoop(stepper, filt, NanFlag(dev), check_every=1000)
    if not fused:  # undo the automatic fusion: separate filter pass
        stepper.fused_filter, loop.fused = False, False
    q = Q0
    for _ in range(2):
        q = loop.step(q, 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        q = loop.step(q, 1e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"filter fused into the last stage = {fused!s:5}: {dt*1e3:7.2f} ms/step; finite={bool(torch.isfinite(q).all())}", flush=True)
