#!/usr/bin/env python3
"""A/B (development tool): the six extrapolation launches of a whole-sphere E7 evaluation as ONE batched launch
(wx_euler3d_batch_extrap_pack) in front of the six per-panel fused launches, against the product's twelve launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.exchange import PanelExchange  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DBatch, Euler3DPlan, RhsEuler3D  # noqa: E402

dev = torch.device("cuda", 0)
n, H, V = 8, 60, 8
ops = synthetic.dfr_ops(n)
plans, qs = {}, []
for p in range(6):
    plans[p] = Euler3DPlan(n, H, V, 31, p, ops, metric3d_torch(CubedSphere3DTile(n, H, V, p, 10000.0, 31), dev))
    qs.append(synthetic.euler3d_state(n, H, V, p, dev, 20250824))
Q = torch.stack(qs)
ex = PanelExchange(5 * V * H * n * n, dev, rank=0, world_size=1)
rhs = RhsEuler3D(plans, ex)
batch = Euler3DBatch(plans, ex)
out = torch.empty_like(Q)
halos = {p: ex.halo_views(p) for p in range(6)}


def product():
    return rhs(Q)


def k1_batched():
    batch.extrap_pack(Q)
    for p in range(6):
        plans[p].rhs(Q[p], halos[p], out[p], _lib.WX_REGION_ALL)
    return out


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ref = product().clone()
got = k1_batched().clone()
print("identical:", bool(torch.equal(ref, got)))
for rnd in range(4):
    print(f"round {rnd}: product (12 launches) {timeit(product):.4f} ms; K1 as one batched launch + 6 fused launches {timeit(k1_batched):.4f} ms", flush=True)
