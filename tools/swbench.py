#!/usr/bin/env python3
"""Shallow-water S7 micro-benchmark (development tool): whole-sphere R(Q), 6 panels on one GPU,
n=8, 60x60 elements per panel; eager launches vs one captured HIP graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch  # noqa: E402
from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan  # noqa: E402

dev = torch.device("cuda", 0)
n, H = 8, 60
ops = synthetic.dfr_ops(n)
plans, qs = {}, []
for p in range(6):
    plans[p] = SwPlan(n, H, p, ops, metric2d_torch(CubedSphereTile2D(n, H, p, phi0=0.7853981633974483), dev))
    qs.append(synthetic.sw_state(n, H, p, dev))
Q = torch.stack(qs)
rhs = RhsShallowWater(plans)
for _ in range(5):
    R = rhs(Q)
torch.cuda.synchronize()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    R = rhs(Q)
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / reps
dof = 3 * 6 * H * H * n * n
print(f"S7 eager : {te*1e6:8.1f} us per whole-sphere RHS  -> {dof/te/1e9:7.2f} G DOF-updates/s  (chk {float(R.abs().max()):.6e})")

# graph capture of the 12 launches
g = torch.cuda.CUDAGraph()
Rg = torch.empty_like(Q)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        Rg.copy_(rhs(Q))
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        Rg.copy_(rhs(Q))
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / reps
print(f"S7 graph : {tg*1e6:8.1f} us per whole-sphere RHS  -> {dof/tg/1e9:7.2f} G DOF-updates/s  (match {bool(torch.equal(R, Rg))})")
print(f"algorithmic bytes 156 B/point: {156.0*6*H*H*n*n/tg/1e9:.1f} GB/s")
