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

# graph capture of the whole evaluation (2 launches: all panels per phase).  Round 2's version of this benchmark captured
# `Rg.copy_(rhs(Q))` - a third kernel copying 33 MB - which is what made its replay slower than the eager call.
from wxfactory_amd.graph import GraphedFunction  # noqa: E402

gf = GraphedFunction(rhs, Q)
for _ in range(5):
    Rg = gf(Q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    Rg = gf(Q)
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / reps
print(f"S7 graph : {tg*1e6:8.1f} us per whole-sphere RHS  -> {dof/tg/1e9:7.2f} G DOF-updates/s  (match {bool(torch.equal(R, Rg))})")
print(f"algorithmic bytes 156 B/point: eager {156.0*6*H*H*n*n/te/1e9:.1f} GB/s, graph {156.0*6*H*H*n*n/tg/1e9:.1f} GB/s")
# the two launches separately (HIP events)
bt = rhs._batch if hasattr(rhs, "_batch") else None
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    rhs(Q)
ev[1].record()
torch.cuda.synchronize()
print(f"S7 GPU time per evaluation (events around {reps} back-to-back evaluations): {ev[0].elapsed_time(ev[1])/reps*1e3:.1f} us")

# SSP-RK3 step (integrators/tvdrk3.py:12-19): fused stages with a separate extrapolation launch per stage, against the stage
# pipeline (wx_sw_batch_stage: the stage's kernel extrapolates its own output; 3 launches per step instead of 6)
from wxfactory_amd.integrators import Tvdrk3  # noqa: E402


def step_us(stepper, reps=100):
    q = Q
    for _ in range(5):
        q = stepper.step(q, 1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        q = stepper.step(q, 1.0)
    b.record()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(q).all())
    return a.elapsed_time(b) / reps * 1e3


two_f, two_p = RhsShallowWater(plans), RhsShallowWater(plans)
two_f.direct = two_p.direct = False   # the two-kernel form: fused stages with an extrapolation launch each, and its stage pipeline
fused, piped = Tvdrk3(two_f, pipeline=False), Tvdrk3(two_p)
assert piped.pipeline and not fused.pipeline
auto = Tvdrk3(RhsShallowWater(plans))   # the default: the direct form at this size, one launch (+ the ring pack) per fused stage
assert not auto.pipeline and auto.fused
tf, tp, ta = step_us(fused), step_us(piped), step_us(auto)
print(f"S7 SSP-RK3 step: fused stages {tf:.1f} us, stage pipeline {tp:.1f} us ({tf / tp:.2f} x)  -> per stage {tp / 3:.1f} us = "
      f"{156.0 * 6 * H * H * n * n / (tp / 3 * 1e-6) / 1e9:.1f} GB/s on 156 B/point")
print(f"S7 SSP-RK3 step, default (direct form, fused stages): {ta:.1f} us  -> per stage {ta / 3:.1f} us = "
      f"{156.0 * 6 * H * H * n * n / (ta / 3 * 1e-6) / 1e9:.1f} GB/s on 156 B/point")

# ... and the same two steps replayed from HIP graphs (the host out of the way: what the kernels themselves take)
for name, stepper in (("fused stages", fused), ("stage pipeline", piped)):
    if name == "stage pipeline":
        stepper.rhs.reserve(stage=True)

    def one_step(q, stepper=stepper):
        if hasattr(stepper.rhs, "invalidate_faces"):
            stepper.rhs.invalidate_faces()       # a replay starts from a copied-in state: its faces are not prepared
        return stepper.step(q, 1.0)

    gs = GraphedFunction(one_step, Q)
    q = Q
    for _ in range(5):
        q = gs(q)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        q = gs(Q)
    b.record()
    torch.cuda.synchronize()
    print(f"S7 SSP-RK3 step from a HIP graph, {name}: {a.elapsed_time(b) / 100 * 1e3:.1f} us")


# the direct form (no interface buffer: ring-only pack + ONE launch) against the two-kernel form, eager, GPU time from events
direct, two = RhsShallowWater(plans), RhsShallowWater(plans)
direct.direct, two.direct = True, False   # (the default is "auto": the direct form at this size)
# (alternating rounds, median: either form alone moves by +-5 % between runs on this power-limited chip)
times = {"two kernels": [], "direct form": []}
last = {}
for rnd in range(9):
    for name, r in (("two kernels", two), ("direct form", direct)):
        for _ in range(5):
            Rd = r(Q)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            Rd = r(Q)
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) / reps * 1e-3)
        last[name] = Rd
for name, ts in times.items():
    t = sorted(ts)[len(ts) // 2]
    print(f"S7 R(Q), {name}: {t*1e6:6.1f} us (median of 9 alternating rounds, {min(ts)*1e6:.1f}-{max(ts)*1e6:.1f}) -> "
          f"{156.0*6*H*H*n*n/t/1e9:7.1f} GB/s on 156 B/point = {156.0*6*H*H*n*n/t/1e9/80:.1f} % of 8 TB/s"
          f"  (max |diff| vs two kernels {float((last[name] - R).abs().max()):.2e})")
