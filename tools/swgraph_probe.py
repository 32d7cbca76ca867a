#!/usr/bin/env python3
"""Shallow water S7: where a HIP-graph replay of R(Q) spends its time against eager launches (development tool).
eager | GraphedFunction (copies its input in) | the same graph replayed on its own static input (no copy) | ten evaluations
in one graph | a graph of an EMPTY kernel sequence's cost is read off the difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wxfactory_amd import synthetic  # noqa: E402
from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch  # noqa: E402
from wxfactory_amd.graph import GraphedFunction  # noqa: E402
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


def t(fn, reps=400, per=1):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps / per * 1e3


gf = GraphedFunction(rhs, Q)
static_in = gf.inputs[0]


def ten(q):
    r = None
    for _ in range(10):
        r = rhs(q)
    return r


g10 = GraphedFunction(ten, Q)
Q2 = Q.clone()
for rnd in range(3):
    print(f"eager R(Q)                                   {t(lambda: rhs(Q)):6.1f} us")
    print(f"graph, input copied in (33 MB)               {t(lambda: gf(Q)):6.1f} us")
    print(f"graph on its own static input (no copy)      {t(lambda: gf(static_in)):6.1f} us")
    print(f"graph.replay() alone                         {t(gf.graph.replay):6.1f} us")
    print(f"the copy alone (Q2.copy_(Q))                 {t(lambda: Q2.copy_(Q)):6.1f} us")
    print(f"ten evaluations in one graph, per evaluation {t(g10.graph.replay, reps=100, per=10):6.1f} us")
    print(f"ten eager evaluations, per evaluation        {t(lambda: ten(Q), reps=100, per=10):6.1f} us", flush=True)
