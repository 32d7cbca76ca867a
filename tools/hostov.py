import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from wxfactory_amd import synthetic
from wxfactory_amd.geometry import CubedSphereTile2D, metric2d_torch
from wxfactory_amd.rhs_sw import RhsShallowWater, SwPlan
dev = torch.device("cuda", 0)
n, H = 8, 4
ops = synthetic.dfr_ops(n)
plans, qs = {}, []
for p in range(6):
    plans[p] = SwPlan(n, H, p, ops, metric2d_torch(CubedSphereTile2D(n, H, p, phi0=0.78), dev))
    qs.append(synthetic.sw_state(n, H, p, dev))
Q = torch.stack(qs)
for direct in (True, False):
    rhs = RhsShallowWater(plans); rhs.direct = direct
    for _ in range(20): rhs(Q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): rhs(Q)
    torch.cuda.synchronize()
    print(f"direct={direct}: {(time.perf_counter()-t0)/2000*1e6:.1f} us per call (tiny problem: host-bound)")
rhs = RhsShallowWater(plans); rhs.direct = True
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): rhs(Q)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
