#!/usr/bin/env python3
"""The reference's own RHS benchmark matrix (tests/rhs_benchmark/run.sh:67-71: 60^3 points per panel at n = 2..6) and E7
at V = 1, whole sphere on one GPU (development tool).  Wall time per evaluation and - with HIP events around each
phase - the two launches separately; under `rocprofv3 --kernel-trace --stats` the kernels separate by template name.

    python tools/matrixbench.py [--orders 2,3,4,5,6,8] [--reps 50]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib, synthetic  # noqa: E402
from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch  # noqa: E402
from wxfactory_amd.initial import initial_state  # noqa: E402
from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D  # noqa: E402

SIZES = {2: (30, 30), 3: (20, 20), 4: (15, 15), 5: (12, 12), 6: (10, 10), 7: (9, 9), 8: (60, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", default="2,3,4,5,6,8")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for n in (int(x) for x in a.orders.split(",")):
        H, V = SIZES[n]
        plans, q = {}, []
        for p in range(6):
            t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
            plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, dev))
            q.append(torch.from_numpy(initial_state(t)).to(dev))
        Q = torch.stack(q)
        rhs = RhsEuler3D(plans)
        assert rhs._small_tiles()
        for _ in range(5):
            rhs(Q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            rhs(Q)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.reps
        # the two batched launches separately
        ex = rhs.exchange_for(torch.float64)
        bt = rhs._batch_for(torch.float64, plans, ex)
        out = torch.empty_like(Q)
        t1, t2 = [], []
        for it in range(a.reps):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            bt.extrap_pack(Q)
            e1.record()
            bt.rhs(Q, out, _lib.WX_REGION_ALL)
            e2.record()
            torch.cuda.synchronize()
            t1.append(e0.elapsed_time(e1))
            t2.append(e1.elapsed_time(e2))
        # the complex-step matvec the benchmark's epi2 + KIOPS integrator calls once per Krylov vector
        from wxfactory_amd.matvec import matvec_fun

        R = rhs(Q)
        v = (torch.rand(Q.shape, device=dev, dtype=Q.dtype) - 0.5).flatten()
        for _ in range(5):
            matvec_fun(v, 1.0, Q, R, rhs, "complex")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            matvec_fun(v, 1.0, Q, R, rhs, "complex")
        torch.cuda.synchronize()
        mv = (time.perf_counter() - t0) / a.reps
        pts = 6 * V * H * H * n**3
        bpp = plans[0].bytes_per_point
        k1, k2 = sorted(t1)[len(t1) // 2], sorted(t2)[len(t2) // 2]
        print(f"n={n} {H}x{H}x{V}: wall {wall*1e3:7.4f} ms/eval = {bpp*pts/wall/1e9:7.1f} GB/s ({bpp:.0f} B/pt); "
              f"extrap {k1*1e3:6.1f} us, rhs kernel {k2*1e3:6.1f} us = {bpp*pts/(k2*1e-3)/1e9:7.1f} GB/s "
              f"({bpp*pts/(k2*1e-3)/1e9/80:.1f}% of 8 TB/s); complex-step matvec {mv*1e3:7.4f} ms = {mv/wall:.2f} x R(Q)", flush=True)
        del rhs, plans, Q, q, out, bt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
