#!/usr/bin/env python3
"""JVP kernel micro-benchmark (development tool): ONE E7 panel, the kernels of a complex-step Jacobian-vector product
timed with HIP events - the prepared form a Krylov solve runs (tangent extrapolation + euler_jvp_kernel reading cached
face values) and the unprepared one.  Run directly, or under rocprofv3 (program right after "--"):

    python tools/jvpkbench.py [--V 8] [--H 60] [--n 8] [--reps 20] [--full-metric] [--unprepared]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--H", type=int, default=60)
    ap.add_argument("--V", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--full-metric", action="store_true", help="keep the nine rotation Christoffel symbols (384 B/pt RHS)")
    ap.add_argument("--unprepared", action="store_true")
    a = ap.parse_args()
    import torch

    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    dev = torch.device("cuda", 0)
    n, H, V = a.n, a.H, a.V
    m = synthetic.euler3d_metric(n, H, V, 0, dev)
    if not a.full_metric:
        m["christoffel"].view(3, 9, -1)[:, :3] = 0.0
    plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m, dtype=torch.complex128, dual=True)
    q = synthetic.euler3d_state(n, H, V, 0, dev)
    v = (torch.rand(q.shape, device=dev, dtype=torch.float64) - 0.5) * q.abs().amax(dim=(1, 2, 3, 4), keepdim=True) * 1e-3
    eps = 1.4901161193847656e-08
    sv = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    stn = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=dev)
    sc = torch.zeros((4, plan.edge_count), dtype=torch.complex128, device=dev)
    out = torch.empty_like(q)
    if not a.unprepared:
        plan.jvp_prepare(q, list(sv))
    t1, t2 = [], []
    for it in range(a.reps + 3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        if a.unprepared:
            plan.jvp_extrap_pack(q, v, eps, list(sc))
            e1.record()
            plan.jvp(q, v, eps, list(sc), out, 1.0 / eps, _lib.WX_REGION_ALL)
        else:
            plan.jvp_tangent_pack(q, v, eps, list(stn))
            e1.record()
            plan.jvp_prepared(q, v, eps, list(sv), list(stn), out, 1.0 / eps, _lib.WX_REGION_ALL)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            t1.append(e0.elapsed_time(e1))
            t2.append(e1.elapsed_time(e2))
    pts = V * H * H * n**3
    k1, k2 = sum(t1) / len(t1), sum(t2) / len(t2)
    # compulsory bytes per point of the JVP kernel: Q, v, the RHS kernel's static fields, the real tangent out
    static = plan.bytes_per_point - 80.0
    bpp = 40.0 + 40.0 + static + 40.0
    bpp1 = (16.0 + 40.0) + 6.0 * 5 * 8 / n   # tangent extrapolation: two log rows of Q + v in, face tangents out
    print(f"jvp {'unprepared' if a.unprepared else 'prepared'} n={n} H={H} V={V}: extrap {k1:7.4f} ms "
          f"({bpp1 * pts / k1 / 1e6:7.1f} GB/s on {bpp1:.1f} B/pt)  jvp kernel {k2:7.4f} ms (min {min(t2):.4f}) "
          f"= {bpp * pts / k2 / 1e6:7.1f} GB/s on {bpp:.0f} B/pt = {bpp * pts / k2 / 1e6 / 80:.1f}% of 8 TB/s; "
          f"chk {float(out.abs().max()):.6e}", flush=True)


if __name__ == "__main__":
    main()
