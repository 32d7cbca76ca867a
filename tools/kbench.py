#!/usr/bin/env python3
"""Kernel A/B micro-benchmark (development tool): one E7 panel, K1 + K2 timed with HIP events.

    python tools/kbench.py [--V 8] [--H 60] [--n 8] [--reps 20] [--cplx] lib1.so lib2.so ...

Each library variant (built by `python -m wxfactory_amd.build -DKNOB=.. --out=name.so`) runs in
its own subprocess (WXHIP_LIB selects it)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch

    from wxfactory_amd import _lib, synthetic
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    dev = torch.device("cuda", 0)
    n, H, V = args.n, args.H, args.V
    dtype = torch.complex128 if args.cplx else torch.float64
    m = synthetic.euler3d_metric(n, H, V, 0, dev)
    if args.rot_zero:  # a non-rotating planet: the plan finds the nine rotation symbols zero and skips them
        m["christoffel"].view(3, 9, -1)[:, :3] = 0.0
    if args.column:   # make the synthetic metric the same on all levels and take the column form of the plan
        synthetic.make_level_invariant(m, n, H, V)
    plan = Euler3DPlan(n, H, V, 31, 0, synthetic.dfr_ops(n), m, dtype=dtype, dual=args.dual, column_metric=bool(args.column))
    q = synthetic.euler3d_state(n, H, V, 0, dev)
    if args.cplx:
        q = q + 1e-8j * q
    send = torch.zeros((4, plan.edge_count), dtype=dtype, device=dev)
    sp = [send[e].data_ptr() for e in range(4)]
    out = torch.empty_like(q)
    t1, t2 = [], []
    for it in range(args.reps + 3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        plan.extrap_pack(q, sp)
        e1.record()
        plan.rhs(q, sp, out, _lib.WX_REGION_ALL)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            t1.append(e0.elapsed_time(e1))
            t2.append(e1.elapsed_time(e2))
    pts = V * H * H * n**3
    k1, k2 = sum(t1) / len(t1), sum(t2) / len(t2)
    print(f"{os.path.basename(_lib.LIB_PATH):28s} K1 {k1:7.4f} ms  K2 {k2:7.4f} ms (min {min(t2):.4f})  "
          f"K2 algorithmic ({plan.bytes_per_point:.0f} B/pt) {plan.bytes_per_point*pts/k2/1e6:7.1f} GB/s = "
          f"{plan.bytes_per_point*pts/k2/1e6/80:.1f}% of 8 TB/s; "
          f"chk {float(out.abs().max()):.6e}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--H", type=int, default=60)
    ap.add_argument("--V", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cplx", action="store_true")
    ap.add_argument("--dual", action="store_true", help="with --cplx: dual-number arithmetic (WX_DUAL128)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--column", action="store_true", help="column-invariant synthetic metric + the column form of the plan")
    ap.add_argument("--rot-zero", action="store_true", help="zero the rotation Christoffel symbols (non-rotating planet)")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a)
    else:
        libdir = os.path.join(ROOT, "wxfactory_amd", "lib")
        for lib in a.libs or ["libwxhip.so"]:
            env = dict(os.environ, WXHIP_LIB=lib if os.path.isabs(lib) else os.path.join(libdir, lib))
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--n", str(a.n), "--H", str(a.H), "--V", str(a.V),
                   "--reps", str(a.reps)] + (["--cplx"] if a.cplx else []) + (["--dual"] if a.dual else []) + (
                       ["--rot-zero"] if a.rot_zero else []) + (["--column"] if a.column else [])
            r = subprocess.run(cmd, env=env)
            if r.returncode != 0:
                print(f"{lib}: FAILED rc={r.returncode}", flush=True)
