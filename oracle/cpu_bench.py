"""One CPU worker of bench.py's cpu_baseline: times the CPU restatement of the E7 workload
on one tile and prints one JSON line.  bench.py starts six of these at once, one cube panel each, with
OMP/BLAS threads = floor(cores / 6).

    python -m oracle.cpu_bench --flavour cpp|dense|numpy --n 8 --H 30 --V 8 --reps 3 --threads 2 --panel 0

TEST INFRASTRUCTURE (it times the checker; nothing in the product path imports it).
"""
import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flavour", choices=("cpp", "dense", "numpy"), default="cpp")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--H", type=int, default=30)
    ap.add_argument("--V", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20250824)
    a = ap.parse_args()
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = str(a.threads)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch

    torch.set_num_threads(a.threads)
    from wxfactory_amd import synthetic

    n, H, V = a.n, a.H, a.V
    m = synthetic.euler3d_metric(n, H, V, a.panel, "cpu", a.seed)
    q = synthetic.euler3d_state(n, H, V, a.panel, "cpu", a.seed).numpy()
    om = {"sqrtG_new": m["sqrtG"].numpy(), "inv_sqrtG_new": (1.0 / m["sqrtG"]).numpy(),
          "h_contra_new": m["h_contra"].numpy(), "christoffel": m["christoffel"].numpy(),
          "inv_dzdeta_new": m["inv_dzdeta"].numpy()}
    for d in "ijk":
        om[f"sqrtG_itf_{d}_new"] = m[f"sqrtG_itf_{d}"].numpy()
        om[f"h_contra_itf_{d}_new"] = m[f"h_contra_itf_{d}"].numpy()
    bsn = np.tile(m["boundary_sn"].numpy().reshape(H, 1, n), (1, n, 1))
    if a.flavour == "cpp":
        from oracle.c_port import Euler3DPortC

        o = Euler3DPortC(n, H, V, 31, synthetic.dfr_ops(n), om, bsn, bsn, panel=a.panel, threads=a.threads)
    elif a.flavour == "dense":
        from oracle.euler3d_dense import Euler3DOracleDense

        o = Euler3DOracleDense(n, H, V, 31, synthetic.dfr_ops(n), om, bsn, bsn, panel=a.panel)
    else:
        from oracle.euler3d import Euler3DOracle

        o = Euler3DOracle(n, H, V, 31, synthetic.dfr_ops(n), om, bsn, bsn, panel=a.panel)
    itf = o.extrapolate(q)
    halo = o.pack_edges(itf)   # any finite halo: timing only
    o.rhs(q, halo, itf=itf)    # warm-up
    print("READY", flush=True)
    sys.stdin.readline()       # all six workers start their timed loop together
    t0 = time.perf_counter()
    for _ in range(a.reps):
        itf = o.extrapolate(q)
        o.pack_edges(itf)
        r = o.rhs(q, halo, itf=itf)
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"flavour": a.flavour, "panel": a.panel, "s_per_eval": dt, "dof": 5 * V * H * H * n**3,
                      "threads": a.threads, "finite": bool(np.isfinite(r).all())}), flush=True)


if __name__ == "__main__":
    main()
