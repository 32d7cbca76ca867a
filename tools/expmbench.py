#!/usr/bin/env python3
"""Cost of scipy.linalg.expm on a KIOPS-sized Hessenberg matrix on this host (development tool)."""
import time

import numpy as np
from scipy.linalg import expm

rng = np.random.default_rng(0)
for m in (17, 48, 65):
    H = np.triu(rng.standard_normal((m, m)), -1) * 3.0
    expm(H)
    t0 = time.perf_counter()
    for _ in range(20):
        expm(H)
    t1 = (time.perf_counter() - t0) / 20
    try:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1):
            expm(H)
            t0 = time.perf_counter()
            for _ in range(20):
                expm(H)
            t2 = (time.perf_counter() - t0) / 20
    except Exception as e:  # pragma: no cover
        t2 = float("nan")
        print(e)
    print(f"m={m}: expm {t1*1e3:.2f} ms, single-threaded BLAS {t2*1e3:.2f} ms")
