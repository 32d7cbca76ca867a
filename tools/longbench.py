#!/usr/bin/env python3
"""The two streaming kernels of KIOPS' long-vector build timed alone (development tool): wx_kiops_long_a_scaled (5 sweeps at
p = 1, iop = 2) and wx_kiops_long_b_scaled (4 sweeps) on vectors of N doubles (default: the whole E7 sphere, 442 368 000).
ROWS=k: a basis of k rows (the kernels work on the last three); PAD=0: rows of n + 1 doubles, as before the workspace
padded them (odd row stride: every other row starts 8 bytes off a 16-byte boundary).   python tools/longbench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 442_368_000
dev = torch.device("cuda", 0)
lib = _lib.load()
p, iop = 1, 2
j = int(os.environ.get("ROWS", "4")) - 1   # the basis has ROWS rows, the kernels work on the last three (does its size matter?)
ld = -(-(n + p) // 32) * 32 if os.environ.get("PAD", "1") != "0" else n + p
V = torch.empty((j + 1, ld), dtype=torch.float64, device=dev)[:, : n + p]
V[j - iop - 1:].normal_()
aw = torch.randn(n, dtype=torch.float64, device=dev)
u = torch.randn((n, p), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.wx_kiops_long_workspace()), dtype=torch.float64, device=dev)
dots = torch.empty(4, dtype=torch.float64, device=dev)
nrm2 = torch.empty(1, dtype=torch.float64, device=dev)
scales = torch.ones(j + 1, dtype=torch.float64, device=dev)
h = torch.full((iop,), 1e-3, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def a():
    _lib.check(lib.wx_kiops_long_a_scaled(V.data_ptr(), V.stride(0), j, n, p, iop, aw.data_ptr(), u.data_ptr(), dots.data_ptr(),
                                          work.data_ptr(), scales.data_ptr(), st), "a")


def b():
    _lib.check(lib.wx_kiops_long_b_scaled(V.data_ptr(), V.stride(0), j, n, p, iop, h.data_ptr(), nrm2.data_ptr(), work.data_ptr(),
                                          scales.data_ptr(), st), "b")


for name, fn, sweeps in (("long_a", a, 3 + iop), ("long_b", b, 2 + iop)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"PAD={os.environ.get('PAD', '1')} {name}: {ms:.3f} ms, {sweeps} sweeps of {n * 8 / 1e9:.2f} GB -> {sweeps * n * 8 / ms / 1e6:.0f} GB/s; "
          f"dots {dots[:iop].tolist() if name == 'long_a' else nrm2.tolist()}", flush=True)
