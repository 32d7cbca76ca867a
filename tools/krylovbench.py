import sys, time
sys.path.insert(0, '/root/repo')
import torch
from wxfactory_amd.solvers import _Basis, fgmres
dev = 'cuda:0'
n = 442_368_000
m = 12
V = torch.randn((m + 1, n), device=dev, dtype=torch.float64)
w = torch.randn(n, device=dev, dtype=torch.float64)
b = _Basis(V)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for k in (1, 4, 8, 12):
    td = t(lambda: b.dots(0, k, w)); ta = t(lambda: b.subtract(w, 0, k, torch.full((k,), 1e-9, device=dev, dtype=torch.float64)))
    tt = t(lambda: V[:k] @ w)
    gb = (k + 1) * n * 8 / 1e9
    print(f"rows {k:2d}: multi_dot {td:7.2f} ms ({gb/td:6.1f} GB/ms... {gb/td*1e3/1e3:5.2f} TB/s)  multi_axpy {ta:7.2f} ms ({(gb + n*8/1e9)/ta:5.2f} TB/s)  torch mv {tt:7.2f} ms", flush=True)
