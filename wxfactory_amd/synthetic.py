"""Seeded synthetic cubed-sphere states and metric fields with physical magnitudes.

The reference's geometry / initial-condition generators are setup code outside the hot path
and cannot travel to the GPU box, so timing runs use these fields (SURVEY.md section 8d gives
the recipe: magnitudes only - values do not affect speed).  Parity uses the golden fixtures.
Arrays come out in the reference's layouts (geometry/cubed_sphere_3d.py:187-205, metric3d.py:1108-1157),
interface metric single-valued per face with the outermost slots zero as cubed_sphere_3d.py:825-826.
"""
import math
from typing import Dict

import torch


def _u(shape, gen, device):
    """xi ~ U(-1, 1)"""
    return torch.rand(shape, generator=gen, device=device, dtype=torch.float64) * 2.0 - 1.0


def _pad_itf(face: torch.Tensor, axis: int, n2: int) -> torch.Tensor:
    """(…, faces=E+1 along `axis`, n2) single-valued face field -> halo-padded (…, E+2, 2*n2) layout."""
    E = face.shape[axis] - 1
    shape = list(face.shape)
    shape[axis] = E + 2
    shape[-1] = 2 * n2
    out = torch.zeros(shape, dtype=face.dtype, device=face.device)
    idx_minus = [slice(None)] * face.ndim
    idx_plus = [slice(None)] * face.ndim
    idx_minus[axis] = slice(1, None)   # minus slot of padded index f+1 = face f
    idx_minus[-1] = slice(0, n2)
    idx_plus[axis] = slice(0, E + 1)   # plus slot of padded index f = face f
    idx_plus[-1] = slice(n2, None)
    out[tuple(idx_minus)] = face
    out[tuple(idx_plus)] = face
    # outermost slots (never read) are zero like the reference's
    z0 = [slice(None)] * face.ndim
    z0[axis] = 0
    z0[-1] = slice(0, n2)
    out[tuple(z0)] = 0.0
    z1 = [slice(None)] * face.ndim
    z1[axis] = E + 1
    z1[-1] = slice(n2, None)
    out[tuple(z1)] = 0.0
    return out


def euler3d_metric(n: int, H: int, V: int, panel: int, device, seed: int = 20250824, damping: bool = False
                   ) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000 * panel + 1)
    n2, n3 = n * n, n**3
    pts = (V, H, H, n3)
    m = {}
    m["sqrtG"] = 1e9 * (1.0 + 0.1 * _u(pts, gen, device))
    h = torch.empty((3, 3) + pts, dtype=torch.float64, device=device)
    diag = (4e-10, 4e-10, 6e-7)
    for a in range(3):
        h[a, a] = diag[a] * (1.0 + 0.1 * _u(pts, gen, device))
    for a, b in ((0, 1), (0, 2), (1, 2)):
        h[a, b] = 1e-11 * _u(pts, gen, device)
        h[b, a] = h[a, b]
    m["h_contra"] = h
    m["christoffel"] = 1e-3 * _u((3, 9) + pts, gen, device)
    m["inv_dzdeta"] = torch.full(pts, 1.6e-3, dtype=torch.float64, device=device)
    for name, axis, fshape, pshape in (
        ("i", 2, (V, H, H + 1, n2), None),
        ("j", 1, (V, H + 1, H, n2), None),
        ("k", 0, (V + 1, H, H, n2), None),
    ):
        sg = 1e9 * (1.0 + 0.1 * _u(fshape, gen, device))
        m[f"sqrtG_itf_{name}"] = _pad_itf(sg, axis, n2).contiguous()
        hf = torch.empty((3, 3) + fshape, dtype=torch.float64, device=device)
        for a in range(3):
            hf[a, a] = diag[a] * (1.0 + 0.1 * _u(fshape, gen, device))
        for a, b in ((0, 1), (0, 2), (1, 2)):
            hf[a, b] = 1e-11 * _u(fshape, gen, device)
            hf[b, a] = hf[a, b]
        m[f"h_contra_itf_{name}"] = _pad_itf(hf, axis + 2, n2).contiguous()
    x = (torch.arange(H * n, dtype=torch.float64, device=device) + 0.5) / (H * n) * (math.pi / 2) - math.pi / 4
    m["boundary_sn"] = torch.tan(x).contiguous()
    m["boundary_we"] = torch.tan(x).contiguous()
    if damping:
        m["damp_coef"] = 0.02 * (_u(pts, gen, device) > 0.5).to(torch.float64)
        m["damp_uref"] = 1e-6 * (20.0 + _u((3,) + pts, gen, device))
    return m


def euler3d_state(n: int, H: int, V: int, panel: int, device, seed: int = 20250824, ztop: float = 10000.0,
                  row: int = 0, col: int = 0, k: int = 1) -> torch.Tensor:
    """State of one tile: H x H x V elements.  With k > 1 the tile is the (row, col) one of the k x k tiles of `panel`
    (row along x2, col along x1) and holds the corresponding cut of the PANEL's state - the same sphere whatever the
    decomposition, so that results of different decompositions can be compared (bench.py: checksum)."""
    if k > 1:
        full = euler3d_state(n, H * k, V, panel, device, seed, ztop)
        return full[:, :, row * H:(row + 1) * H, col * H:(col + 1) * H, :].contiguous()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000 * panel + 2)
    n3 = n**3
    pts = (V, H, H, n3)
    # height of each point: level index = ek*n + kl
    ek = torch.arange(V, dtype=torch.float64, device=device).view(V, 1, 1, 1)
    kl = (torch.arange(n3, device=device) // (n * n)).to(torch.float64).view(1, 1, 1, n3)
    z = ztop * (ek * n + kl + 0.5) / (n * V)
    q = torch.empty((5,) + pts, dtype=torch.float64, device=device)
    rho = 1.2 * torch.exp(-z / 8000.0) * (1.0 + 0.01 * _u(pts, gen, device))
    q[0] = rho
    q[1] = rho * 1e-6 * (20.0 + _u(pts, gen, device))
    q[2] = rho * 1e-6 * (20.0 + _u(pts, gen, device))
    q[3] = rho * 1e-7 * _u(pts, gen, device)
    q[4] = rho * 300.0 * (1.0 + 0.01 * _u(pts, gen, device))
    return q


def make_level_invariant(m, n: int, H: int, V: int):
    """Overwrite the synthetic 3-D metric `m` (euler3d_metric) in place with its first level / first vertical node, so that
    it is the same on all levels as a shallow atmosphere's without topography is: the case the column form of the plan
    (Euler3DPlan(column_metric=...)) is for.  Development benchmarks only."""
    n2 = n * n
    for k in ("sqrtG", "h_contra", "christoffel", "inv_dzdeta"):
        v = m[k].view(*m[k].shape[:-4], V, H, H, n, n2)
        v.copy_(v[..., 0:1, :, :, 0:1, :].clone().expand_as(v))
    for k, hh in (("sqrtG_itf_i", (H, H + 2)), ("h_contra_itf_i", (H, H + 2)), ("sqrtG_itf_j", (H + 2, H)),
                  ("h_contra_itf_j", (H + 2, H))):
        v = m[k].view(*m[k].shape[:-4], V, hh[0], hh[1], 2, n, n)
        v.copy_(v[..., 0:1, :, :, :, 0:1, :].clone().expand_as(v))
    for k in ("sqrtG_itf_k", "h_contra_itf_k"):
        v = m[k].view(*m[k].shape[:-4], V + 2, H, H, 2, n2)
        v.copy_(v[..., 1:2, :, :, 0:1, :].clone().expand_as(v))
    return m


def _pad_itf2(face: torch.Tensor, axis: int, n: int) -> torch.Tensor:
    """2-D twin of _pad_itf; the shallow-water interface metric keeps its values in the outermost
    slots too (geometry/metric2d.py), harmless either way."""
    E = face.shape[axis] - 1
    shape = list(face.shape)
    shape[axis] = E + 2
    shape[-1] = 2 * n
    out = torch.zeros(shape, dtype=face.dtype, device=face.device)
    im = [slice(None)] * face.ndim
    ip = [slice(None)] * face.ndim
    im[axis], im[-1] = slice(1, None), slice(0, n)
    ip[axis], ip[-1] = slice(0, E + 1), slice(n, None)
    out[tuple(im)] = face
    out[tuple(ip)] = face
    return out


def sw_metric(n: int, H: int, panel: int, device, seed: int = 20250824, topo: bool = False) -> Dict[str, torch.Tensor]:
    """S7-style synthetic shallow-water metric: sqrtG ~ 1e9, H_contra ~ 4e-10 SPD,
    Christoffel ~ 1e-2 xi."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000 * panel + 11)
    pts = (H, H, n * n)
    m = {"sqrtG": 1e9 * (1.0 + 0.1 * _u(pts, gen, device))}
    m["H_contra_11"] = 4e-10 * (1.0 + 0.1 * _u(pts, gen, device))
    m["H_contra_22"] = 4e-10 * (1.0 + 0.1 * _u(pts, gen, device))
    m["H_contra_12"] = 1e-11 * _u(pts, gen, device)
    m["H_contra_21"] = m["H_contra_12"].clone()
    for k in ("1_01", "1_02", "1_11", "1_12", "2_01", "2_02", "2_12", "2_22"):
        m["christoffel_" + k] = 1e-2 * _u(pts, gen, device)
    fi, fj = (H, H + 1, n), (H + 1, H, n)
    m["sqrtG_itf_i"] = _pad_itf2(1e9 * (1.0 + 0.1 * _u(fi, gen, device)), 1, n).contiguous()
    m["sqrtG_itf_j"] = _pad_itf2(1e9 * (1.0 + 0.1 * _u(fj, gen, device)), 0, n).contiguous()
    m["H_contra_11_itf_i"] = _pad_itf2(4e-10 * (1.0 + 0.1 * _u(fi, gen, device)), 1, n).contiguous()
    m["H_contra_21_itf_i"] = _pad_itf2(1e-11 * _u(fi, gen, device), 1, n).contiguous()
    m["H_contra_22_itf_j"] = _pad_itf2(4e-10 * (1.0 + 0.1 * _u(fj, gen, device)), 0, n).contiguous()
    m["H_contra_12_itf_j"] = _pad_itf2(1e-11 * _u(fj, gen, device), 0, n).contiguous()
    x = (torch.arange(H * n, dtype=torch.float64, device=device) + 0.5) / (H * n) * (math.pi / 2) - math.pi / 4
    m["boundary_sn"] = torch.tan(x).contiguous()
    m["boundary_we"] = torch.tan(x).contiguous()
    if topo:
        m["hsurf"] = 500.0 * (1.0 + _u(pts, gen, device))
        m["dzdx1"] = 1e2 * _u(pts, gen, device)
        m["dzdx2"] = 1e2 * _u(pts, gen, device)
        m["hsurf_itf_i"] = _pad_itf2(500.0 * (1.0 + _u(fi, gen, device)), 1, n).contiguous()
        m["hsurf_itf_j"] = _pad_itf2(500.0 * (1.0 + _u(fj, gen, device)), 0, n).contiguous()
    return m


def sw_state(n: int, H: int, panel: int, device, seed: int = 20250824) -> torch.Tensor:
    """h = 8000 (1 + 0.05 xi), hu^i = h 1e-6 (30 + 10 xi)   """
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000 * panel + 12)
    pts = (H, H, n * n)
    q = torch.empty((3,) + pts, dtype=torch.float64, device=device)
    q[0] = 8000.0 * (1.0 + 0.05 * _u(pts, gen, device))
    q[1] = q[0] * 1e-6 * (30.0 + 10.0 * _u(pts, gen, device))
    q[2] = q[0] * 1e-6 * (30.0 + 10.0 * _u(pts, gen, device))
    return q


def dfr_ops(n: int):
    """1-D DFR operator pieces on Gauss-Legendre points, computed from their definitions
    (reference geometry/operators.py:55-80, 86-99, 144-148 builds the same objects with sympy):
    Lagrange extrapolation to -1/+1, the differentiation matrix on the extended point set
    [-1, GL nodes, +1] restricted to solution points (diff_solpt) with its two boundary columns
    (correction), and the modal filter removing the highest Legendre mode."""
    import numpy as np
    from numpy.polynomial import legendre as L

    x, _ = L.leggauss(n)
    ext = np.concatenate(([-1.0], x, [1.0]))

    def bary_diff(pts):
        m = len(pts)
        w = np.array([1.0 / np.prod([pts[j] - pts[k] for k in range(m) if k != j]) for j in range(m)])
        Dm = np.zeros((m, m))
        for i in range(m):
            for j in range(m):
                if i != j:
                    Dm[i, j] = (w[j] / w[i]) / (pts[i] - pts[j])
            Dm[i, i] = -Dm[i].sum()
        return Dm

    Dext = bary_diff(ext)
    Vd = L.legvander(x, n - 1)
    invV = np.linalg.inv(Vd)
    em = (L.legvander(np.array([-1.0]), n - 1) @ invV).reshape(-1)
    ep = (L.legvander(np.array([1.0]), n - 1) @ invV).reshape(-1)
    feye = np.eye(n)
    feye[-1, -1] = 0.0
    return {
        "extrap_neg": em, "extrap_pos": ep,
        "diff_solpt": np.ascontiguousarray(Dext[1:-1, 1:-1]),
        "correction": np.ascontiguousarray(np.column_stack((Dext[1:-1, 0], Dext[1:-1, -1]))),
        "highfilter": Vd @ (feye @ invV),
    }
