"""Full-size cross-check: one whole E7 panel (n = 8, 60 x 60 x 8 elements, 14.7 M points - the benchmark's own size,
BASELINE.json's metric line) through the HIP kernels against the C++ restatement of the reference's algorithm
(oracle/c/euler3d_port.cpp, itself pinned to the reference's golden vectors in float64 AND complex128 by
tests/test_oracle_c.py), on ALL FIVE rows of R and of the complex-step Jacobian-vector product.

What the small fixtures cannot show: the index arithmetic at offsets of gigabytes (the 27 Christoffel fields of one
panel span 3.2 GB), the element / region decode at H = 60, the halo addressing of 60-element edges, the INTERIOR +
BOUNDARY split at the launch shape of the multi-GPU runs.  Geometry, metric and initial state are the product's own
(geometry3d, initial: DCMIP 3-1 + a seeded 1 % perturbation, so that R is O(1) and the bound is not about cancellation).
The same at the size of the reference's own benchmark matrix (6.48 M DOF; `extra.rhs_benchmark_matrix`) for the forms the low
orders take: n = 2 in the one-kernel form (bricks cut by the tile edge), n = 4 in the batched two-kernel form.
Reference: rhs/rhs_dfr.py:48-313, solvers/matvec.py:56-61.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CHECKS = {31: 4, 21: 0}   # the panel compared (panel 4: its N and W edges are flipped, all four rotated; panel 0: the mountain's)
# case 31: DCMIP 3-1 (config 4's physics; a shallow atmosphere without topography, non-rotating: 18 Christoffel fields);
# case 21: DCMIP 2-1 (config 5's physics: the Schaer mountain - a metric that depends on the level - and the Rayleigh sponge above
# 20 km, pde_euler_cubesphere.py:285-288 + init/dcmip.py:676-757: the has_damp loads, +32 B/point)
CASES = {31: 10000.0, 21: 30000.0}


def _ref_metric(m):
    """geometry3d's device metric of one panel -> the reference's names on the host (what the oracle takes)."""
    om = {"sqrtG_new": m["sqrtG"], "h_contra_new": m["h_contra"], "christoffel": m["christoffel"],
          "inv_dzdeta_new": m["inv_dzdeta"]}
    for k in ("damp_coef", "damp_uref"):
        if k in m:
            om[k] = m[k]
    for d in "ijk":
        om[f"sqrtG_itf_{d}_new"] = m[f"sqrtG_itf_{d}"]
        om[f"h_contra_itf_{d}_new"] = m[f"h_contra_itf_{d}"]
    return {k: v.cpu().numpy() for k, v in om.items()}


# (order, elements per side, levels, case): the E7 panels, and the reference's benchmark matrix (tests/rhs_benchmark/run.sh:67-71,
# 6.48 M DOF) at two of its orders - n = 2 through the ONE-kernel form on bricks cut by the tile edge (30 = 3 x 8 + 6), n = 4 through
# the batched two-kernel form: the sizes `extra.rhs_benchmark_matrix` times
SIZES = [(8, 60, 8, 31), (8, 60, 8, 21), (2, 30, 30, 31), (4, 15, 15, 31)]


@pytest.mark.parametrize("N,H,V,CASE", SIZES)
def test_e7_panel_all_rows_against_the_c_port(built_lib, N, H, V, CASE):
    from oracle import cubed_sphere as cs
    from oracle.c_port import Euler3DPortC
    from wxfactory_amd import _lib
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.initial import initial_state
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    CHECK = CHECKS[CASE]
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    ops = dfr_ops(N)
    plans, metrics, qs, bnd = {}, {}, [], {}
    gen = torch.Generator(device=DEV).manual_seed(2025)
    topo = topography_for_case(CASE, planet_for_case(CASE)[0])
    for p in range(6):
        t = CubedSphere3DTile(N, H, V, p, CASES[CASE], CASE, topo=topo)
        metrics[p] = metric3d_torch(t, DEV)
        assert ("damp_coef" in metrics[p]) == (CASE == 21)
        plans[p] = Euler3DPlan(N, H, V, CASE, p, ops, metrics[p])
        q = torch.from_numpy(np.array(initial_state(t))).to(DEV)
        qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=DEV, dtype=q.dtype) - 0.5)))
        b = metrics[p]["boundary_sn"].cpu().numpy().reshape(H, 1, N)
        bnd[p] = (np.tile(b, (1, N, 1)), np.tile(metrics[p]["boundary_we"].cpu().numpy().reshape(H, 1, N), (1, N, 1)))
    assert int(plans[0].lib.wx_euler3d_uses_matrix_cores(plans[0]._h, _lib.WX_KERNEL_RHS)) == (1 if N == 8 else 0)
    assert bool(plans[0].one_kernel) == (N == 2)
    # compulsory bytes per point of this plan at n = 8: 312 (case 31, no rotation symbols) / 344 (case 21: + the four sponge fields)
    if N == 8:
        assert plans[0].bytes_per_point == (312.0 if CASE == 31 else 344.0)
    if CASE == 21:   # the mountain makes the metric depend on the level: the general kernel's case, not the column form's
        h13 = metrics[CHECK]["h_contra"][0, 2]   # (terrain-following levels: dz/dx at fixed eta fades with height)
        assert float((h13[0] - h13[-1]).abs().max()) > 1e-3 * float(h13.abs().max()) > 0.0
        assert float(metrics[CHECK]["damp_coef"].abs().max()) > 0.0
    Q = torch.stack(qs)
    del qs
    rhs = RhsEuler3D(plans)
    assert rhs._small_tiles() == (N != 8)   # per-panel launches: bench.py's path; one launch per phase for all panels: the matrix's
    R = rhs(Q)
    torch.cuda.synchronize()

    # INTERIOR + BOUNDARY on the checked panel, with the halos the exchange holds: bit for bit the ALL launch
    ex = rhs.exchange_for(torch.float64)
    split = torch.full_like(Q[CHECK], float("nan"))
    plans[CHECK].rhs(Q[CHECK], None, split, _lib.WX_REGION_INTERIOR)
    plans[CHECK].rhs(Q[CHECK], ex.halo_views(CHECK), split, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(split, R[CHECK])
    del split

    # ---- the oracle's side: faces of all six panels, routed edges, then the checked panel's R
    qh = Q.cpu().numpy()
    ports = {p: Euler3DPortC(N, H, V, CASE, ops, {}, bnd[p][0], bnd[p][1], panel=p, threads=threads) for p in range(6)}
    sends = [ports[p].pack_edges(ports[p].extrapolate(qh[p])) for p in range(6)]
    halo = cs.route(sends)[CHECK]
    # what the kernels packed and exchanged for the checked panel is what the oracle routes to it
    for e in range(4):
        got = ex.halo_views(CHECK)[e].cpu().numpy().reshape(halo[e].shape)
        assert np.abs(got - halo[e]).max() <= 1e-13 * np.abs(halo[e]).max(), e
    port = Euler3DPortC(N, H, V, CASE, ops, _ref_metric(metrics[CHECK]), bnd[CHECK][0], bnd[CHECK][1], panel=CHECK,
                        threads=threads)
    ref = port.rhs(qh[CHECK], halo)
    got = R[CHECK].cpu().numpy()
    ax = (1, 2, 3, 4)
    rmax = np.abs(ref).max(axis=ax)
    err = np.abs(got - ref).max(axis=ax)
    assert np.isfinite(got).all() and (rmax > 0).all()
    assert (err <= 1e-10 * rmax).all(), err / rmax   # all five rows: rho, rho u1, rho u2, rho w, rho theta

    # ---- the complex-step Jacobian-vector product of the same panel (unprepared and prepared: identical bits)
    v = (torch.rand(Q.shape, generator=gen, device=DEV, dtype=Q.dtype) - 0.5) * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True) * 1e-3
    eps = float(np.sqrt(np.finfo(float).eps))
    dual = rhs._jvp_plans()[CHECK]
    assert N != 8 or int(dual.lib.wx_euler3d_uses_matrix_cores(dual._h, _lib.WX_KERNEL_JVP)) == 1 or os.environ.get("WXHIP_JVP_LEAN") == "0"
    J = rhs.jvp(Q, v, eps, 1.0 / eps)
    if N == 8:   # (the prepared form is the large tiles')
        assert rhs.jvp_prepare(Q)
        Jp = rhs.jvp(Q, v, eps, 1.0 / eps)
        rhs.jvp_release()
        torch.cuda.synchronize()
        assert torch.equal(J, Jp)
        del Jp
    vh = v.cpu().numpy()
    sends_c = {}
    for p in [CHECK] + [cs.NEIGHBOR[CHECK][e] for e in range(4)]:
        if p not in sends_c:
            qc = qh[p] + 1j * eps * vh[p]
            sends_c[p] = ports[p].pack_edges(ports[p].extrapolate(qc))
    halo_c = [None] * 4
    for p, sd in sends_c.items():
        for e in range(4):
            if cs.NEIGHBOR[p][e] == CHECK:
                halo_c[cs.landing_edge(p, e)] = sd[e]
    assert all(h is not None for h in halo_c)
    jref = port.rhs(qh[CHECK] + 1j * eps * vh[CHECK], halo_c).imag / eps
    jgot = J[CHECK].cpu().numpy()
    jmax = np.abs(jref).max(axis=ax)
    jerr = np.abs(jgot - jref).max(axis=ax)
    assert np.isfinite(jgot).all() and (jmax > 0).all()
    assert (jerr <= 1e-10 * jmax).all(), jerr / jmax
