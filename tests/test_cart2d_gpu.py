"""The reference's compiled `pde` module functions and the 2-D Cartesian RHS on the GPU, against
golden vectors produced by the reference (its Python RHS driving its native pde_cpp kernels)."""
import numpy as np
import pytest
import torch

from tests.util import CART2D_FIXTURES, golden_cart, var_err, var_max

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", CART2D_FIXTURES)
def test_legacy_pde_functions(name, built_lib):
    from wxfactory_amd.device import HipDevice

    dev = HipDevice()
    g = golden_cart(name)
    q = dev.array(g.q())
    f1, f3 = torch.zeros_like(q), torch.zeros_like(q)
    dev.pde.pointwise_eulercartesian_2d(q, f1, f3, g.nx, g.nz, g.n**2)
    qi1, qi3 = dev.array(g["phase/q_itf_x1"]), dev.array(g["phase/q_itf_x3"])
    r1, r3 = torch.zeros_like(qi1), torch.zeros_like(qi3)
    dev.pde.riemann_eulercartesian_ausm_2d(qi1, qi3, r1, r3, g.nx, g.nz, g.n)
    dev.synchronize()
    for got, key, tol in ((f1, "f_x1", 1e-14), (f3, "f_x3", 1e-14), (r1, "f_itf_x1", 1e-13), (r3, "f_itf_x3", 1e-13)):
        ref = g["phase/" + key]
        assert np.abs(dev.to_host(got) - ref).max() <= tol * np.abs(ref).max(), key
    with pytest.raises(TypeError):
        dev.pde.pointwise_eulercartesian_2d(q.float(), f1.float(), f3.float(), g.nx, g.nz, g.n**2)


def test_forcing_euler_cubesphere_3d(built_lib):
    """HIP twin of the reference's forcing kernel against the Christoffel part of the reference's
    forcing phase (pde_euler_cubesphere.py:12-25 evaluates the same formula in Python)."""
    from tests.util import golden
    from wxfactory_amd.device import HipDevice

    dev = HipDevice()
    g = golden("euler3d_c31_n3_h4_v2")
    p = 4
    q = dev.array(g.q(p))
    pres = dev.array(g[f"p{p}/phase/pressure"])
    m = g.metric(p)
    forcing = torch.zeros_like(q)
    dev.pde.forcing_euler_cubesphere_3d(q, pres, dev.array(m["sqrtG_new"]), dev.array(m["h_contra_new"]),
                                        dev.array(m["christoffel"]), forcing, g.H, g.H, g.V, g.n**3, 0)
    dev.synchronize()
    ref = g[f"p{p}/phase/forcing"].copy()
    # the reference's phase array also carries the gravity term on the rho-w row: remove it
    from oracle.euler3d import gravity
    from tests.util import make_oracle

    o = make_oracle(g, p)
    ref[3] -= m["inv_dzdeta_new"] * gravity * m["inv_sqrtG_new"] * o.highfilter_k(m["sqrtG_new"] * g.q(p)[0])
    got = dev.to_host(forcing)
    assert np.abs(got[0]).max() == 0 and np.abs(got[4]).max() == 0
    for v in (1, 2, 3):
        assert np.abs(got[v] - ref[v]).max() <= 1e-11 * np.abs(g[f"p{p}/phase/forcing"][v]).max(), v


@pytest.mark.parametrize("name", CART2D_FIXTURES)
@pytest.mark.parametrize("cplx", [False, True])
def test_fused_rhs(name, cplx, built_lib):
    from wxfactory_amd.rhs_cart2d import RhsCart2D

    g = golden_cart(name)
    w = {}
    g.oracle().rhs(g.q(), want=w)
    scale = np.maximum(var_max(g.r()), np.maximum(var_max(w["d1"]), var_max(w["d3"])))
    dtype = torch.complex128 if cplx else torch.float64
    rhs = RhsCart2D(g.n, g.nx, g.nz, g.dx1, g.dx3, g.ops, DEV, dtype)
    R = rhs(_dev(g.q(cplx))).cpu().numpy()
    ref = g.r(cplx)
    assert (var_err(R.real, ref.real) <= 1e-10 * scale).all(), var_err(R.real, ref.real) / scale
    if cplx:
        assert (var_err(R.imag, ref.imag) <= 1e-10 * var_max(ref.imag)).all()
    rhs.close()


@pytest.mark.parametrize("fused", [True, False])
def test_rk3_time_loop_matches_reference(fused, built_lib):
    """BASELINE config 1's explicit time loop (config/gaussian_bubble.ini on a small grid: Tvdrk3.step + apply_filters,
    simulation.py:147-155, on the 2-D Cartesian RHS with the reference's compiled pde kernels) against the reference's own
    run: the state after one and after five steps."""
    from wxfactory_amd.integrators import Tvdrk3
    from wxfactory_amd.rhs_cart2d import RhsCart2D

    g = golden_cart("cart2d_rk3_bubble_n4")
    rhs = RhsCart2D(g.n, g.nx, g.nz, g.dx1, g.dx3, g.ops, DEV)
    stepper = Tvdrk3(rhs, fused=fused, pipeline=False)
    dt, nsteps = float(g["meta/rk3_dt"]), int(g["meta/rk3_steps"])
    Q = _dev(g["Q"])
    for i in range(nsteps):
        Q = stepper.step(Q, dt)
        if i in (0, nsteps - 1):
            ref = g["rk3_1" if i == 0 else "rk3_n"]
            moved = np.abs(ref - g["Q"]).max(axis=(1, 2, 3))
            err = np.abs(Q.cpu().numpy() - ref).max(axis=(1, 2, 3))
            assert (moved > 0).all() and (err <= 1e-9 * moved + 1e-14 * np.abs(ref).max(axis=(1, 2, 3))).all(), (i, err, moved)
    rhs.close()


def test_shipped_bubble_integrator_epi2_with_pmex(built_lib):
    """config/gaussian_bubble.ini's own integrator - epi2, the schema's default exponential solver pmex, complex-step JVP,
    dt = 5 s (integrators/epi.py + solvers/pmex.py on the 2-D Cartesian RHS) - against the reference's own run on a small
    grid: the state after each of three steps and pmex's statistics of every step (5 sub-steps, 320 vectors each)."""
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.rhs_cart2d import RhsCart2D

    g = golden_cart("cart2d_epi2_pmex_bubble_n4")
    assert str(g["meta/epi_solver"]) == "pmex"
    rhs = RhsCart2D(g.n, g.nx, g.nz, g.dx1, g.dx3, g.ops, DEV)
    stepper = Epi(int(g["meta/epi_order"]), rhs, tol=float(g["meta/epi_tol"]), exponential_solver="pmex")
    dt, nsteps = float(g["meta/epi_dt"]), int(g["meta/epi_steps"])
    ref_stats = g["meta/epi_solver_stats"]
    Q, prev = _dev(g["Q"]), g["Q"]
    for i in range(nsteps):
        Q = stepper.step(Q, dt)
        info = stepper.solver_info
        got = [info[k] for k in ("substeps", "rejected", "iterations", "exps", "krylov_size", "own_norms")]
        assert got == [int(ref_stats[i][k]) for k in (0, 1, 2, 3, 5, 6)], (i, got, ref_stats[i])
        ref = g[f"epi_{i + 1}"]
        moved = np.abs(ref - prev).max(axis=(1, 2, 3))
        err = np.abs(Q.cpu().numpy() - ref).max(axis=(1, 2, 3))
        assert (moved > 0).all() and (err <= 1e-6 * moved).all(), (i, err, moved)
        prev = ref
    rhs.close()
