"""pmex - the schema's default `exponential_solver` (config-format.json; config/case6.ini, density_current.ini) - on
the GPU against the reference's own solvers/pmex.py and integrators/epi.py, run on the states of the callers fixtures
(tests/golden/pmex_euler3d_*.npz, made by oracle/refharness/gen_golden.py: the inputs are those of
callers_euler3d_*.npz, checked by max|Q|, max|R| per panel and variable).  n = 3, and the benchmark order n = 8 (the
matrix-core JVP kernel under the solver)."""
import os

import numpy as np
import pytest
import torch

from tests.util import Golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
AX = (0, 2, 3, 4, 5)


@pytest.fixture(scope="module", params=["n3_h3_v2", "n8_h2_v2"])
def setup(built_lib, request):
    from tests.gpu_util import device_metric
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("callers_euler3d_" + request.param)
    px = np.load(os.path.join(os.path.dirname(__file__), "golden", f"pmex_euler3d_{request.param}.npz"))
    assert str(px["meta/base"]) == "callers_euler3d_" + request.param
    for p in range(6):   # same inputs as the callers fixture
        assert np.array_equal(px[f"p{p}/Q_absmax"], np.abs(g[f"p{p}/Q"]).max(axis=(1, 2, 3, 4)))
        assert np.array_equal(px[f"p{p}/R_absmax"], np.abs(g[f"p{p}/R"]).max(axis=(1, 2, 3, 4)))
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
    rhs = RhsEuler3D(plans)
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    pstack = lambda key: np.stack([px[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    return g, px, rhs, stack, pstack


def _same_decisions(stats, ref, tol):
    # sub-steps, rejections, Krylov vectors, exponentials, last basis size, norms that needed their own reduction
    assert [int(stats[i]) for i in (0, 1, 2, 3, 5, 6)] == [int(ref[i]) for i in (0, 1, 2, 3, 5, 6)], (stats, ref.tolist())
    # the accumulated error estimate: 5 %, or 1 % of the tolerance where it sits far below it (at n = 8 the accepted
    # sub-steps carry estimates of 1e-10 for a tolerance of 1e-7: the last entries of the exponential, rounding-sized)
    assert abs(float(stats[4]) - float(ref[4])) <= max(0.05 * float(ref[4]), 0.01 * tol), (stats, ref.tolist())


def test_pmex_phi1_as_epi_calls_it(setup):
    """phi_1(dt J) R with the complex-step JVP, called as integrators/epi.py:314-315 calls it."""
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.solvers import pmex

    g, px, rhs, stack, pstack = setup
    Q, R = stack("Q"), stack("R")
    dt = float(px["meta/dt_jvp"])
    vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
    vec[1] = R.flatten()
    phiv, stats = pmex([1.0], ComplexStepOperator(dt, Q, R, rhs, "complex"), vec, tol=1e-7, mmax=64, task1=False)
    _same_decisions(stats, px["p0/pmex_stats"], 1e-7)
    ref = pstack("pmex_phiv")
    err = np.abs(phiv.cpu().numpy().reshape(ref.shape) - ref).max(axis=AX) / np.abs(ref).max(axis=AX)
    assert (err < 1e-8).all(), err


def test_pmex_three_rows_and_intermediate_times(setup):
    """u with three rows (phi_0 u_0 + phi_1 u_1 + phi_2 u_2: two augmented components), three output times, task1."""
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.solvers import pmex

    g, px, rhs, stack, pstack = setup
    Q, R, V = stack("Q"), stack("R"), stack("V")
    dt = float(px["meta/dt_jvp"])
    A = lambda x: matvec_fun(x, dt, Q, R, rhs, "complex")  # noqa: E731
    vec = torch.zeros((3, R.numel()), dtype=torch.float64, device=DEV)
    vec[0] = V.flatten()
    vec[1] = R.flatten()
    vec[2] = 0.01 * A(V.flatten()).flatten()
    w, stats = pmex([0.25, 0.6, 1.0], A, vec, tol=1e-9, m_init=6, mmin=6, mmax=40, task1=True)
    _same_decisions(stats, px["p0/pmex3_stats"], 1e-9)
    ref = np.stack([px[f"p{p}/pmex3_w"] for p in range(6)], axis=1)   # (time, panel, variable, ...)
    got = w.cpu().numpy().reshape(ref.shape)
    for k in range(3):
        # How closely can two runs agree?  tools/pmex_sensitivity.py (profiles/r03_pmex_sensitivity.log) repeats this very
        # call with every product of the operator multiplied by (1 + 1e-13 xi): same decisions, and results that differ
        # from the unperturbed run by 2e-11 / 4e-7 / 2e-7 in the 2-norm at the three times (|w| = 84 / 31 / 19) at n = 8
        # - the stiff operator amplifies the last digits of the products over 8 sub-steps and 320 vectors.  The
        # reference's run (NumPy's complex arithmetic in the complex step, ours: dual numbers) differs from ours by
        # 6e-10 / 5e-7 / 3e-7: the same thing.  Bound: 1e-7 of the result's norm, 1e-6 of each variable's maximum.
        dn, rn = np.linalg.norm(got[k] - ref[k]), np.linalg.norm(ref[k])
        assert dn <= 1e-7 * rn, (k, dn, rn)
        err = np.abs(got[k] - ref[k]).max(axis=AX) / np.abs(ref[k]).max(axis=AX)
        assert (err < 1e-6).all(), (k, err)


def test_epi2_step_with_pmex(setup):
    from wxfactory_amd.integrators import Epi

    g, px, rhs, stack, pstack = setup
    Q = stack("Q")
    stepper = Epi(2, rhs, tol=1e-7, exponential_solver="pmex")
    Qn = stepper.step(Q, float(px["meta/dt_jvp"]))
    ref_stats = px["p0/pmex_stats"]
    assert stepper.solver_info["iterations"] == int(ref_stats[2]) and stepper.solver_info["rejected"] == int(ref_stats[1])
    refq, q0 = pstack("epi2_pmex"), Q.cpu().numpy()
    upd = np.abs(refq - q0).max(axis=AX)
    assert (np.abs(Qn.cpu().numpy() - refq).max(axis=AX) <= 1e-7 * upd).all()
    with pytest.raises(ValueError, match="Unrecognized exponential solver"):
        Epi(2, rhs, exponential_solver="exode")


@pytest.mark.parametrize("p,taus", [(1, [1.0]), (3, [0.4, 1.0])])
def test_pmex_device_kernels_split_form_and_exact_result(built_lib, p, taus, monkeypatch):
    """pmex on device vectors (wx_krylov_aug_update, wx_multi_dot2, wx_multi_axpy_scaled) against the same algorithm on CPU
    tensors (torch expressions), the several-rank form of its reductions taken on one rank (products of the n-long parts
    all-reduced, the replicated augmented components added once afterwards), and the exact result of a diagonal operator."""
    import math

    from wxfactory_amd.solvers import pmex

    n = 50_000
    gen = torch.Generator(device=DEV).manual_seed(23 + p)
    lam = -(0.2 + 2.5 * torch.rand(n, generator=gen, device=DEV, dtype=torch.float64))
    u = torch.randn((p + 1, n), generator=gen, device=DEV, dtype=torch.float64)
    A = lambda v: lam * v  # noqa: E731
    args = dict(tol=1e-10, m_init=12, mmin=10, mmax=40)
    w_dev, st_dev = pmex(taus, A, u, **args)
    lam_c, u_c = lam.cpu(), u.cpu()
    w_cpu, st_cpu = pmex(taus, lambda v: lam_c * v, u_c, **args)
    assert st_dev[:4] == st_cpu[:4] and st_dev[5:] == st_cpu[5:], (st_dev, st_cpu)
    scale = float(w_cpu.abs().max())
    assert float((w_dev.cpu() - w_cpu).abs().max()) <= 1e-11 * scale
    w_split, st_split = pmex(taus, A, u, _force_split=True, **args)
    assert st_split[:4] == st_dev[:4] and float((w_split - w_dev).abs().max()) <= 1e-12 * scale
    # the default at this length is the vector in four fused launches (pmex_aug_dot2_kernel ...); the row-batched launches that
    # long vectors take: the same decisions, the same vectors to rounding (the products are summed in other groups)
    monkeypatch.setenv("WXHIP_PMEX_FUSED", "0")
    w_rows, st_rows = pmex(taus, A, u, **args)
    monkeypatch.delenv("WXHIP_PMEX_FUSED")
    assert st_rows[:4] == st_dev[:4] and st_rows[5:] == st_dev[5:], (st_rows, st_dev)
    assert float((w_rows - w_dev).abs().max()) <= 1e-12 * scale
    # ... and the one-rank form with a host round trip per vector (the projector on the host, as the reference has it)
    # against the default, whose vectors are built by wx_pmex_vector with none
    monkeypatch.setenv("WXHIP_PMEX_DEVICE", "0")
    w_host, st_host = pmex(taus, A, u, **args)
    w_hsplit, st_hsplit = pmex(taus, A, u, _force_split=True, **args)   # the several-rank form with the host in between
    monkeypatch.delenv("WXHIP_PMEX_DEVICE")
    assert st_hsplit[:4] == st_dev[:4] and float((w_hsplit - w_split).abs().max()) <= 1e-12 * scale
    assert st_host[:4] == st_dev[:4] and st_host[5:] == st_dev[5:], (st_host, st_dev)
    assert float((w_host - w_dev).abs().max()) <= 1e-12 * scale

    def phi(k, z):
        if k == 0:
            return torch.exp(z)
        return (phi(k - 1, z) - 1.0 / math.factorial(k - 1)) / z

    for i, tau in enumerate(taus):   # exact: w(tau) = sum_k tau^k phi_k(tau lam) u_k
        ref = sum((tau ** k) * phi(k, tau * lam) * u[k] for k in range(p + 1))
        assert float((w_dev[i] - ref).abs().max()) <= 1e-8 * float(ref.abs().max())


@pytest.mark.parametrize("solver", ["kiops", "pmex"])
def test_krylov_basis_is_sized_to_the_free_memory(built_lib, solver, monkeypatch):
    """The reference asks for mmax = 64 whatever the problem size (integrators/epi.py:315, 334); 65 vectors of the whole E7
    sphere are 230 GB.  The solvers take the largest basis that fits (all ranks the same), say so, and the controller
    works within it; when not even mmin fits they raise."""
    from wxfactory_amd import solvers

    n = 20_000
    gen = torch.Generator(device=DEV).manual_seed(5)
    lam = -(0.2 + 40.0 * torch.rand(n, generator=gen, device=DEV, dtype=torch.float64))
    u = torch.randn((2, n), generator=gen, device=DEV, dtype=torch.float64)
    fn = getattr(solvers, solver)
    args = dict(tol=1e-10, m_init=30, mmin=10, mmax=64)
    w_full, st_full = fn([1.0], lambda v: lam * v, u, **args)
    row = (n + 1) * 8
    monkeypatch.setattr(solvers, "_BASIS_CHECK_BYTES", 0)   # (the check is skipped for bases under a gigabyte)
    real = torch.cuda.mem_get_info
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (27 * row, real(dev)[1]))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda dev=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 0)
    with pytest.warns(RuntimeWarning, match="limited to 18 vectors"):
        w_small, st_small = fn([1.0], lambda v: lam * v, u, **args)
    assert st_small[5] <= 18 and st_small[0] >= st_full[0]
    ref = (torch.exp(lam) * u[0] + (torch.exp(lam) - 1.0) / lam * u[1])
    for w in (w_full, w_small):
        assert float((w[0] - ref).abs().max()) <= 1e-8 * float(ref.abs().max())
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (12 * row, real(dev)[1]))
    with pytest.raises(MemoryError, match="mmin"):
        fn([1.0], lambda v: lam * v, u, **args)


def test_pmex_happy_breakdown_on_the_device(built_lib, monkeypatch):
    """An operator with five distinct eigenvalues: the Krylov space closes after six vectors.  The pass builds its vectors
    without a host round trip, so the breakdown is seen when the columns are read: the vectors past it are discarded and the
    result is exact (both with and without the round trip per vector)."""
    from wxfactory_amd.solvers import pmex

    n = 30_000
    gen = torch.Generator(device=DEV).manual_seed(99)
    lam = -torch.tensor([0.5, 1.0, 1.5, 2.0, 2.5], device=DEV, dtype=torch.float64)[torch.randint(0, 5, (n,), generator=gen, device=DEV)]
    u = torch.zeros((2, n), device=DEV, dtype=torch.float64)
    u[0] = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64)
    ref = torch.exp(lam) * u[0]
    for device_pass in ("1", "0"):
        monkeypatch.setenv("WXHIP_PMEX_DEVICE", device_pass)
        # breakdown="exact": the Hessenberg column of the vector that breaks down is kept (the reference stores it only
        # after its test, pmex.py:225-233 - breakdown="reference", the default)
        w, st = pmex([1.0], lambda v: lam * v, u, tol=1e-9, m_init=12, mmin=12, mmax=30, breakdown="exact")
        assert float((w[0] - ref).abs().max()) <= 1e-10 * float(ref.abs().max()), (device_pass, st)
        w_ref, st_ref = pmex([1.0], lambda v: lam * v, u, tol=1e-9, m_init=12, mmin=12, mmax=30)
        assert st_ref[:4] == st[:4]
        if st[2] == 12:
            # the breakdown was NOT seen: the norm of the sixth vector is estimated as sqrt(<w, w> - sum g_k^2) (pmex.py:194-218),
            # a difference of two numbers of size |w|^2 whose rounding noise, sqrt(eps) |w| ~ 1e-8 |w|, can land above the
            # tolerance - then the vector of noise is normalised and the pass runs to its end; every column past the fifth
            # weighs nothing and both forms are exact.  Which way it falls depends on how the products were summed.
            assert st[0] == 1 and st[1] == 0, st
            assert float((w_ref[0] - ref).abs().max()) <= 1e-10 * float(ref.abs().max()), (device_pass, st_ref)
            continue
        assert st[0] == 1 and st[1] == 0 and st[2] in (5, 6) and st[4] == 0.0, st   # (5 or 6: see tests/test_solvers_cpu.py)
        err = float((w_ref[0] - ref).abs().max()) / float(ref.abs().max())
        # seen at once (5 vectors): the reference's result misses the last column's projections; seen a vector of noise
        # later (6), that column weighs nothing and both forms are exact
        assert (1e-8 < err < 1e-2) if st[2] == 5 else err <= 1e-10, (device_pass, st, err)
