"""Host logic of the Krylov solvers (wxfactory_amd/solvers.py) on CPU tensors against dense linear algebra:
kiops (reference solvers/kiops.py:10-347) against the phi functions from one matrix exponential of the augmented
matrix, fgmres (solvers/fgmres.py:97-276) against numpy.linalg.solve; on one rank and with the vectors split over
two gloo ranks (the reference's MPI allreduce in global_operations.py:14-36)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy.linalg import expm


@pytest.fixture(autouse=True, scope="module")
def _library(built_lib):
    """fgmres' host-side loops are the library's (wx_fgmres_rotate_columns: plain host C, no GPU needed): build it first."""
    return built_lib


def _problem(n=96, p=3, seed=3):
    rng = np.random.default_rng(seed)
    A = -np.diag(rng.uniform(0.1, 6.0, n)) + 0.4 * rng.standard_normal((n, n)) / np.sqrt(n)
    A = A + 2.0 * (np.triu(rng.standard_normal((n, n)), 1) - np.triu(rng.standard_normal((n, n)), 1).T) / np.sqrt(n)
    u = rng.standard_normal((p + 1, n)) * (10.0 ** rng.uniform(-2, 2, (p + 1, 1)))
    return A, u


def _phi_exact(A, u, tau):
    """sum_k tau^k phi_k(tau A) u_k = [I 0] exp(tau [[A, B], [0, K]]) [u_0; e_p]  (Al-Mohy & Higham 2011, thm 2.1)"""
    n, p = A.shape[0], u.shape[0] - 1
    big = np.zeros((n + p, n + p))
    big[:n, :n] = A
    big[:n, n:] = u[:0:-1].T
    big[n:, n:] = np.diag(np.ones(p - 1), 1)
    v = np.concatenate([u[0], np.zeros(p)])
    v[-1] = 1.0
    return (expm(tau * big) @ v)[:n]


@pytest.mark.parametrize("p", [1, 3])
@pytest.mark.parametrize("taus", [[1.0], [0.25, 0.6, 1.0]])
def test_kiops_against_dense_phi(p, taus):
    from wxfactory_amd.solvers import kiops

    A, u = _problem(p=p)
    At = torch.from_numpy(A)
    calls = [0]

    def matvec(v):
        calls[0] += 1
        return At @ v

    w, stats = kiops(taus, matvec, torch.from_numpy(u), tol=1e-10, m_init=8, mmin=8, mmax=48)
    assert w.shape == (len(taus), A.shape[0])
    for k, tau in enumerate(taus):
        ref = _phi_exact(A, u, tau)
        err = np.abs(w[k].numpy() - ref).max() / np.abs(ref).max()
        assert err < 1e-8, (k, tau, err, stats)
    assert stats[2] == calls[0] and stats[0] >= 1 and stats[4] < 1e-8


def test_kiops_single_vector_and_task1():
    """u with one row (p = 0): exp(tau A) u_0; task1 divides each output by its tau (kiops.py:341-344)."""
    from wxfactory_amd.solvers import kiops

    A, u = _problem(p=1)
    w, _ = kiops([0.5, 1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u[:1]), tol=1e-10, m_init=10, mmax=40)
    for k, tau in enumerate([0.5, 1.0]):
        ref = expm(tau * A) @ u[0]
        assert np.abs(w[k].numpy() - ref).max() < 1e-8 * np.abs(ref).max()
    w1, _ = kiops([0.5, 1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u[:1]), tol=1e-10, m_init=10, mmax=40,
                  task1=True)
    assert torch.allclose(w1[0] * 0.5, w[0], rtol=1e-12, atol=0) and torch.allclose(w1[1], w[1], rtol=1e-12, atol=0)


def test_kiops_happy_breakdown():
    """A Krylov space that closes after a few vectors (u in a 4-dimensional invariant subspace): the exact answer,
    with fewer matvecs than the requested basis size."""
    from wxfactory_amd.solvers import kiops

    n = 40
    lam = -np.linspace(0.5, 3.0, n)
    u = np.zeros((2, n))
    u[1, [1, 5, 9, 20]] = [1.0, -2.0, 0.5, 3.0]
    calls = [0]

    def matvec(v):
        calls[0] += 1
        return torch.from_numpy(lam) * v

    w, stats = kiops([1.0], matvec, torch.from_numpy(u), tol=1e-9, m_init=12, mmin=12, mmax=30)
    ref = (np.expm1(lam) / lam) * u[1]
    assert np.abs(w[0].numpy() - ref).max() < 1e-10
    assert stats[0] == 1 and stats[2] <= 6, stats


@pytest.mark.parametrize("ortho", ["igs", "cgs"])
def test_fgmres_happy_breakdown(ortho):
    """A Krylov space that closes early: A = c I (one vector spans it: the lagged norm of the second row is exactly
    zero), and an operator with three distinct eigenvalues (closes after three vectors, up to rounding).  Both
    orthogonalisations return the exact solution with flag 0 instead of dividing by the vanished norm."""
    from wxfactory_amd.solvers import fgmres

    rng = np.random.default_rng(8)
    b = torch.from_numpy(rng.uniform(-1.0, 1.0, 64))
    x, norm_r, norm_b, niter, flag, _ = fgmres(lambda v: 2.0 * v, b, tol=1e-12, restart=10)
    assert flag == 0 and niter <= 2 and torch.allclose(x, b / 2.0, rtol=1e-14, atol=0) and norm_r <= 1e-12 * norm_b
    lam = torch.from_numpy(np.repeat([1.5, -0.7, 4.0], [20, 20, 24]))
    x, norm_r, norm_b, niter, flag, _ = fgmres(lambda v: lam * v, b, tol=1e-12, restart=10)
    assert flag == 0 and niter <= 5, (flag, niter)
    assert torch.allclose(x, b / lam, rtol=1e-11, atol=1e-13)
    # with a preconditioner (a second set of vectors is kept) and a non-zero first guess
    x, norm_r, norm_b, niter, flag, _ = fgmres(lambda v: lam * v, b, x0=0.3 * b, tol=1e-12, restart=10,
                                               preconditioner=lambda v: 0.5 * v, ortho=ortho)
    assert flag == 0 and torch.allclose(x, b / lam, rtol=1e-11, atol=1e-13)
    x, *_, flag, _ = fgmres(lambda v: 2.0 * v, b, tol=1e-12, restart=10, ortho=ortho)
    assert flag == 0 and torch.allclose(x, b / 2.0, rtol=1e-14, atol=0)
    x, *_, flag, _ = fgmres(lambda v: lam * v, b, tol=1e-12, restart=10, ortho=ortho)
    assert flag == 0 and torch.allclose(x, b / lam, rtol=1e-11, atol=1e-13)


def test_fgmres_against_dense_solve():
    from wxfactory_amd.solvers import fgmres

    A, u = _problem(n=80)
    M = np.eye(80) - 0.3 * A
    x, norm_r, norm_b, niter, flag, res = fgmres(lambda v: torch.from_numpy(M) @ v, torch.from_numpy(u[0]), tol=1e-11,
                                                 restart=25, maxiter=20)
    ref = np.linalg.solve(M, u[0])
    assert flag == 0 and np.abs(x.numpy() - ref).max() < 1e-8 * np.abs(ref).max()
    assert float(norm_r) / float(norm_b) < 1e-11 and 0 < niter <= len(res)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _CommOverGloo:
    """The duck type reduce.py takes for the library's communicator (RcclComm: `world`, `allreduce(tensor, op)`), carried by
    gloo here: with it as `group`, EVERY reduction of a solver must arrive at allreduce() - a reduction that went to
    torch.distributed's default group directly would be a call the several-GPU data path does not have."""

    def __init__(self, world):
        self.world, self.calls = world, 0

    def allreduce(self, t, op="sum"):
        self.calls += 1
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
        return t


def _split_worker(rank, world, port, q, through_comm=False):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd import solvers as _solvers
        from wxfactory_amd.solvers import fgmres as _fgmres, kiops as _kiops, pmex as _pmex

        comm = _CommOverGloo(world) if through_comm else None
        if through_comm:
            # no solver code may reach the default group behind the communicator's back: every torch.distributed collective
            # that is not issued by _CommOverGloo.allreduce (or by this test's own gather) raises
            real_all_reduce = dist.all_reduce

            def guarded(t, *a, **k):
                import inspect

                if inspect.stack()[1].function != "allreduce":
                    raise AssertionError("a reduction went to torch.distributed directly, not through the communicator")
                return real_all_reduce(t, *a, **k)

            dist.all_reduce = guarded
        fgmres = lambda *a, **k: _fgmres(*a, group=comm, **k)  # noqa: E731
        kiops = lambda *a, **k: _kiops(*a, group=comm, **k)  # noqa: E731
        pmex = lambda *a, **k: _pmex(*a, group=comm, **k)  # noqa: E731

        A, u = _problem(p=2)
        n = A.shape[0]
        lo, hi = rank * n // world, (rank + 1) * n // world
        Arows = torch.from_numpy(A[lo:hi])

        def gather(v):
            parts = [torch.empty(((r + 1) * n // world - r * n // world,), dtype=v.dtype) for r in range(world)]
            dist.all_gather(parts, v.contiguous())
            return torch.cat(parts)

        w, stats = kiops([0.5, 1.0], lambda v: Arows @ gather(v), torch.from_numpy(u[:, lo:hi].copy()), tol=1e-10,
                         m_init=8, mmin=8, mmax=48)
        for k, tau in enumerate([0.5, 1.0]):
            ref = _phi_exact(A, u, tau)
            err = np.abs(w[k].numpy() - ref[lo:hi]).max() / np.abs(ref).max()
            assert err < 1e-8, (rank, k, err, stats)
        wp, stats_p = pmex([0.5, 1.0], lambda v: Arows @ gather(v), torch.from_numpy(u[:, lo:hi].copy()), tol=1e-10,
                           m_init=8, mmin=8, mmax=48)
        for k, tau in enumerate([0.5, 1.0]):
            ref = _phi_exact(A, u, tau)
            err = np.abs(wp[k].numpy() - ref[lo:hi]).max() / np.abs(ref).max()
            assert err < 1e-8, (rank, k, err, stats_p)
        stats = (stats, stats_p)
        Mrows = torch.from_numpy((np.eye(n) - 0.3 * A)[lo:hi])
        x, norm_r, norm_b, niter, flag, _ = fgmres(lambda v: Mrows @ gather(v), torch.from_numpy(u[0, lo:hi].copy()),
                                                   tol=1e-11, restart=25, maxiter=20)
        ref = np.linalg.solve(np.eye(n) - 0.3 * A, u[0])
        assert flag == 0 and np.abs(x.numpy() - ref[lo:hi]).max() < 1e-8 * np.abs(ref).max()
        if through_comm:
            assert comm.calls > 50, comm.calls   # (two per Krylov vector of kiops, one per vector of pmex / fgmres, ...)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", stats))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise


@pytest.mark.parametrize("through_comm", [False, True])
def test_solvers_with_vectors_split_over_gloo_ranks(through_comm):
    """through_comm: the reductions over a communicator object (the library's RcclComm on GPUs; a gloo-backed stand-in here)
    instead of torch.distributed's default group - the same adaptive decisions on every rank either way."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q, through_comm)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]
    assert res[0][2] == res[1][2]  # every rank took the same adaptive decisions


def test_fgmres_reorthogonalises_after_cancellation(monkeypatch):
    """An operator close to the identity: A v is almost v, the first Gram-Schmidt pass cancels four digits and
    the second pass must run (twice the dot-product sweeps); a well-separated operator takes one pass."""
    from wxfactory_amd import solvers

    calls = [0]
    orig = solvers._Basis.dots

    def counting(self, lo, hi, w, out=None):
        calls[0] += 1
        return orig(self, lo, hi, w, out=out)

    monkeypatch.setattr(solvers._Basis, "dots", counting)
    A, u = _problem(n=80)
    for scale, passes in ((1e-4, 2), (1.0, 1)):
        M = np.eye(80) - scale * A
        calls[0] = 0
        x, norm_r, norm_b, niter, flag, _ = solvers.fgmres(lambda v: torch.from_numpy(M) @ v, torch.from_numpy(u[0]),
                                                          tol=1e-12, restart=30, maxiter=10, ortho="cgs")
        ref = np.linalg.solve(M, u[0])
        assert flag == 0 and np.abs(x.numpy() - ref).max() < 1e-9 * np.abs(ref).max()
        assert passes * niter - 2 <= calls[0] <= passes * niter, (scale, calls[0], niter)


def test_low_sync_gram_schmidt_is_the_default_and_agrees_with_cgs():
    """fgmres' default orthogonalisation is the reference's one-synchronisation iterated Gram-Schmidt
    (solvers/fgmres.py:16-73): one reduction per Krylov vector.  Same solution as the classical variant, same
    iteration count on a well-conditioned problem, through restarts, with and without a preconditioner."""
    from wxfactory_amd import solvers

    A, u = _problem(n=120)
    M = np.eye(120) - 0.4 * A
    Mt = torch.from_numpy(M)
    ref = np.linalg.solve(M, u[0])
    reductions = [0]
    orig = solvers._Basis.dots2

    def counting(self, m, a, b):
        reductions[0] += 1
        return orig(self, m, a, b)

    solvers._Basis.dots2 = counting
    try:
        out = {}
        for ortho in ("igs", "cgs"):
            reductions[0] = 0
            x, norm_r, norm_b, niter, flag, res = solvers.fgmres(lambda v: Mt @ v, torch.from_numpy(u[0]), tol=1e-12,
                                                                restart=12, maxiter=40, ortho=ortho)
            assert flag == 0 and np.abs(x.numpy() - ref).max() < 1e-9 * np.abs(ref).max(), ortho
            out[ortho] = (niter, reductions[0], len(res))
        assert abs(out["igs"][0] - out["cgs"][0]) <= 1, out
        # one fused reduction per Krylov vector plus one per restart cycle (the first vector's), none in "cgs"
        cycles = -(-out["igs"][0] // 12)
        assert out["igs"][1] == out["igs"][0] + cycles and out["cgs"][1] == 0, out
    finally:
        solvers._Basis.dots2 = orig
    # flexible: a (linear) preconditioner, separate Z vectors
    D = torch.from_numpy(1.0 / np.diag(M))
    x, _, _, niter, flag, _ = solvers.fgmres(lambda v: Mt @ v, torch.from_numpy(u[0]), tol=1e-12, restart=12, maxiter=40,
                                            preconditioner=lambda v: D * v)
    assert flag == 0 and np.abs(x.numpy() - ref).max() < 1e-9 * np.abs(ref).max()


def _idle_rank_worker(rank, world, port, q):
    """Vectors split over the first `world - 2` ranks only: the last two own EMPTY slices (ranks 6, 7 of an 8-GPU node
    with one panel per GPU) and must make the same collective calls as the others."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.solvers import fgmres, kiops

        A, u = _problem(n=96, p=1)
        n, active = A.shape[0], world - 2
        bounds = [min(r, active) * n // active for r in range(world + 1)]
        lo, hi = bounds[rank], bounds[rank + 1]

        def gather(v):   # (gloo's all_gather wants equal sizes: pad to the longest slice)
            width = max(bounds[r + 1] - bounds[r] for r in range(world))
            parts = [torch.empty((width,), dtype=v.dtype) for _ in range(world)]
            dist.all_gather(parts, torch.cat((v, v.new_zeros(width - v.numel()))))
            return torch.cat([parts[r][: bounds[r + 1] - bounds[r]] for r in range(world)])

        M = np.eye(n) - 0.3 * A
        Mrows = torch.from_numpy(M[lo:hi])
        ref = np.linalg.solve(M, u[0])
        for ortho in ("igs", "cgs"):
            x, _, _, niter, flag, _ = fgmres(lambda v: Mrows @ gather(v), torch.from_numpy(u[0, lo:hi].copy()), tol=1e-11,
                                             restart=20, maxiter=20, ortho=ortho)
            assert flag == 0 and x.numel() == hi - lo
            if hi > lo:
                assert np.abs(x.numpy() - ref[lo:hi]).max() < 1e-8 * np.abs(ref).max()
        Arows = torch.from_numpy(A[lo:hi])
        w, stats = kiops([1.0], lambda v: Arows @ gather(v), torch.from_numpy(u[:, lo:hi].copy()), tol=1e-10, m_init=8,
                         mmin=8, mmax=48)
        if hi > lo:
            refw = _phi_exact(A, u, 1.0)
            assert np.abs(w[0].numpy() - refw[lo:hi]).max() < 1e-8 * np.abs(refw).max()
        # the basis-size query (_affordable_mmax) is collective: with the "small basis" shortcut at a size that the
        # working ranks exceed and the idle ranks (n = 0) do not, every rank must still reach its all-reduce - decided
        # from rank-local n they would part ways there and pair the MIN with the next SUM (or hang)
        import wxfactory_amd.solvers as solvers_mod

        solvers_mod._BASIS_CHECK_BYTES = 2000
        for solver in (kiops, solvers_mod.pmex):
            w2, stats2 = solver([1.0], lambda v: Arows @ gather(v), torch.from_numpy(u[:, lo:hi].copy()), tol=1e-10, m_init=8,
                                mmin=8, mmax=48)
            if hi > lo:
                assert np.abs(w2[0].numpy() - refw[lo:hi]).max() < 1e-8 * np.abs(refw).max()
        assert solvers_mod._affordable_mmax(hi - lo, 1, 48, 8, "cpu", torch.float64) == 48
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", stats))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise


def test_solvers_with_idle_ranks():
    world = 4   # two working ranks, two with empty slices
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_idle_rank_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]
    assert all(r[2] == res[0][2] for r in res)


DENSE_PROBLEMS = ("phi1", "phi3_outputs", "long_interval", "backwards", "invariant_subspace", "unit_test_identity")


@pytest.mark.parametrize("problem", DENSE_PROBLEMS)
@pytest.mark.parametrize("solver", ("kiops", "pmex"))
def test_exponential_solvers_against_reference_runs(problem, solver):
    """kiops and pmex on seeded dense operators against the arrays and `stats` the reference's own solvers/kiops.py and
    solvers/pmex.py returned for them (tests/golden/solvers_dense.npz, made by oracle/refharness/gen_golden.py
    solvers_dense): every decision of the two adaptive controllers - sub-steps, rejections, Krylov vectors, exponentials,
    final basis size, and for pmex the number of norms that needed a reduction of their own - and the phi-vectors."""
    import json

    from wxfactory_amd import solvers

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "solvers_dense.npz"))
    A = torch.from_numpy(g[f"{problem}/A"])
    u = torch.from_numpy(g[f"{problem}/u"])
    args = json.loads(str(g[f"{problem}/{solver}_args"]))
    w, stats = getattr(solvers, solver)(g[f"{problem}/tau"].tolist(), lambda v: A @ v, u, **args)
    ref_w, ref_stats = g[f"{problem}/{solver}_w"], g[f"{problem}/{solver}_stats"]
    exact = (0, 1, 2, 3, 5) + ((6,) if solver == "pmex" else ())
    if problem == "invariant_subspace" and solver == "pmex":
        # At the breakdown pmex estimates a norm of 1e-14 as the square root of a difference of O(1) numbers: whether
        # that difference comes out at -1e-16 (own reduction, breakdown seen at once: 5 vectors, here) or +1e-16 (estimate
        # 1e-8 > tol, a vector of noise is normalised and the breakdown is seen one vector later: 6, the reference's run)
        # is rounding.  The result is the same (given that the last Hessenberg column is stored before the test, see
        # solvers.pmex); the vector count is not compared.
        exact = (0, 1, 3, 5)
    assert [int(stats[i]) for i in exact] == [int(ref_stats[i]) for i in exact], (stats, ref_stats.tolist())
    assert abs(float(stats[4]) - float(ref_stats[4])) <= 1e-3 * float(ref_stats[4]) + 1e-300
    err = np.abs(w.numpy() - ref_w).max(axis=1) / np.abs(ref_w).max(axis=1)
    assert (err < 1e-11).all(), err


@pytest.mark.parametrize("p", [1, 3])
@pytest.mark.parametrize("taus", [[1.0], [0.25, 0.6, 1.0]])
def test_pmex_against_dense_phi(p, taus):
    """pmex against the phi functions from one exponential of the augmented matrix (as for kiops above)."""
    from wxfactory_amd.solvers import pmex

    A, u = _problem(p=p)
    w, stats = pmex(taus, lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), tol=1e-10, m_init=8, mmin=8, mmax=48)
    assert len(stats) == 7 and stats[0] >= 1 and stats[2] >= 8
    for k, tau in enumerate(taus):
        ref = _phi_exact(A, u, tau)
        assert np.abs(w[k].numpy() - ref).max() < 1e-8 * np.abs(ref).max(), (k, stats)


def test_pmex_single_row_and_nan():
    """One row in `u` (p = 0): exp(tau A) u_0 - the reference's own p = 0 branch cannot run (it stacks a row of the wrong
    length, pmex.py:53-56), the meaning is that of kiops; a NaN from the operator raises instead of looping for ever."""
    from wxfactory_amd.solvers import pmex

    A, u = _problem(n=60, p=1)
    w, _ = pmex([0.5, 1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u[:1]), tol=1e-10, m_init=10, mmax=40)
    for k, tau in enumerate([0.5, 1.0]):
        ref = expm(tau * A) @ u[0]
        assert np.abs(w[k].numpy() - ref).max() < 1e-8 * np.abs(ref).max()
    with pytest.raises(ValueError, match="NaN"):
        pmex([1.0], lambda v: torch.from_numpy(A) @ v * float("nan"), torch.from_numpy(u), tol=1e-8)


@pytest.mark.parametrize("solver", ("kiops", "pmex"))
def test_restart_powers(solver):
    """A stiff operator that needs several sub-steps, three phi functions: with the reference's restart exponents
    (solvers/kiops.py:139-141, pmex.py:137-139: i = p - k + 1, two more than phipm's) the result is not the phi-sum - that
    IS what the reference returns, and what the default reproduces (pinned by tests/golden/solvers_dense.npz
    `long_interval` and the EPI fixtures) - while restart_powers="phipm" is exact.  One phi function (EPI2) is exact
    either way."""
    from wxfactory_amd import solvers

    rng = np.random.default_rng(1)
    n = 120
    A = 8 * (-np.diag(rng.uniform(0.05, 8, n)) + 0.4 * rng.standard_normal((n, n)) / np.sqrt(n))
    taus = [0.25, 0.6, 1.0]
    fn = getattr(solvers, solver)
    for p, powers, exact in ((1, "reference", True), (2, "reference", False), (2, "phipm", True), (3, "phipm", True)):
        u = np.random.default_rng(p).standard_normal((p + 1, n))
        w, stats = fn(taus, lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), tol=1e-9, m_init=6, mmin=6, mmax=20,
                      restart_powers=powers)
        assert stats[0] > 1   # several sub-steps
        err = max(np.abs(w[k].numpy() - _phi_exact(A, u, t)).max() / np.abs(_phi_exact(A, u, t)).max()
                  for k, t in enumerate(taus))
        assert (err < 1e-9) if exact else (err > 1e-3), (p, powers, err)
    with pytest.raises(ValueError, match="restart_powers"):
        fn(taus, lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), restart_powers="other")


def test_pmex_breakdown_option():
    """A happy breakdown that is seen at once (u in a four-dimensional invariant subspace of a block-diagonal operator).
    The reference stores the Hessenberg column of the vector that breaks down only after its test (pmex.py:225-233), so its
    result misses the projections of the last product: breakdown="reference" (the default: drop-in behaviour) is off by
    their weight, breakdown="exact" keeps the column and is exact.  Same statistics either way."""
    from wxfactory_amd.solvers import pmex

    rng = np.random.default_rng(3)
    n = 40
    A = np.zeros((n, n))
    A[:4, :4] = 0.8 * rng.standard_normal((4, 4))
    A[4:, 4:] = np.diag(rng.uniform(-2, -0.1, n - 4))
    u = np.zeros((2, n))
    u[0, :4], u[1, :4] = rng.standard_normal(4), rng.standard_normal(4)
    ref = _phi_exact(A, u, 1.0)
    out = {}
    for mode in ("reference", "exact"):
        w, st = pmex([1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), tol=1e-10, m_init=10, mmin=10, mmax=30,
                     breakdown=mode)
        out[mode] = (np.abs(w[0].numpy() - ref).max() / np.abs(ref).max(), st)
    assert out["exact"][0] < 1e-13 and 1e-4 < out["reference"][0] < 1e-1, out
    assert out["exact"][1] == out["reference"][1] and out["exact"][1][2] == 4       # four vectors, then the breakdown
    with pytest.raises(ValueError, match="breakdown"):
        pmex([1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), breakdown="other")


@pytest.mark.parametrize("seed", range(10))
def test_low_sync_gram_schmidt_breakdown_is_resolved(seed):
    """The lagged norm of the one-synchronisation Gram-Schmidt is the root of <a, a> - s.s, a difference of two numbers of
    size |a|^2: at a breakdown it is rounding of either sign and says nothing below sqrt(eps) |a| (taken at face value, half
    of these seeds return a "norm" of 1e-8 and the row of noise is normalised).  A row that lies entirely in the span of
    its finished predecessors - an invariant subspace of dimension three - must come back as vanished: 0.0."""
    import wxfactory_amd.solvers as S

    rng = np.random.default_rng(seed)
    n, j = 200, 5
    V = torch.zeros((8, n), dtype=torch.float64)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, 6)))
    for k in range(j - 2):
        V[k] = torch.from_numpy(Qm[:, k])                                   # finished rows 0 .. j-3
    V[j - 2] = torch.from_numpy(Qm[:, : j - 2] @ rng.standard_normal(j - 2))   # row j-2: inside their span
    V[j - 1] = torch.from_numpy(rng.standard_normal(n))
    gs = S._LowSyncGramSchmidt(S._Basis(V), 8)
    assert gs.step(j) == 0.0
    assert np.isfinite(gs.R).all() and gs.R[j - 2, j - 2] == 0.0
    # ... while a row with a genuine orthogonal part of 1e-7 of its length keeps it (to the accuracy the explicit norm has)
    V[j - 2] = torch.from_numpy(Qm[:, : j - 2] @ rng.standard_normal(j - 2) + 1e-7 * Qm[:, j - 2])
    gs = S._LowSyncGramSchmidt(S._Basis(V.clone()), 8)
    assert abs(gs.step(j) - 1e-7) < 1e-12


@pytest.mark.parametrize("solver", ["kiops", "pmex"])
def test_padded_basis_rows_change_nothing(solver, monkeypatch):
    """Long Krylov bases are handed out with their rows on 256-byte boundaries (solvers._basis_rows: a view of a padded
    allocation, the row stride is not the row length).  With the threshold lowered so that this small problem counts as long:
    the same phi-vectors and statistics, bit for bit, as on dense rows."""
    from wxfactory_amd import solvers

    rows = solvers._basis_rows(5, 300_001, torch.float64, "cpu")
    assert rows.shape == (5, 300_001) and rows.stride() == (300_032, 1) and not rows.is_contiguous()
    assert all(rows[r].data_ptr() % 256 == rows[0].data_ptr() % 256 for r in range(5))
    dense = solvers._basis_rows(5, 1000, torch.float64, "cpu")
    assert dense.is_contiguous()

    A, u = _problem(p=1)
    fn = getattr(solvers, solver)
    args = dict(tol=1e-10, m_init=8, mmin=8, mmax=48)
    w0, st0 = fn([0.4, 1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), **args)
    monkeypatch.setattr(solvers.KiopsWorkspace, "max_fused_len", 16)   # (n + p = 61 is "long" now: rows of 64)
    assert not solvers._basis_rows(3, A.shape[0] + 1, torch.float64, "cpu").is_contiguous()
    w1, st1 = fn([0.4, 1.0], lambda v: torch.from_numpy(A) @ v, torch.from_numpy(u), **args)
    assert torch.equal(w0, w1) and tuple(st0) == tuple(st1)


def _interpreted_columns(R, j0, j1, restart, vn, cs, sn, g, Hm, tol_abs, rate):
    """fgmres.py:75-94, 202-246 as the interpreter runs it (the checker of wx_fgmres_rotate_columns: Python floats, ** 2)."""
    import math

    def rotg(a, b):
        if b == 0.0:
            return 1.0, 0.0
        if a == 0.0:
            return 0.0, 1.0
        scl = min(abs(a), abs(b))
        sigma = math.copysign(1.0, a) if abs(a) > abs(b) else math.copysign(1.0, b)
        r = sigma * (scl * math.sqrt((a / scl) ** 2 + (b / scl) ** 2))
        return a / r, b / r

    res, stopped = [], False
    for j in range(j0, j1):
        hj = R[: j + 2, j + 1].tolist()
        for i in range(j):
            t = cs[i] * hj[i] + sn[i] * hj[i + 1]
            hj[i + 1] = -sn[i] * hj[i] + cs[i] * hj[i + 1]
            hj[i] = t
        if hj[j + 1] != 0.0:
            c, s = rotg(hj[j], hj[j + 1])
            hj[j], hj[j + 1] = c * hj[j] + s * hj[j + 1], 0.0
            g[j], g[j + 1] = c * g[j] + s * g[j + 1], -s * g[j] + c * g[j + 1]
        else:
            c, s = 1.0, 0.0
        cs[j], sn[j] = c, s
        Hm[j][: j + 2] = hj
        if g[j] != 0.0 and abs(g[j + 1]) > 0.0:
            q = abs(g[j + 1]) / abs(g[j])
            rate = q if j == 0 or rate is None else 0.5 * (rate + q)
        res.append(abs(g[j + 1]))
        if j < restart - 1 or vn[j + 1] == 0.0:
            if res[-1] < tol_abs or res[-1] != res[-1] or vn[j + 1] == 0.0:
                stopped = True
                break
    return len(res), stopped, rate, res


@pytest.mark.parametrize("seed", range(6))
def test_host_side_of_fgmres_in_c_is_the_interpreted_loop(built_lib, seed):
    """wx_fgmres_rotate_columns / wx_fgmres_back_substitute (plain host C behind the C ABI) against the reference's interpreted
    loops: the same bits in the rotations, g, the rotated columns, the estimates, the decay rate, the stopping column and y -
    whole cycles at once and in ragged chunks (what device passes hand over), zeros in the subdiagonal, a vanished row, NaN."""
    from wxfactory_amd.solvers import _host_lib, _rotate_columns

    rng = np.random.default_rng(seed)
    restart = 30
    ld = restart + 2
    R = np.triu(rng.standard_normal((ld, ld)), -1) * np.exp(rng.uniform(-8, 2, (ld, ld)))
    R[np.arange(1, ld), np.arange(0, ld - 1)] = 0.0        # column j of the Hessenberg matrix is R[: j + 2, j + 1]
    vn = np.abs(rng.standard_normal(ld)) + 0.1
    if seed == 1:
        R[7, 7] = 0.0                                      # h_{j+1,j} = 0: the identity rotation
    if seed == 2:
        vn[12] = 0.0                                       # a vanished row: breakdown, the estimate is exact
    if seed == 3:
        R[5, 9] = float("nan")
    tol_abs = 1e-9 if seed != 4 else 1e-2                  # (seed 4: converges inside the cycle)
    g0 = float(np.abs(rng.standard_normal()) + 1.0)
    # the interpreted loop, one cycle
    cs_p, sn_p, g_p = [0.0] * (restart + 1), [0.0] * (restart + 1), [0.0] * ld
    g_p[0] = g0
    Hm_p = [[0.0] * ld for _ in range(restart)]
    n_p, stop_p, rate_p, res_p = _interpreted_columns(R, 0, restart, restart, vn.tolist(), cs_p, sn_p, g_p, Hm_p, tol_abs, None)
    # the C helper, in ragged chunks
    cs, sn, g, Hm = np.zeros(restart + 1), np.zeros(restart + 1), np.zeros(ld), np.zeros((restart, ld))
    g[0] = g0
    rate, stopped, res = np.full(1, np.nan), np.zeros(1, dtype=np.int32), np.zeros(restart + 1)
    j, got_res = 0, []
    while j < restart and not stopped[0]:
        j1 = min(restart, j + int(rng.integers(1, 9)))
        took = _rotate_columns(R, j, j1, restart, vn, cs, sn, g, Hm, tol_abs, rate, res, stopped)
        got_res += res[:took].tolist()
        j += took
    assert (j, bool(stopped[0])) == (n_p, stop_p)
    same = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert same(cs[:j], cs_p[:j]) and same(sn[:j], sn_p[:j]) and same(g, g_p) and same(got_res, res_p)
    assert same(Hm[:j], np.asarray(Hm_p)[:j])
    assert (rate_p is None and np.isnan(rate[0])) or same(rate[0], rate_p)
    # the back substitution
    y_p = [0.0] * j
    for i in range(j - 1, -1, -1):
        acc = g_p[i]
        for l in range(i + 1, j):
            acc -= Hm_p[l][i] * y_p[l]
        y_p[i] = acc / Hm_p[i][i]
    y = np.zeros(j)
    assert _host_lib().wx_fgmres_back_substitute(Hm.ctypes.data, ld, j, g.ctypes.data, y.ctypes.data) == 0
    assert same(y, y_p)
