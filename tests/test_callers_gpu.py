"""Callers of the path on the GPU: JVPs, the Rosenbrock operator, one SSP-RK3 step and
one Ros2+FGMRES step, against values produced by the reference's own matvec.py / tvdrk3.py / ros2.py."""
import numpy as np
import pytest
import torch

from tests.util import Golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup(built_lib):
    from tests.gpu_util import device_metric
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("callers_euler3d_n3_h3_v2")
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
    rhs = RhsEuler3D(plans)
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    return g, rhs, stack


def _rel(a, b):
    a = a.cpu().numpy().reshape(b.shape)
    ax = (0, 2, 3, 4, 5)
    return np.abs(a - b).max(axis=ax) / np.abs(b).max(axis=ax)


def test_stacked_state_rhs(setup):
    g, rhs, stack = setup
    R = rhs(stack("Q"))
    assert (_rel(R, stack("R").cpu().numpy()) < 1e-11).all()


def test_matvec_fun_and_rat(setup):
    from wxfactory_amd.matvec import matvec_fun, matvec_rat

    g, rhs, stack = setup
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    jc = matvec_fun(V.flatten(), dt, Q, R, rhs, "complex")
    assert (_rel(jc, stack("jvp_complex").cpu().numpy()) < 1e-9).all()
    # finite differences amplify rounding by 1/eps_fd = 2900: same formula, looser bound
    jf = matvec_fun(V.flatten(), dt, Q, R, rhs, "fd")
    assert (_rel(jf, stack("jvp_fd").cpu().numpy()) < 1e-6).all()
    ra = matvec_rat(V.flatten(), dt, Q, R, rhs)
    assert (_rel(ra, stack("rat").cpu().numpy()) < 1e-6).all()
    # the two JVP flavours agree with each other to FD truncation error
    assert (_rel(jf, jc.cpu().numpy().reshape(6, *Q.shape[1:])) < 1e-3).all()


@pytest.mark.parametrize("fused,pipeline", [(False, False), (True, False), (True, True)])
def test_tvdrk3_step(setup, fused, pipeline):
    from wxfactory_amd.integrators import Tvdrk3

    g, rhs, stack = setup
    stepper = Tvdrk3(rhs, fused=fused, pipeline=pipeline)
    assert stepper.fused == fused and stepper.pipeline == pipeline
    rhs.batched = not pipeline  # (small tiles: the batched stage update would pre-empt the pipeline)
    try:
        Qn = stepper.step(stack("Q"), float(g["meta/dt_rk"]))
    finally:
        rhs.batched = True
    ref = stack("rk3").cpu().numpy()
    dq = np.abs(ref - stack("Q").cpu().numpy()).max(axis=(0, 2, 3, 4, 5))
    err = np.abs(Qn.cpu().numpy() - ref).max(axis=(0, 2, 3, 4, 5))
    assert (err <= 1e-9 * dq + 1e-14 * np.abs(ref).max(axis=(0, 2, 3, 4, 5))).all(), (err, dq)


def test_stage_pipeline_equals_separate_extrapolation(setup):
    """Three SSP-RK3 steps with the stage pipeline (each stage's kernel also writes the faces of its
    output; two interface/edge buffer sets alternate) against the same steps with a separate
    extrapolation pass per stage.  A state modified in place between steps must not reuse stale faces."""
    from wxfactory_amd.integrators import Tvdrk3

    g, rhs, stack = setup
    dt = float(g["meta/dt_rk"])
    plain, piped = Tvdrk3(rhs, pipeline=False), Tvdrk3(rhs, pipeline=True)
    rhs.batched = False  # (tiles this small would otherwise take the batched stage update, not the pipeline)
    try:
        _pipeline_against_plain(plain, piped, stack, dt)
    finally:
        rhs.batched = True


def _pipeline_against_plain(plain, piped, stack, dt):
    Qa = Qb = stack("Q")
    for step in range(3):
        Qa, Qb = plain.step(Qa, dt), piped.step(Qb, dt)
        if step == 1:  # in-place edit: the pipeline has to notice and re-extrapolate
            Qa.mul_(1.0 + 1e-3)
            Qb.mul_(1.0 + 1e-3)
    scale = Qa.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    assert ((Qa - Qb).abs() <= 1e-14 * scale).all(), ((Qa - Qb).abs() / scale).amax(dim=(0, 2, 3, 4, 5))
    # a recycled-looking tensor (same values, other storage) is not mistaken for the pipeline's output
    Qc = piped.step(Qb.clone(), dt)
    Qd = plain.step(Qa, dt)
    assert ((Qc - Qd).abs() <= 1e-13 * scale).all()


def test_pipeline_contract_invalidate_and_debug_check(setup):
    """A write the version counter cannot see (here through `.data`, as a raw-pointer kernel would) must be announced
    with invalidate_faces(); check_faces turns a forgotten announcement into an error instead of stale faces."""
    from wxfactory_amd.integrators import Tvdrk3

    g, rhs, stack = setup
    dt = float(g["meta/dt_rk"])
    plain, piped = Tvdrk3(rhs, pipeline=False), Tvdrk3(rhs, pipeline=True)
    rhs.batched = False
    try:
        Qa, Qb = plain.step(stack("Q"), dt), piped.step(stack("Q"), dt)
        v0 = Qb._version
        Qa.data.mul_(1.0 + 1e-3)
        Qb.data.mul_(1.0 + 1e-3)
        assert Qb._version == v0   # invisible to torch
        rhs.check_faces = True
        with pytest.raises(RuntimeError, match="invalidate_faces"):
            piped.step(Qb, dt)
        rhs.check_faces = False
        rhs.invalidate_faces()
        Qa2, Qb2 = plain.step(Qa, dt), piped.step(Qb, dt)
        scale = Qa2.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
        assert ((Qa2 - Qb2).abs() <= 1e-14 * scale).all()
        # an honest chain passes the check
        rhs.check_faces = True
        Qb3 = piped.step(Qb2, dt)
        assert ((plain.step(Qa2, dt) - Qb3).abs() <= 1e-14 * scale).all()
    finally:
        rhs.check_faces = False
        rhs.batched = True


@pytest.mark.parametrize("ortho", ["igs", "cgs"])
def test_ros2_fgmres_step(setup, ortho):
    """One Ros2 step (integrators/ros2.py:24-81) through fgmres with the reference's one-synchronisation
    Gram-Schmidt (the default) and with the classical variant: the reference needed 151 iterations for 1e-9."""
    from wxfactory_amd.integrators import Ros2

    g, rhs, stack = setup
    ros = Ros2(rhs, tol=1e-9, gmres_restart=30, ortho=ortho)
    Qn = ros.step(stack("Q"), float(g["meta/dt_jvp"]))
    info = ros.solver_info
    assert info["flag"] == 0 and info["rel_residual"] < 1e-9
    assert abs(info["iterations"] - 151) <= 2, info["iterations"]
    ref, q0 = stack("ros2").cpu().numpy(), stack("Q").cpu().numpy()
    ax = (0, 2, 3, 4, 5)
    upd = np.abs(ref - q0).max(axis=ax)
    err = np.abs(Qn.cpu().numpy() - ref).max(axis=ax)
    assert (err <= 1e-7 * upd).all(), (err / upd)


def test_fgmres_device_passes_take_the_host_loop_decisions(setup, monkeypatch):
    """fgmres with the Gram-Schmidt step on the device and one read-back per pass of several Krylov vectors (wx_fgmres_vector,
    wx_euler3d_batch_fgmres_vector: the operator's scale read from device memory) against the one-vector-at-a-time loop on the
    same step: the same iteration count, the same solution, the reference's 151 iterations; vectors built past the end of a
    cycle are counted."""
    from wxfactory_amd.integrators import Ros2

    g, rhs, stack = setup
    dt = float(g["meta/dt_jvp"])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WXHIP_FGMRES_VECTOR", mode)
        ros = Ros2(rhs, tol=1e-9, gmres_restart=30)
        res[mode] = (ros.step(stack("Q"), dt), ros.solver_info)
    dev, host = res["1"][1], res["0"][1]
    assert dev["device_passes"] > 0 and dev["vectors_built"] >= dev["iterations"] - 12 and host["device_passes"] == 0, (dev, host)
    assert dev["flag"] == 0 and dev["iterations"] == host["iterations"] and abs(dev["iterations"] - 151) <= 2, (dev["iterations"], host["iterations"])
    assert dev["wasted_vectors"] <= 20 * (1 + dev["iterations"] // 30), dev
    ax = (0, 2, 3, 4, 5)
    upd = (res["0"][0] - stack("Q")).abs().amax(dim=ax)
    assert (((res["1"][0] - res["0"][0]).abs().amax(dim=ax)) <= 1e-8 * upd).all()


@pytest.mark.parametrize("n", [3000, 29160, 103680, 131072, 200000, 262144])
def test_fgmres_step_in_one_launch_is_the_three_launch_step(built_lib, monkeypatch, n):
    """wx_fgmres_vector at launch-bound lengths: products, the step's algebra and the update of the two rows from ONE launch (the
    workgroups meet at a barrier inside it) against the three launches it replaces - the same bits in every row, in R, T, K, the
    norms and the coefficients, over a run of steps on one workspace (the barrier's sequence numbers), flag clear."""
    from wxfactory_amd import _lib

    lib = _lib.load()
    rows, steps = 14, 10
    gen = torch.Generator(device=DEV).manual_seed(7 + n)
    V0 = torch.randn((rows, n), generator=gen, device=DEV, dtype=torch.float64)
    q, _ = torch.linalg.qr(V0[:2].T)
    V0[:2] = q.T.contiguous()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WXHIP_FGMRES_ONE_LAUNCH", mode)
        V = V0.clone()
        st = torch.zeros(3 * rows * rows + rows + 1, dtype=torch.float64, device=DEV)
        R, T, K = (st[i * rows * rows: (i + 1) * rows * rows] for i in range(3))
        vn = st[3 * rows * rows: 3 * rows * rows + rows]
        flag = st[3 * rows * rows + rows:].view(torch.int32)[:1]
        coef = torch.zeros(3 * rows, dtype=torch.float64, device=DEV)
        work = torch.full((int(lib.wx_fgmres_workspace(rows)),), float("nan"), dtype=torch.float64, device=DEV)   # (nothing to zero)
        stream = torch.cuda.current_stream().cuda_stream
        for J in range(3, 3 + steps):
            _lib.check(lib.wx_fgmres_vector(V.data_ptr(), V.stride(0), J, n, R.data_ptr(), T.data_ptr(), K.data_ptr(), rows,
                                            coef.data_ptr(), vn.data_ptr(), flag.data_ptr(), work.data_ptr(), None, stream),
                       "wx_fgmres_vector")
        torch.cuda.synchronize()
        out[mode] = (V, st.clone(), coef)
    assert int(out["1"][1][3 * rows * rows + rows:].view(torch.int32)[0]) == 0
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)
    # and the steps did something: the rows the steps finished are orthonormal
    G = out["1"][0][: steps + 1] @ out["1"][0][: steps + 1].T
    assert float((G - torch.eye(steps + 1, device=DEV, dtype=torch.float64)).abs().max()) < 1e-10


def test_kiops_and_epi2_step(setup):
    """phi_1(dt J) R through KIOPS with the complex-step JVP, and the EPI2 step built on it
    (the integrator config/dcmip31.ini actually ships with), against the reference's kiops.py/epi.py."""
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.solvers import kiops

    g, rhs, stack = setup
    Q, R = stack("Q"), stack("R")
    dt = float(g["meta/dt_jvp"])
    vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
    vec[1] = R.flatten()
    phiv, stats = kiops([1], lambda v: matvec_fun(v, dt, Q, R, rhs, "complex"), vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
    ref_stats = g["p0/kiops_stats"]
    # the adaptive controller takes the reference's decisions: steps, rejections, Krylov vectors, exponentials, last m
    assert [int(stats[i]) for i in (0, 1, 2, 3, 5)] == [int(ref_stats[i]) for i in (0, 1, 2, 3, 5)], (stats, ref_stats)
    assert abs(float(stats[4]) - float(ref_stats[4])) <= 1e-3 * float(ref_stats[4])
    ref = stack("kiops_phiv").cpu().numpy()
    ax = (0, 2, 3, 4, 5)
    err = np.abs(phiv.cpu().numpy().reshape(ref.shape) - ref).max(axis=ax) / np.abs(ref).max(axis=ax)
    assert (err < 1e-8).all(), err
    Qn = Epi(2, rhs, tol=1e-7).step(Q, dt)
    refq, q0 = stack("epi2").cpu().numpy(), Q.cpu().numpy()
    upd = np.abs(refq - q0).max(axis=ax)
    assert (np.abs(Qn.cpu().numpy() - refq).max(axis=ax) <= 1e-7 * upd).all()


def test_complex_step_jvp_with_dual_arithmetic(setup):
    """matvec_fun's complex step served by the dual-number kernels (complex_arith='dual'): the same JVP
    as the reference's complex arithmetic, about 20 % cheaper on E7."""
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g, rhs, stack = setup
    rhs_dual = RhsEuler3D(rhs.plans, complex_arith="dual")
    Q, V, R = stack("Q"), stack("V"), stack("R")
    jd = matvec_fun(V.flatten(), float(g["meta/dt_jvp"]), Q, R, rhs_dual, "complex")
    assert next(iter(rhs_dual.plans_for(torch.complex128).values())).dual
    assert (_rel(jd, stack("jvp_complex").cpu().numpy()) < 1e-9).all()


def test_fused_jvp_equals_unfused_complex_step(setup):
    """wx_euler3d_jvp (dual state formed on load, real tangent stored) == the literal complex-step recipe."""
    from wxfactory_amd.matvec import matvec_fun

    g, rhs, stack = setup
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    a = matvec_fun(V.flatten(), dt, Q, R, rhs, "complex")           # fused path (supports_jvp)
    rhs.fused_jvp = False
    try:
        b = matvec_fun(V.flatten(), dt, Q, R, rhs, "complex")       # torch.complex + complex128 kernels + .imag
    finally:
        rhs.fused_jvp = True
    ref = stack("jvp_complex").cpu().numpy()
    assert (_rel(a, ref) < 1e-9).all() and (_rel(b, ref) < 1e-9).all()
    assert (_rel(a, b.cpu().numpy().reshape(ref.shape)) < 1e-12).all()


def test_shift_on_load_equals_materialised_shift(setup):
    """matvec_fun("fd") / matvec_rat with Q + eps v formed inside the kernels (wx_euler3d_shifted_*) against the
    same operators with Q + eps v formed by a torch pass first: the same arithmetic, bit for bit."""
    from wxfactory_amd.matvec import matvec_fun, matvec_rat

    g, rhs, stack = setup
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    assert rhs.supports_shift
    a, b = matvec_fun(V.flatten(), dt, Q, R, rhs, "fd"), matvec_rat(V.flatten(), dt, Q, R, rhs)
    rhs.fused_shift = False
    try:
        a2, b2 = matvec_fun(V.flatten(), dt, Q, R, rhs, "fd"), matvec_rat(V.flatten(), dt, Q, R, rhs)
    finally:
        rhs.fused_shift = True
    assert torch.equal(a, a2) and torch.equal(b, b2)


def test_generic_dual_jvp_kernel_in_a_subprocess():
    """wx_euler3d_jvp runs the JVP-specialised kernel (tangent-only LDS staging) by default; WXHIP_JVP_LEAN=0 (read
    once per process) selects the generic dual-number instantiation, which must pass the same JVP parity tests."""
    import os
    import subprocess
    import sys

    if os.environ.get("WXHIP_JVP_LEAN") == "0":
        pytest.skip("already inside the generic-kernel run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WXHIP_JVP_LEAN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_callers_gpu.py", "-k",
                        "matvec_fun_and_rat or fused_jvp or kiops"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.parametrize("m,n", [(1, 1000), (7, 4097), (8, 65536), (21, 100003)])
def test_krylov_vector_kernels(built_lib, m, n):
    """wx_multi_dot / wx_multi_axpy (the Gram-Schmidt sweeps of fgmres) against torch, on a dense basis and on a padded one
    (a view with a longer row stride - what solvers._basis_rows hands out for long vectors): the same bits."""
    from wxfactory_amd.solvers import _Basis

    gen = torch.Generator(device=DEV).manual_seed(m * 1000 + n)
    Vfull = torch.randn((m + 2, n + 5), generator=gen, device=DEV, dtype=torch.float64)
    V = Vfull[:, :n]  # row stride n + 5: the kernels take it as ldv
    Vc = V.contiguous()
    w = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64)
    basis = _Basis(Vc)
    assert basis.gpu
    h = basis.dots(1, 1 + m, w)
    ref = Vc[1:1 + m] @ w
    assert torch.allclose(h, ref, rtol=1e-12, atol=1e-12 * float(ref.abs().max()))
    assert torch.equal(h, basis.dots(1, 1 + m, w))  # deterministic reduction
    w2 = w.clone()
    out = basis.subtract(w2, 1, 1 + m, h)
    assert out is w2
    ref2 = w - h @ Vc[1:1 + m]
    assert torch.allclose(w2, ref2, rtol=1e-12, atol=1e-12 * float(ref2.abs().max()))
    padded = _Basis(V)
    assert padded.gpu and not V.is_contiguous()
    assert torch.equal(padded.dots(1, 1 + m, w), h)
    w3 = w.clone()
    padded.subtract(w3, 1, 1 + m, h)
    assert torch.equal(w3, w2)
    assert not _Basis(Vfull.t()[:n]).gpu   # rows that are not contiguous: the torch expressions


@pytest.mark.parametrize("m,n", [(0, 513), (1, 1000), (5, 4097), (16, 65536), (17, 30011), (35, 100003)])
def test_two_vector_krylov_kernels(built_lib, m, n):
    """wx_multi_dot2 / wx_pair_update (the one-synchronisation Gram-Schmidt step of fgmres, solvers/fgmres.py:16-73) against
    torch, for row counts below, at and beyond one pass (16 rows), and the streaming combination sum_i y_i V[i]."""
    from wxfactory_amd.solvers import _Basis

    gen = torch.Generator(device=DEV).manual_seed(m * 7919 + n)
    V = torch.randn((m + 3, n), generator=gen, device=DEV, dtype=torch.float64)
    basis = _Basis(V)
    assert basis.gpu
    a, b = V[m], V[m + 1]
    a0, b0 = a.clone(), b.clone()
    if m:
        g = basis.dots2(m, a, b)
        ref = torch.cat((V[:m] @ a0, V[:m] @ b0))
        assert torch.allclose(g, ref, rtol=1e-12, atol=1e-12 * float(ref.abs().max()))
        assert torch.equal(g, basis.dots2(m, a, b))   # deterministic reduction
    ha = torch.randn(m, generator=gen, device=DEV, dtype=torch.float64).cpu().numpy()
    hb = torch.randn(m, generator=gen, device=DEV, dtype=torch.float64).cpu().numpy()
    basis.pair_update(a, b, m, ha, hb, 0.7, -0.3, 1.9)
    ra = a0 - (torch.from_numpy(ha).to(DEV) @ V[:m] if m else 0.0)
    rb = b0 - (torch.from_numpy(hb).to(DEV) @ V[:m] if m else 0.0)
    ra = ra * 0.7
    rb = (rb - (-0.3) * ra) * 1.9
    scale = float(max(ra.abs().max(), rb.abs().max()))
    assert float((a - ra).abs().max()) <= 1e-13 * scale and float((b - rb).abs().max()) <= 1e-13 * scale
    if m:
        y = torch.randn(m, generator=gen, device=DEV, dtype=torch.float64).cpu().tolist()
        comb = basis.combine(m, y)
        refc = torch.tensor(y, dtype=torch.float64, device=DEV) @ V[:m]
        assert float((comb - refc).abs().max()) <= 1e-13 * float(refc.abs().max())


def test_prepared_jvp_equals_unprepared(setup):
    """wx_euler3d_jvp_prepare: the face values of the linearisation state cached once, only tangents extrapolated and
    exchanged per product - the same Jacobian-vector product as the unprepared path (bit for bit), against the
    reference's complex-step value; a modified or different Q falls back to the unprepared path."""
    from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun

    g, rhs, stack = setup
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    rhs.batched = False   # (tiles this small would take the batched launches; the prepared path is the large-tile one)
    try:
        rhs.jvp_release()
        plain = matvec_fun(V.flatten(), dt, Q, R, rhs, "complex")
        op = ComplexStepOperator(dt, Q, R, rhs)
        assert rhs._jvp_is_prepared(Q)
        for scale in (1.0, -0.37):
            a = op((scale * V).flatten())
            rhs.jvp_release()
            b = matvec_fun((scale * V).flatten(), dt, Q, R, rhs, "complex")
            rhs.jvp_prepare(Q)
            assert torch.equal(a, b)
        assert (_rel(op(V.flatten()), stack("jvp_complex").cpu().numpy()) < 1e-9).all()
        # another state, or the same one modified in place: not the prepared one
        Q2 = Q * (1.0 + 1e-3)
        assert not rhs._jvp_is_prepared(Q2)
        c = matvec_fun(V.flatten(), dt, Q2, R, rhs, "complex")
        rhs.jvp_release()
        assert torch.equal(c, matvec_fun(V.flatten(), dt, Q2, R, rhs, "complex"))
        rhs.jvp_prepare(Q)
        Q.mul_(1.0)   # version bump
        assert not rhs._jvp_is_prepared(Q)
        assert torch.equal(matvec_fun(V.flatten(), dt, Q, R, rhs, "complex"), plain)
    finally:
        rhs.jvp_release()
        rhs.batched = True


def test_kiops_vector_kernels_and_graph_replayed_passes(setup):
    """Launch-bound sizes (the shipped .ini files): a Krylov vector of KIOPS is built by wx_kiops_finish (three short
    launches instead of the array-expression recurrence) or, with the complex-step operator, from ONE host call
    (wx_euler3d_batch_kiops_vector); whole passes can be replayed as HIP graphs (BASELINE config 5).  All four ways
    give the same phi-vectors and the same adaptive decisions."""
    import os

    from wxfactory_amd import _lib
    from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun
    from wxfactory_amd.solvers import KiopsWorkspace, kiops

    # the finish kernels against the expressions they replace
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(9)
    n, p, iop, j = 5000, 3, 2, 4
    V = torch.randn((8, n + p), generator=gen, device=DEV, dtype=torch.float64)
    aw = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64)
    uf = torch.randn((n, p), generator=gen, device=DEV, dtype=torch.float64)
    hcol = torch.zeros(9, device=DEV, dtype=torch.float64)
    work = torch.empty(int(lib.wx_kiops_finish_workspace(n + p)), device=DEV, dtype=torch.float64)
    ref = V.clone()
    ref[j, :n] = aw + uf @ ref[j - 1, n:]
    ref[j, n:] = torch.cat((ref[j - 1, n + 1:], ref.new_zeros(1)))
    h = ref[j - iop:j] @ ref[j]
    ref[j] -= h @ ref[j - iop:j]
    nrm = ref[j].norm()
    ref[j] /= nrm
    _lib.check(lib.wx_kiops_finish(V.data_ptr(), V.stride(0), j, n, p, iop, aw.data_ptr(), uf.data_ptr(), hcol.data_ptr(),
                                   work.data_ptr(), torch.cuda.current_stream().cuda_stream), "wx_kiops_finish")
    assert torch.allclose(V, ref, rtol=1e-12, atol=1e-13)
    assert torch.allclose(hcol[j - iop:j], h, rtol=1e-12, atol=1e-13) and abs(float(hcol[j]) - float(nrm)) <= 1e-12 * float(nrm)

    g, rhs, stack = setup
    Q, R = stack("Q"), stack("R")
    dt = float(g["meta/dt_jvp"])
    vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
    vec[1] = R.flatten()
    args = dict(tol=1e-7, m_init=1, mmin=16, mmax=64)
    os.environ["WXHIP_KIOPS_FUSED"] = "0"
    try:
        w_expr, st_expr = kiops([1], lambda v: matvec_fun(v, dt, Q, R, rhs, "complex"), vec, **args)
    finally:
        del os.environ["WXHIP_KIOPS_FUSED"]
    w_fin, st_fin = kiops([1], lambda v: matvec_fun(v, dt, Q, R, rhs, "complex"), vec, **args)
    op = ComplexStepOperator(dt, Q, R, rhs)
    # (WXHIP_JVP_LEAN=0, the generic dual kernel, has no one-call vector build: then op is a plain matvec)
    assert (op.kiops_vector is not None) == (os.environ.get("WXHIP_JVP_LEAN") != "0")
    w_one, st_one = kiops([1], op, vec, **args)
    ws = KiopsWorkspace()
    outs = [kiops([1], op, vec, workspace=ws, graph_token=(Q.data_ptr(), R.data_ptr(), dt), **args) for _ in range(3)]
    assert ws.captures >= 1 and ws.replays >= 2
    scale = w_expr.abs().max()
    for w, st in [(w_fin, st_fin), (w_one, st_one)] + outs:
        assert st[:4] == st_expr[:4] and st[5] == st_expr[5], (st, st_expr)
        assert float((w - w_expr).abs().max()) <= 1e-10 * float(scale)
    for w, st in outs:   # eager first occurrence, captured second, replayed third: identical bits
        assert torch.equal(w, w_one)


def test_hipgraph_captured_matvec(setup):
    """BASELINE config 5, "hipGraph-captured matvec": the complex-step and the finite-difference Jacobian-vector
    products captured once and replayed with one host call."""
    from wxfactory_amd.graph import GraphedFunction
    from wxfactory_amd.matvec import matvec_fun

    g, rhs, stack = setup
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    for method in ("complex", "fd"):
        eager = lambda v: matvec_fun(v, dt, Q, R, rhs, method)  # noqa: E731
        graphed = GraphedFunction(eager, V.flatten())
        for scale in (1.0, -0.37):
            v = (scale * V).flatten()
            assert torch.equal(graphed(v), eager(v)), method

    # (replay against eager timings are reported by bench.py: extra.euler_ini_sizes - no timing assertion in the
    # parity suite)


@pytest.mark.parametrize("order", [3, 4, 5, 6])
def test_multistep_epi_orders(built_lib, order):
    """EPI of orders 3 to 6 (integrators/epi.py:28-141): start-up steps by EPI2, then two regular steps whose
    phi-vectors are assembled from the previous states; KIOPS runs with 2-4 augmented components on the GPU.
    Against the reference's own epi.py + kiops.py from the same state (both sides converge each step to tol 1e-7)."""
    from tests.gpu_util import device_metric
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("epi_multistep_n3_h2_v2")
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
    rhs = RhsEuler3D(plans)
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    Q0, dt = stack("Q"), float(g["meta/dt"])
    epi, Q = Epi(order, rhs, tol=1e-7), Q0
    assert epi.n_prev == {3: 1, 4: 2, 5: 3, 6: 4}[order] and epi.max_phi == {3: 2, 4: 3, 5: 4, 6: 4}[order]
    for _ in range(int(g[f"meta/steps_epi{order}"])):
        Q = epi.step(Q, dt)
    ref, q0 = stack(f"epi{order}").cpu().numpy(), Q0.cpu().numpy()
    ax = (0, 2, 3, 4, 5)
    upd = np.abs(ref - q0).max(axis=ax)
    err = np.abs(Q.cpu().numpy() - ref).max(axis=ax)
    assert (err <= 2e-5 * upd + 1e-13 * np.abs(ref).max(axis=ax)).all(), (err / upd)


@pytest.mark.parametrize("order", [3, 4])
def test_stiffness_resilient_epi(built_lib, order):
    """integrators/epi_stiff.py (time_integrator = epi_stiff3 in config/dcmip20.ini) as simulation.py:336-340 builds it -
    ten EPI2 sub-steps per start-up step, then regular steps whose remainders enter from phi_3 on - with the default
    exponential solver pmex, against the reference's own run from the same state."""
    from tests.gpu_util import device_metric
    from wxfactory_amd.integrators import EpiStiff
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("epi_stiff_n3_h2_v2")
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
    rhs = RhsEuler3D(plans)
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    Q0, dt = stack("Q"), float(g["meta/dt"])
    epi, Q = EpiStiff(order, rhs, tol=1e-7, init_substeps=10, exponential_solver="pmex"), Q0
    assert epi.n_prev == order - 2 and epi.max_phi == order and epi.first_row == 3
    for _ in range(int(g[f"meta/steps_epi{order}"])):
        Q = epi.step(Q, dt)
    ref, q0 = stack(f"epi{order}").cpu().numpy(), Q0.cpu().numpy()
    ax = (0, 2, 3, 4, 5)
    upd = np.abs(ref - q0).max(axis=ax)
    err = np.abs(Q.cpu().numpy() - ref).max(axis=ax)
    assert (err <= 2e-5 * upd + 1e-13 * np.abs(ref).max(axis=ax)).all(), (err / upd)
    with pytest.raises(ValueError, match="Unsupported order"):
        EpiStiff(1, rhs)


@pytest.mark.parametrize("p,taus", [(1, [1.0]), (3, [0.4, 1.0])])
def test_kiops_long_vector_build(built_lib, p, taus, monkeypatch):
    """Vectors too long for the one-workgroup finish take the three streaming stages (wx_kiops_long_a/b_scaled + _c_lazy): same
    phi-vectors and the same adaptive decisions as the array-expression recurrence they replace (WXHIP_KIOPS_LONG=0),
    and the exact result for a diagonal operator."""
    from wxfactory_amd import solvers
    from wxfactory_amd.solvers import kiops

    monkeypatch.setattr(solvers.KiopsWorkspace, "max_fused_len", 4096)   # (make a modest length count as long)
    n = 50_000
    gen = torch.Generator(device=DEV).manual_seed(17 + p)
    lam = -(0.2 + 2.5 * torch.rand(n, generator=gen, device=DEV, dtype=torch.float64))
    u = torch.randn((p + 1, n), generator=gen, device=DEV, dtype=torch.float64)
    class Diagonal:
        linear = True   # exactly linear in v: may be handed un-normalised rows (lazy normalisation)

        def __call__(self, v):
            return lam * v

    A = Diagonal()
    args = dict(tol=1e-10, m_init=12, mmin=10, mmax=40)
    w_long, st_long = kiops(taus, A, u, **args)
    monkeypatch.setenv("WXHIP_KIOPS_LONG", "0")
    w_expr, st_expr = kiops(taus, A, u, **args)
    assert st_long[:4] == st_expr[:4] and st_long[5] == st_expr[5], (st_long, st_expr)
    scale = float(w_expr.abs().max())
    assert float((w_long - w_expr).abs().max()) <= 1e-11 * scale
    # the several-rank form of the same stages (products of the n-long parts all-reduced, the replicated augmented
    # components added once afterwards), taken here on one rank
    monkeypatch.delenv("WXHIP_KIOPS_LONG")
    w_split, st_split = kiops(taus, A, u, _force_split=True, **args)
    assert st_split[:4] == st_long[:4] and float((w_split - w_long).abs().max()) <= 1e-12 * scale
    # the stages above carry the rows' norms as scale factors (lazy normalisation, 9 sweeps per vector); with every row
    # rewritten for its norm (wx_kiops_long_c, 11 sweeps): the same decisions, the same vectors to rounding
    monkeypatch.setenv("WXHIP_KIOPS_LAZY", "0")
    w_eager, st_eager = kiops(taus, A, u, **args)
    monkeypatch.delenv("WXHIP_KIOPS_LAZY")
    assert st_eager[:4] == st_long[:4] and st_eager[5] == st_long[5], (st_eager, st_long)
    assert float((w_eager - w_long).abs().max()) <= 1e-12 * scale
    # exact: w(tau) = sum_k tau^k phi_k(tau lam) u_k
    import math

    def phi(k, z):
        if k == 0:
            return torch.exp(z)
        return (phi(k - 1, z) - 1.0 / math.factorial(k - 1)) / z

    for i, tau in enumerate(taus):
        ref = sum((tau ** k) * phi(k, tau * lam) * u[k] for k in range(p + 1))
        assert float((w_long[i] - ref).abs().max()) <= 1e-8 * float(ref.abs().max())
    # An operator that does NOT declare itself linear - a finite-difference product: truncation term quadratic in v - is
    # always applied to normalised rows, as the reference does (solvers/kiops.py:170): the long-vector stages then rewrite
    # every row for its norm (wx_kiops_long_c) and reproduce the array-expression recurrence, decisions and vectors.
    fd = lambda v: lam * v + 1e-3 * lam * v * v  # noqa: E731
    ws = solvers.KiopsWorkspace()
    w_fd, st_fd = kiops(taus, fd, u, workspace=ws, **args)
    assert float((ws.scales - 1.0).abs().max()) == 0.0          # no row was left un-normalised
    monkeypatch.setenv("WXHIP_KIOPS_LONG", "0")
    w_fd_expr, st_fd_expr = kiops(taus, fd, u, **args)
    monkeypatch.delenv("WXHIP_KIOPS_LONG")
    assert st_fd[:4] == st_fd_expr[:4] and st_fd[5] == st_fd_expr[5], (st_fd, st_fd_expr)
    assert float((w_fd - w_fd_expr).abs().max()) <= 1e-11 * float(w_fd_expr.abs().max())
    # a NaN in the operator ends the solve with an error instead of an endless loop of rejections
    bad = lambda v: lam * v * float("nan")  # noqa: E731
    with pytest.raises(ValueError, match="NaN"):
        kiops([1.0], bad, u[:2], **args)
