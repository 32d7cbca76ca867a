"""The benchmark order n = 8 through every caller, pinned to values produced by the reference itself.

At n = 8 / float64 the library selects its MATRIX-CORE instantiations (v_mfma_f64_4x4x4_4b_f64 for the derivative,
correction and filter contractions): the JVP kernel of matvec_fun (`euler_jvp_body_mf`), the stage kernel of the
explicit time loop with its fused exponential filter (`euler_rhs_kernel<8, double, true>`), the standalone filter
(`expfilter_kernel<8>`).  These are the kernels bench.py times for BASELINE configs 4 and 5; every test below first
asserts (through wx_euler3d_uses_matrix_cores) that it is about to run them, then compares with

  * `Im R(Q + i eps V)` of the reference's rhs (euler3d_c31*_n8_h2_v2: solvers/matvec.py:56-61 on rhs_dfr.py),
  * the reference's matvec_fun / matvec_rat / Tvdrk3.step / Ros2.step / kiops / Epi.step (callers_euler3d_n8_h2_v2),
  * DFROperators.apply_filters (filters_c21_n8_h2_v2) and the filtered time loop (steploop_c21_n8_h2_v2),
  * BASELINE config 5's own combination: config/dcmip21.ini + EPI2 + KIOPS + filter (config5_c21_*).
"""
import os

import numpy as np
import pytest
import torch

from tests.util import tight_tangent, GOLDEN, Golden, golden, halo7, make_oracle, var_err, var_max

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
AX = (0, 2, 3, 4, 5)   # all axes of a stacked state but the variable


def _mc(plan, kernel):
    return int(plan.lib.wx_euler3d_uses_matrix_cores(plan._h, kernel))


def test_matrix_core_instantiations_are_selected(built_lib):
    from tests.gpu_util import make_plan
    from wxfactory_amd import _lib

    g8, g3 = golden("euler3d_c31p_n8_h2_v2"), golden("euler3d_c31p_n3_h4_v2")
    p8, p3 = make_plan(g8, 0), make_plan(g3, 0)
    for k in (_lib.WX_KERNEL_RHS, _lib.WX_KERNEL_STAGE, _lib.WX_KERNEL_BATCH_RHS):
        assert _mc(p8, k) == 1 and _mc(p3, k) == 0
    d8 = p8.twin(torch.complex128, dual=True)
    assert _mc(d8, _lib.WX_KERNEL_BATCH_JVP) == 1
    assert _mc(d8, _lib.WX_KERNEL_JVP) == (0 if os.environ.get("WXHIP_JVP_LEAN") == "0" else 1)
    assert _mc(p8.twin(torch.complex128), _lib.WX_KERNEL_RHS) == 0   # true complex arithmetic: vector pipe
    assert _mc(p3.twin(torch.complex128, dual=True), _lib.WX_KERNEL_JVP) == 0


# ---------------------------------------------------------------------------------------------------------
# the JVP kernel itself, one panel at a time, against Im R(Q + i eps V) of the reference
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["euler3d_c31p_n8_h2_v2", "euler3d_c31_n8_h2_v2", "euler3d_c21p_n8_h2_v2"])
def test_jvp_kernel_n8_matches_reference_complex_step(built_lib, name):
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd import _lib
    from wxfactory_amd.panels import NEIGHBOR, landing_edge
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden(name)
    tight = tight_tangent(name)   # balanced states: numpy.maximum's tie-break decides the tangent (tests/test_oracle_euler3d.py)
    for p in g.metric_panels():
        plan = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV), dtype=torch.complex128, dual=True)
        assert _mc(plan, _lib.WX_KERNEL_JVP) == 1 or os.environ.get("WXHIP_JVP_LEAN") == "0"
        q, v = to_dev(g[f"p{p}/Q"]), to_dev(g[f"p{p}/V"])
        hc = [halo7(h) for h in g.halo(p, True)]
        halo = [to_dev(h) for h in hc]
        ref = g.r(p, True).imag
        o = make_oracle(g, p)
        want = {}
        o.rhs(g.q(p, True), g.halo(p, True), want=want)
        floor = np.maximum(var_max(ref), g.eps * o.cancel_scale(want) * 1e-3)

        # (a) the JVP kernel: dual state formed on load, real tangent stored
        send = torch.zeros((4, plan.edge_count), dtype=torch.complex128, device=DEV)
        plan.jvp_extrap_pack(q, v, g.eps, list(send))
        out = torch.full_like(q, float("nan"))
        plan.jvp(q, v, g.eps, halo, out, 1.0)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.isfinite(got).all()
        ierr = var_err(got, ref) / floor
        assert (ierr <= (1e-10 if tight else 1e-3)).all(), (name, p, ierr)
        # what it packed for the neighbours is what the reference delivered to them
        sent = send.cpu().numpy().reshape(4, 5, g.V, g.H, g.n**2)
        for e in range(4):
            want_e = halo7(g.halo(NEIGHBOR[p][e], True)[landing_edge(p, e)])
            assert np.abs(sent[e].real - want_e.real).max() <= 1e-13 * np.abs(want_e.real).max(), (p, e)
            assert np.abs(sent[e].imag - want_e.imag).max() <= 1e-12 * np.abs(want_e.imag).max(), (p, e)

        # (b) the generic dual-number instantiation of the RHS kernel (vector pipe) on the same state: another kernel,
        # another summation order, the same tangent to rounding
        qc = to_dev(g.q(p, True))
        gen = torch.full_like(qc, float("nan"))
        plan.extrap_pack(qc, None)
        plan.rhs(qc, halo, gen)
        torch.cuda.synchronize()
        assert (var_err(got, gen.cpu().numpy().imag) <= 1e-12 * floor).all(), (name, p)

        # (c) the prepared form (values cached, tangents per product): bit for bit the unprepared result
        sv = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        stn = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        plan.jvp_prepare(q, list(sv))
        plan.jvp_tangent_pack(q, v, g.eps, list(stn))
        hv = [to_dev(np.ascontiguousarray(h.real)) for h in hc]
        ht = [to_dev(np.ascontiguousarray(h.imag)) for h in hc]
        out2 = torch.full_like(q, float("nan"))
        plan.jvp_prepared(q, v, g.eps, hv, ht, out2, 1.0)
        torch.cuda.synchronize()
        assert torch.equal(out, out2), (name, p)
        assert torch.equal(sv, send.real.contiguous()) and torch.equal(stn, send.imag.contiguous())
        plan.close()


# ---------------------------------------------------------------------------------------------------------
# regions at n = 8 with an interior (H >= 3): the launch shape of every multi-GPU run
# ---------------------------------------------------------------------------------------------------------
def _synthetic_case(n, H, V, panel, case, seed=23):
    from oracle.euler3d import Euler3DOracle
    from wxfactory_amd import synthetic

    ops = synthetic.dfr_ops(n)
    m = synthetic.euler3d_metric(n, H, V, panel, "cpu", seed=seed, damping=case in (21, 22))
    q = synthetic.euler3d_state(n, H, V, panel, "cpu", seed=seed)
    om = {"sqrtG_new": m["sqrtG"].numpy(), "inv_sqrtG_new": (1.0 / m["sqrtG"]).numpy(),
          "h_contra_new": m["h_contra"].numpy(), "christoffel": m["christoffel"].numpy(),
          "inv_dzdeta_new": m["inv_dzdeta"].numpy()}
    for d in "ijk":
        om[f"sqrtG_itf_{d}_new"] = m[f"sqrtG_itf_{d}"].numpy()
        om[f"h_contra_itf_{d}_new"] = m[f"h_contra_itf_{d}"].numpy()
    if case in (21, 22):
        om["damp_coef"], om["damp_uref"] = m["damp_coef"].numpy(), m["damp_uref"].numpy()
    bsn = np.tile(m["boundary_sn"].numpy().reshape(H, 1, n), (1, n, 1))
    return ops, m, q, Euler3DOracle(n, H, V, case, ops, om, bsn, bsn, panel=panel)


@pytest.mark.parametrize("H,V,panel,case", [(3, 2, 4, 31), (4, 1, 1, 21)])
def test_regions_compose_at_n8(built_lib, H, V, panel, case):
    """INTERIOR + BOUNDARY == ALL, bit for bit, for the three n = 8 matrix-core kernels (RHS, stage with prepared
    faces and fused filter, JVP), and the ALL result agrees with the pinned oracle."""
    from wxfactory_amd import _lib
    from wxfactory_amd.filters import make_filter
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    n = 8
    ops, m, q, o = _synthetic_case(n, H, V, panel, case)
    md = {k: v.to(DEV) for k, v in m.items()}
    itf = o.extrapolate(q.numpy())
    sends = o.pack_edges(itf)
    halo_np = [sends[1], sends[0], sends[3], sends[2]]
    want = {}
    ref = o.rhs(q.numpy(), halo_np, itf=itf, want=want)
    plan = Euler3DPlan(n, H, V, case, panel, ops, md)
    assert _mc(plan, _lib.WX_KERNEL_RHS) == 1 and _mc(plan, _lib.WX_KERNEL_STAGE) == 1
    qd = q.to(DEV)
    halo = [torch.from_numpy(np.ascontiguousarray(h)).to(DEV) for h in halo_np]
    nan = lambda: torch.full_like(qd, float("nan"))  # noqa: E731

    # plain RHS
    a, b = nan(), nan()
    plan.extrap_pack(qd, None)
    plan.rhs(qd, halo, a, _lib.WX_REGION_ALL)
    plan.rhs(qd, None, b, _lib.WX_REGION_INTERIOR)
    plan.rhs(qd, halo, b, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ax = (1, 2, 3, 4)
    scale = np.maximum(np.abs(ref).max(axis=ax), o.cancel_scale(want))
    assert (np.abs(a.cpu().numpy() - ref).max(axis=ax) <= 1e-10 * scale).all()

    # stage kernel: out = y/3 + 2/3 q + c R(q), filtered, faces of `out` prepared into the other slot
    plan.set_exp_filter(make_filter(0.1, 4, 0.5, np.polynomial.legendre.leggauss(n)[0]))
    y = (qd * 1.01).contiguous()
    outs, sent, flags = [], [], []
    for regions in ((_lib.WX_REGION_ALL,), (_lib.WX_REGION_INTERIOR, _lib.WX_REGION_BOUNDARY)):
        out = nan()
        ns = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        plan.extrap_pack_slot(qd, None, 0)
        for r in regions:
            plan.stage(qd, None if r == _lib.WX_REGION_INTERIOR else halo, y, None, out, 1 / 3, 2 / 3, 1e-3, 0.0, r, 0,
                       list(ns), 2, flag.data_ptr())
        # the faces it prepared (slot 1) serve the next evaluation: R(out) from them == R(out) from a fresh extrapolation
        nxt = nan()
        plan.stage(out, halo, None, None, nxt, 0.0, 0.0, 1.0, 0.0, _lib.WX_REGION_ALL, 1, None, 0, 0)
        torch.cuda.synchronize()
        outs.append((out, nxt))
        sent.append(ns)
        flags.append(int(flag.item()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(sent[0], sent[1])
    assert flags == [0, 0]
    fresh = nan()
    plan.extrap_pack(outs[0][0], None)
    plan.rhs(outs[0][0], halo, fresh, _lib.WX_REGION_ALL)
    torch.cuda.synchronize()
    assert torch.equal(fresh, outs[0][1])
    # ... and the stage's value is the oracle's: filter(y/3 + 2/3 q + c R(q))
    from oracle import filters as ofilt

    F = make_filter(0.1, 4, 0.5, np.polynomial.legendre.leggauss(n)[0])
    stage_ref = ofilt.apply_filter_3d(y.cpu().numpy() / 3 + (2 / 3) * q.numpy() + 1e-3 * ref, m["sqrtG"].numpy(), F)
    got = outs[0][0].cpu().numpy()
    assert (np.abs(got - stage_ref).max(axis=ax) <= 1e-12 * np.abs(stage_ref).max(axis=ax) + 1e-13 * scale).all()

    # JVP kernel
    dual = plan.twin(torch.complex128, dual=True)
    assert _mc(dual, _lib.WX_KERNEL_JVP) == 1 or os.environ.get("WXHIP_JVP_LEAN") == "0"
    gen = torch.Generator(device="cpu").manual_seed(3)
    v = (torch.rand(q.shape, generator=gen, dtype=torch.float64) - 0.5) * q.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
    vd = v.to(DEV)
    eps = 1.4901161193847656e-08
    qc = (q.numpy() + 1j * eps * v.numpy())
    itf_c = o.extrapolate(qc)
    sc = o.pack_edges(itf_c)
    halo_c = [torch.from_numpy(np.ascontiguousarray(h)).to(DEV) for h in (sc[1], sc[0], sc[3], sc[2])]
    ja, jb = nan(), nan()
    dual.jvp_extrap_pack(qd, vd, eps, None)
    dual.jvp(qd, vd, eps, halo_c, ja, 1.0 / eps, _lib.WX_REGION_ALL)
    dual.jvp(qd, vd, eps, None, jb, 1.0 / eps, _lib.WX_REGION_INTERIOR)
    dual.jvp(qd, vd, eps, halo_c, jb, 1.0 / eps, _lib.WX_REGION_BOUNDARY)
    torch.cuda.synchronize()
    assert torch.equal(ja, jb)
    wc = {}
    jref = o.rhs(qc, [sc[1], sc[0], sc[3], sc[2]], itf=itf_c, want=wc).imag / eps
    jscale = np.maximum(np.abs(jref).max(axis=ax), 1e-3 * o.cancel_scale(wc))
    assert (np.abs(ja.cpu().numpy() - jref).max(axis=ax) <= 1e-10 * jscale).all()


# ---------------------------------------------------------------------------------------------------------
# the callers at n = 8 (reference: matvec.py, tvdrk3.py, ros2.py + fgmres.py, kiops.py, epi.py)
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def callers8(built_lib):
    from tests.gpu_util import device_metric
    from wxfactory_amd import _lib
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("callers_euler3d_n8_h2_v2")
    assert g.n == 8
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}
    assert _mc(plans[0], _lib.WX_KERNEL_RHS) == 1
    rhs = RhsEuler3D(plans)
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    return g, rhs, stack


def _rel(a, b):
    a = a.cpu().numpy().reshape(b.shape)
    return np.abs(a - b).max(axis=AX) / np.abs(b).max(axis=AX)


@pytest.mark.parametrize("batched", [False, True])
def test_n8_matvecs_match_reference(callers8, batched):
    """matvec_fun (complex step, finite difference) and matvec_rat through the per-tile launches (the E7 path: the
    matrix-core JVP kernel, prepared and unprepared) and through the one-launch-per-phase batch."""
    from wxfactory_amd import _lib
    from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun, matvec_rat

    g, rhs, stack = callers8
    Q, V, R = stack("Q"), stack("V"), stack("R")
    dt = float(g["meta/dt_jvp"])
    rhs.batched = batched
    try:
        assert (_rel(rhs(Q), R.cpu().numpy()) < 1e-11).all()
        jc = matvec_fun(V.flatten(), dt, Q, R, rhs, "complex")
        dual = rhs._jvp_plans()[0]
        assert _mc(dual, _lib.WX_KERNEL_BATCH_JVP if batched else _lib.WX_KERNEL_JVP) == 1 \
            or os.environ.get("WXHIP_JVP_LEAN") == "0"
        ref = stack("jvp_complex").cpu().numpy()
        assert (_rel(jc, ref) < 1e-9).all(), _rel(jc, ref)
        jf = matvec_fun(V.flatten(), dt, Q, R, rhs, "fd")
        assert (_rel(jf, stack("jvp_fd").cpu().numpy()) < 1e-6).all()
        ra = matvec_rat(V.flatten(), dt, Q, R, rhs)
        assert (_rel(ra, stack("rat").cpu().numpy()) < 1e-6).all()
        if not batched:   # the prepared operator a Krylov solve holds
            op = ComplexStepOperator(dt, Q, R, rhs)
            assert rhs._jvp_is_prepared(Q)
            jp = op(V.flatten())
            rhs.jvp_release()
            assert torch.equal(jp, jc)
    finally:
        rhs.jvp_release()
        rhs.batched = True


@pytest.mark.parametrize("fused,pipeline", [(False, False), (True, False), (True, True)])
def test_n8_tvdrk3_step(callers8, fused, pipeline):
    from wxfactory_amd.integrators import Tvdrk3

    g, rhs, stack = callers8
    stepper = Tvdrk3(rhs, fused=fused, pipeline=pipeline)
    rhs.batched = False   # per-tile launches: with pipeline=True the stage (PIPE) instantiation
    try:
        Qn = stepper.step(stack("Q"), float(g["meta/dt_rk"]))
    finally:
        rhs.batched = True
    ref, q0 = stack("rk3").cpu().numpy(), stack("Q").cpu().numpy()
    dq = np.abs(ref - q0).max(axis=AX)
    err = np.abs(Qn.cpu().numpy() - ref).max(axis=AX)
    assert (dq[[0, 1, 2, 4]] > 0).all()
    assert (err <= 1e-9 * dq + 1e-14 * np.abs(ref).max(axis=AX)).all(), (err, dq)


@pytest.mark.parametrize("ortho", ["igs", "cgs"])
def test_n8_ros2_fgmres_step(callers8, ortho):
    from wxfactory_amd.integrators import Ros2

    g, rhs, stack = callers8
    rhs.batched = False
    try:
        ros = Ros2(rhs, tol=1e-9, gmres_restart=30, ortho=ortho)
        Qn = ros.step(stack("Q"), float(g["meta/dt_jvp"]))
    finally:
        rhs.batched = True
    info = ros.solver_info
    assert info["flag"] == 0 and info["rel_residual"] < 1e-9, info
    ref, q0 = stack("ros2").cpu().numpy(), stack("Q").cpu().numpy()
    upd = np.abs(ref - q0).max(axis=AX)
    err = np.abs(Qn.cpu().numpy() - ref).max(axis=AX)
    assert (err <= 1e-7 * upd).all(), (err / upd, info["iterations"])


def test_n8_kiops_and_epi2_step(callers8):
    from wxfactory_amd.integrators import Epi
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.solvers import kiops

    g, rhs, stack = callers8
    Q, R = stack("Q"), stack("R")
    dt = float(g["meta/dt_jvp"])
    vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
    vec[1] = R.flatten()
    rhs.batched = False   # per-tile launches and the prepared matrix-core JVP: what config 4/5 run at E7
    try:
        op = ComplexStepOperator(dt, Q, R, rhs)
        assert rhs._jvp_is_prepared(Q)
        phiv, stats = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
        rhs.jvp_release()
        ref_stats = g["p0/kiops_stats"]
        assert [int(stats[i]) for i in (0, 1, 2, 3, 5)] == [int(ref_stats[i]) for i in (0, 1, 2, 3, 5)], (stats, ref_stats)
        # (the error ESTIMATE of the last substep - a difference of two nearly equal small-matrix results after 384 Krylov
        # vectors, 6 substeps and 10 rejections - moves by a few per cent with the summation order; n = 3: 1e-3)
        assert abs(float(stats[4]) - float(ref_stats[4])) <= 0.2 * float(ref_stats[4])
        ref = stack("kiops_phiv").cpu().numpy()
        err = np.abs(phiv.cpu().numpy().reshape(ref.shape) - ref).max(axis=AX) / np.abs(ref).max(axis=AX)
        assert (err < 1e-8).all(), err
        Qn = Epi(2, rhs, tol=1e-7).step(Q, dt)
    finally:
        rhs.jvp_release()
        rhs.batched = True
    refq, q0 = stack("epi2").cpu().numpy(), Q.cpu().numpy()
    upd = np.abs(refq - q0).max(axis=AX)
    assert (np.abs(Qn.cpu().numpy() - refq).max(axis=AX) <= 1e-7 * upd).all()


def test_n8_kiops_long_build_lets_the_matvec_form_the_vector(callers8, monkeypatch):
    """Long vectors, one augmented component, the prepared matrix-core JVP (what EPI2 + KIOPS runs at E7): the n-long part of
    the next Krylov vector, A V[j-1] + u a, is formed in the product's own store (wx_euler3d_jvp_prepared_axpy) together with
    its products with the two rows it is orthogonalised against (partial sums per workgroup, wx_kiops_long_a_finish) - or,
    WXHIP_KIOPS_STORE_DOTS=0, the first streaming stage takes the products (wx_kiops_long_a_formed).  The reference's statistics exactly, its phi-vector to
    1e-8, and the vectors of the unfused stages to rounding."""
    from wxfactory_amd import solvers
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.solvers import kiops

    g, rhs, stack = callers8
    Q, R = stack("Q"), stack("R")
    dt = float(g["meta/dt_jvp"])
    vec = torch.zeros((2, R.numel()), dtype=torch.float64, device=DEV)
    vec[1] = R.flatten()
    monkeypatch.setattr(solvers.KiopsWorkspace, "max_fused_len", 4096)   # (make this length count as long)
    stored, with_products, folded = [0], [0], [0]
    plain = ComplexStepOperator.axpy_into

    def counting(self, *a, **k):
        done = plain(self, *a, **k)
        stored[0] += 1 if done else 0
        with_products[0] += isinstance(done, tuple)
        folded[0] += "fix" in k
        return done

    monkeypatch.setattr(ComplexStepOperator, "axpy_into", counting)
    rhs.batched = False
    try:
        op = ComplexStepOperator(dt, Q, R, rhs)
        assert rhs._jvp_is_prepared(Q) and rhs.jvp_fuses_store(Q)
        assert op.fold_ready()
        phiv, stats = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
        assert stored[0] == with_products[0] == int(stats[2]), (stored, with_products, stats)   # every Krylov vector of the solve
        # ... and the last stage of every vector but the last of a pass - the subtraction of its projections and its norm - rode
        # on the NEXT product's tangent extrapolation (wx_euler3d_jvp_tangent_extrap_pack_fix): no sweep of its own
        passes = int(stats[3])   # (at most one pass of new vectors per matrix exponential; its last vector completes itself)
        assert int(stats[2]) - passes <= folded[0] < int(stats[2]), (folded, stats)
        monkeypatch.setenv("WXHIP_KIOPS_FOLD", "0")            # every vector completes its own last stage (wx_kiops_long_b_scaled)
        before = folded[0]
        phivf, statsf = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
        monkeypatch.delenv("WXHIP_KIOPS_FOLD")
        assert folded[0] == before and stored[0] == with_products[0] == 2 * int(stats[2])
        assert [int(statsf[i]) for i in (0, 1, 2, 3, 5)] == [int(stats[i]) for i in (0, 1, 2, 3, 5)], (statsf, stats)
        assert float((phivf - phiv).abs().max()) <= 1e-9 * float(phiv.abs().max())
        stored[0] = with_products[0] = int(stats[2])
        monkeypatch.setenv("WXHIP_KIOPS_STORE_DOTS", "0")      # the vector from the store, its products from a sweep
        phiv1, stats1 = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
        assert stored[0] == 2 * int(stats[2]) and with_products[0] == int(stats[2])
        monkeypatch.setenv("WXHIP_KIOPS_STORE_AXPY", "0")      # neither
        phiv0, stats0 = kiops([1], op, vec, tol=1e-7, m_init=1, mmin=16, mmax=64)
        assert stored[0] == 2 * int(stats[2])
        rhs.jvp_release()
    finally:
        rhs.jvp_release()
        rhs.batched = True
    ref_stats = g["p0/kiops_stats"]
    for st in (stats, stats1, stats0):
        assert [int(st[i]) for i in (0, 1, 2, 3, 5)] == [int(ref_stats[i]) for i in (0, 1, 2, 3, 5)], (st, ref_stats)
    ref = stack("kiops_phiv").cpu().numpy()
    err = np.abs(phiv.cpu().numpy().reshape(ref.shape) - ref).max(axis=AX) / np.abs(ref).max(axis=AX)
    assert (err < 1e-8).all(), err
    # (384 Krylov vectors, 6 substeps: the one rounding that differs - a x + b z in one store - has grown to 2e-10 by the end)
    assert float((phiv - phiv0).abs().max()) <= 1e-8 * float(phiv0.abs().max())
    assert float((phiv1 - phiv0).abs().max()) <= 1e-8 * float(phiv0.abs().max())


# ---------------------------------------------------------------------------------------------------------
# BASELINE config 5's own combination: config/dcmip21.ini (topography + sponge), EPI2 + KIOPS, per-step filter
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,batched", [("config5_c21_n4_h2_v3", True), ("config5_c21_n8_h2_v2", False),
                                          ("config5_c21_n8_h2_v2", True)])
def test_config5_epi2_kiops_filter_on_dcmip21(built_lib, name, batched):
    """integrators/epi.py:81-360 + solvers/kiops.py + operators.apply_filters on the Schaer-mountain state, with nothing
    reference-supplied but the initial state: own geometry (topography, sponge fields), own filter matrix.  The adaptive
    KIOPS controller must take the reference's decisions step by step (substeps, rejections, Krylov vectors,
    exponentials, final basis size)."""
    from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.integrators import Epi, StepLoop
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, H, V, case = (int(g[f"meta/{k}"]) for k in ("n", "H", "V", "case_number"))
    assert case == 21
    topo = topography_for_case(case, planet_for_case(case)[0])
    metrics, plans = {}, {}
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, float(g["meta/ztop"]), case, topo=topo)
        metrics[p] = metric3d_torch(t, DEV)
        plans[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metrics[p])
    F = make_filter(float(g["meta/expfilter_strength"]), int(g["meta/expfilter_order"]), float(g["meta/expfilter_cutoff"]),
                    np.polynomial.legendre.leggauss(n)[0])
    assert np.abs(F - g["ops/expfilter"]).max() < 1e-14
    rhs = RhsEuler3D(plans)
    rhs.batched = batched
    flag = NanFlag(DEV)
    epi = Epi(2, rhs, tol=float(g["meta/tolerance"]))
    filt = ExpFilter3D(F, [metrics[p]["sqrtG"] for p in range(6)])
    loop = StepLoop(epi, filt, flag)
    stack = lambda key: np.stack([g[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    Q = torch.from_numpy(stack("Q")).to(DEV)
    dt, nsteps = float(g["meta/dt"]), int(g["meta/nsteps"])
    ref_stats = g["meta/kiops_stats"]
    prev = stack("Q")
    for i in range(nsteps):
        Qu = epi.step(Q, dt)
        info = epi.solver_info
        got = [int(info[k]) for k in ("substeps", "rejected", "iterations", "exps", "krylov_size")]
        assert got == [int(ref_stats[i][j]) for j in (0, 1, 2, 3, 5)], (i, info, ref_stats[i])
        ref_u = stack(f"Q{i + 1}_unfiltered")
        upd = np.abs(ref_u - prev).max(axis=AX)
        err = np.abs(Qu.cpu().numpy() - ref_u).max(axis=AX)
        assert (err <= 1e-6 * upd + 1e-13 * np.abs(ref_u).max(axis=AX)).all(), (i, err / upd)
        Q = filt(Qu)
        flag.check(Q)
        flag.raise_if_set()
        ref_f = stack(f"Q{i + 1}")
        errf = np.abs(Q.cpu().numpy() - ref_f).max(axis=AX)
        assert (errf <= 1e-6 * upd + 1e-13 * np.abs(ref_f).max(axis=AX)).all(), (i, errf / upd)
        prev = ref_f
    # the same through StepLoop (= Simulation.step's body) from the start
    epi2 = Epi(2, rhs, tol=float(g["meta/tolerance"]))
    Ql = StepLoop(epi2, ExpFilter3D(F, [metrics[p]["sqrtG"] for p in range(6)]), flag).run(
        torch.from_numpy(stack("Q")).to(DEV), dt, nsteps)
    assert torch.equal(Ql, Q)
    del loop
