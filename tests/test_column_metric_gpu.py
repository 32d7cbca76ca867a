"""Column form of the metric (wx_euler3d_plan_set_column_metric): on a shallow atmosphere without topography the reference's
metric arrays are the same on all levels to rounding; whole-tile launches then read one slab per column and field.
Against the reference's R on its own DCMIP 3-1 fixtures (the full arrays are the fixture's, the slabs are cut from them)
and against the general kernel on the same plan inputs; a mountain (DCMIP 2-1) is refused."""
import numpy as np
import pytest
import torch

from tests.util import golden, var_max

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["euler3d_c31p_n8_h2_v2", "euler3d_c31p_n3_h4_v2", "euler3d_c31p_n2_h4_v3", "euler3d_c31p_n6_h2_v2",
                                  "euler3d_c31_n8_h2_v2"])
def test_column_metric_rhs_matches_reference_and_general_kernel(built_lib, name):
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd import _lib
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden(name)
    for p in g.metric_panels():
        m = device_metric(g, p, DEV)
        general = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m)
        column = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m, column_metric=True)
        assert column.column_metric and not general.column_metric
        assert int(_lib.load().wx_euler3d_plan_has_column_metric(column._h)) == 1
        q = to_dev(g[f"p{p}/Q"])
        halo = [to_dev(h) for h in g.halo(p, False)]
        outs = []
        for plan in (general, column):
            send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
            plan.extrap_pack(q, list(send))
            out = torch.full_like(q, float("nan"))
            plan.rhs(q, halo, out, _lib.WX_REGION_ALL)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        ref = g.r(p, False)
        scale = np.maximum(var_max(ref), 1e-30)
        # the slabs differ from the full arrays by the rounding of the reference's own metric (1e-14): the two kernels agree to
        # that, on the scale of the terms that cancel in R (balanced states: |R| is 1e-6 of them)
        from tests.util import make_oracle

        o = make_oracle(g, p)
        want = {}
        o.rhs(g.q(p, False), g.halo(p, False), want=want)
        cancel = o.cancel_scale(want)
        d = np.abs(outs[1] - outs[0]).max(axis=(1, 2, 3, 4))
        assert (d <= 1e-12 * np.maximum(scale, cancel)).all(), (p, d / np.maximum(scale, cancel))
        err = np.abs(outs[1] - ref).max(axis=(1, 2, 3, 4))
        assert (err <= 1e-10 * np.maximum(scale, cancel)).all(), (p, err / np.maximum(scale, cancel))
        # the split launches of the multi-GPU path on the same plan: INTERIOR + BOUNDARY == ALL, bit for bit
        if g.H > 2:
            out2 = torch.full_like(q, float("nan"))
            column.rhs(q, None, out2, _lib.WX_REGION_INTERIOR)
            column.rhs(q, halo, out2, _lib.WX_REGION_BOUNDARY)
            torch.cuda.synchronize()
            assert np.array_equal(out2.cpu().numpy(), outs[1])


def test_column_metric_is_refused_over_a_mountain(built_lib):
    from tests.gpu_util import device_metric
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden("euler3d_c21_n4_h3_v4")
    p = g.metric_panels()[0]
    m = device_metric(g, p, DEV)
    with pytest.raises(ValueError, match="not the same on all levels"):
        Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m, column_metric=True)
    assert not Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m, column_metric="auto").column_metric


@pytest.mark.parametrize("name", ["euler3d_c31p_n8_h2_v2", "euler3d_c31p_n3_h4_v2"])
def test_column_metric_jvp_matches_general_kernel_and_reference(built_lib, name):
    """The complex-step JVP kernels (wx_euler3d_jvp, and the prepared form) on a dual plan with the column slabs: against the
    same kernels on the full arrays and against Im R(Q + i eps V) of the reference."""
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd import _lib
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    g = golden(name)
    for p in g.metric_panels():
        m = device_metric(g, p, DEV)
        q, v = to_dev(g[f"p{p}/Q"]), to_dev(g[f"p{p}/V"])
        halo = [to_dev(h) for h in g.halo(p, True)]
        res = []
        for col in (False, True):
            plan = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m, dtype=torch.complex128, dual=True, column_metric=col)
            assert plan.column_metric == col
            send = torch.zeros((4, plan.edge_count), dtype=torch.complex128, device=DEV)
            plan.jvp_extrap_pack(q, v, g.eps, list(send))
            out = torch.full_like(q, float("nan"))
            plan.jvp(q, v, g.eps, halo, out, 1.0, _lib.WX_REGION_ALL)
            torch.cuda.synchronize()
            res.append(out.cpu().numpy())
        if g.H > 2:   # (plan: the column one of the loop's last pass) split launches == whole-tile launch
            out2 = torch.full_like(q, float("nan"))
            plan.jvp(q, v, g.eps, None, out2, 1.0, _lib.WX_REGION_INTERIOR)
            plan.jvp(q, v, g.eps, halo, out2, 1.0, _lib.WX_REGION_BOUNDARY)
            torch.cuda.synchronize()
            assert np.array_equal(out2.cpu().numpy(), res[1])
        ref = g.r(p, True).imag
        scale = var_max(ref)
        d = np.abs(res[1] - res[0]).max(axis=(1, 2, 3, 4))
        assert (d <= 1e-11 * scale).all(), (p, d / scale)
        err = np.abs(res[1] - ref).max(axis=(1, 2, 3, 4))
        assert (err <= 1e-9 * scale).all(), (p, err / scale)


def test_column_plans_under_the_rhs_object_and_the_matvec(built_lib):
    """Six column plans under RhsEuler3D (per-tile launches, as whole E7 panels take them): R(Q) and the prepared
    complex-step matvec - whose dual twin plans inherit the slabs - against the reference's values of the callers fixture;
    the batched launches of small tiles keep the general kernels and agree."""
    from tests.gpu_util import device_metric
    from tests.util import Golden
    from wxfactory_amd.matvec import ComplexStepOperator
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = Golden("callers_euler3d_n8_h2_v2")
    plans = {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV), column_metric="auto") for p in range(6)}
    assert all(pl.column_metric for pl in plans.values())
    stack = lambda key: torch.from_numpy(np.stack([g[f"p{p}/{key}"] for p in range(6)])).to(DEV)  # noqa: E731
    Q, V, R = stack("Q"), stack("V"), stack("R")
    ax = (0, 2, 3, 4, 5)
    rhs = RhsEuler3D(plans)
    outs = {}
    for batched in (False, True):
        rhs.batched = batched
        r = rhs(Q)
        err = (r - R).abs().amax(dim=ax) / R.abs().amax(dim=ax)
        assert (err < 1e-11).all(), (batched, err)
        op = ComplexStepOperator(float(g["meta/dt_jvp"]), Q, R, rhs)
        jv = op(V.flatten()).reshape(Q.shape)
        rhs.jvp_release()
        ref = stack("jvp_complex")
        errj = (jv - ref).abs().amax(dim=ax) / ref.abs().amax(dim=ax)
        assert (errj < 1e-9).all(), (batched, errj)
        outs[batched] = (r, jv)
        if not batched:   # the store that forms a (A v) + b z and leaves the vector's products (KIOPS), column kernels
            op = ComplexStepOperator(float(g["meta/dt_jvp"]), Q, R, rhs)
            coef = torch.tensor([1.25, 0.5], dtype=torch.float64, device=DEV)
            z, row = R.flatten().contiguous(), Q.flatten().contiguous()
            out = torch.full_like(z, float("nan"))
            part, count = op.axpy_into(V.flatten().contiguous(), out, z, coef[0:1].data_ptr(), coef[1:2].data_ptr(), [row])
            torch.cuda.synchronize()
            want = 1.25 * jv.flatten() + 0.5 * z
            assert float((out - want).abs().max()) <= 1e-14 * float(want.abs().max())
            got = float(part[: 2 * count].view(count, 2).sum(dim=0)[0])
            assert abs(got - float(torch.dot(row, out))) <= 1e-12 * float(row.norm() * out.norm())
            rhs.jvp_release()
    assert all(pl.column_metric for pl in rhs._jvp_plans().values())   # the dual twins took the slabs
    assert float((outs[True][0] - outs[False][0]).abs().max()) <= 1e-12 * float(R.abs().max())


@pytest.mark.parametrize("name", ["euler3d_c31p_n8_h2_v2", "euler3d_c31p_n3_h4_v2"])
def test_column_metric_stage_pipeline(built_lib, name):
    """The stage kernels on column plans (wx_euler3d_stage with the slabs: the RK stage, its extrapolation epilogue and the
    fused per-step filter): three pipelined SSP-RK3 stages and a filtered last stage against the same stages on the full
    arrays - and RhsEuler3D(column_metric="auto") switches the plans it is given."""
    from tests.gpu_util import device_metric, to_dev
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    g = golden(name)
    panels = g.metric_panels()
    if len(panels) < 6:   # the n = 8 fixture stores two panels: a two-tile "sphere" cannot exchange - one tile, fixture halos
        pytest.skip("needs all six panels") if name != "euler3d_c31p_n8_h2_v2" else None
    if len(panels) == 6:
        mk = lambda: {p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, DEV)) for p in range(6)}  # noqa: E731
        full, col = RhsEuler3D(mk()), RhsEuler3D(mk(), column_metric="auto")
        assert all(pl.column_metric for pl in col.plans.values()) and not any(pl.column_metric for pl in full.plans.values())
        Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
        outs = []
        for rhs in (full, col):
            rhs.batched = False
            dt = 1e-3
            Q1 = rhs.stage(Q, None, 0.0, 1.0, dt)
            Q2 = rhs.stage(Q1, Q, 0.75, 0.25, 0.25 * dt)
            Q3 = rhs.stage(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
            torch.cuda.synchronize()
            outs.append(Q3)
        scale = outs[0].abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
        assert ((outs[1] - outs[0]).abs() <= 1e-13 * scale).all()
        return
    # n = 8 (matrix cores): one tile driven through the plan, halos from the fixture; stage + epilogue on slabs vs full arrays
    from wxfactory_amd import _lib

    p = panels[0]
    m = device_metric(g, p, DEV)
    q = to_dev(g[f"p{p}/Q"])
    halo = [to_dev(h) for h in g.halo(p, False)]
    res = []
    for colm in (False, True):
        plan = Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, m, column_metric=colm)
        plan.reserve(_lib.WX_RESERVE_STAGE)
        send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        nsend = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        plan.extrap_pack_slot(q, list(send), 0)
        out = torch.full_like(q, float("nan"))
        plan.stage(q, halo, None, None, out, 0.0, 1.0, 1e-3, 0.0, _lib.WX_REGION_ALL, 0, list(nsend), 1)
        # the faces the epilogue prepared in slot 1 are those of `out`: a second stage from slot 1 equals one that extrapolates
        out2 = torch.full_like(q, float("nan"))
        plan.stage(out, halo, q, None, out2, 0.75, 0.25, 0.25e-3, 0.0, _lib.WX_REGION_ALL, 1, list(send), 1)
        chk = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
        plan.extrap_pack_slot(out, list(chk), 0)
        torch.cuda.synchronize()
        assert torch.equal(chk, nsend)   # edge lines of the epilogue == those of the extrapolation kernel, bit for bit
        res.append((out.cpu().numpy(), out2.cpu().numpy(), nsend.cpu().numpy()))
        assert _lib.load().wx_euler3d_uses_matrix_cores(plan._h, _lib.WX_KERNEL_STAGE) == 1
    for a, b in zip(res[0], res[1]):
        sc = np.abs(a).max(axis=tuple(range(1, a.ndim)), keepdims=True)
        assert (np.abs(a - b) <= 1e-13 * sc).all()
