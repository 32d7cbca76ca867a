"""Per-step filters: exponential modal filter, NaN flag, 2-D sponge.

CPU: the oracle restatement against values produced by the reference's DFROperators.apply_filters.
GPU: the HIP kernels through the C ABI against the same fixtures and the oracle."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN

FIXTURES = ["filters_c21_n4_h3_v4", "filters_c21_n5_h2_v2", "filters_c21_n8_h2_v2"]   # n = 8: the matrix-core instantiation


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _rel(a, b):
    ax = tuple(i for i in range(b.ndim) if i != b.ndim - 5)
    den = np.abs(b).max(axis=ax)
    return np.abs(a - b).max(axis=ax) / np.where(den > 0, den, 1.0)


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_filter_matrix_and_application(name):
    from oracle import filters

    g = _load(name)
    F = filters.make_filter(float(g["meta/expfilter_strength"]), int(g["meta/expfilter_order"]),
                            float(g["meta/expfilter_cutoff"]), g["ops/solution_points"])
    assert np.abs(F - g["ops/expfilter"]).max() < 1e-14
    for p in range(6):
        out = filters.apply_filter_3d(g[f"p{p}/Q"], g[f"p{p}/metric/sqrtG_new"], F)
        assert (_rel(out, g[f"p{p}/R"]) < 1e-14).all(), p
    assert not filters.has_nan(g["p0/Q"])


def test_host_make_filter_matches_reference():
    from wxfactory_amd.filters import make_filter, sponge_profile

    for name in FIXTURES:
        g = _load(name)
        F = make_filter(float(g["meta/expfilter_strength"]), int(g["meta/expfilter_order"]),
                        float(g["meta/expfilter_cutoff"]), g["ops/solution_points"])
        assert np.abs(F - g["ops/expfilter"]).max() < 1e-14
        assert np.abs(F.sum(axis=1) - 1.0).max() < 1e-13  # mode 0 untouched: constants pass through
    z = np.linspace(0.0, 19500.0, 40)
    b = sponge_profile(z, 19500.0, 9500.0, 5.0)
    assert (b[z < 10000.0] == 0).all() and abs(b[-1] - 0.2) < 1e-15 and (np.diff(b) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("cplx", [False, True])
def test_gpu_expfilter_matches_reference(built_lib, name, cplx):
    import torch

    from wxfactory_amd.filters import ExpFilter3D, NanFlag

    dev = "cuda:0"
    g = _load(name)
    sg = [torch.from_numpy(g[f"p{p}/metric/sqrtG_new"]).to(dev) for p in range(6)]
    flag = NanFlag(dev)
    filt = ExpFilter3D(g["ops/expfilter"], sg, flag)
    # n = 8, float64: the three passes run on the matrix cores (complex states keep the vector pipe)
    assert int(filt.lib.wx_expfilter_uses_matrix_cores(filt._h, 0)) == int(int(g["meta/n"]) == 8)
    assert int(filt.lib.wx_expfilter_uses_matrix_cores(filt._h, 1)) == 0
    Q = np.stack([g[f"p{p}/Q"] for p in range(6)])
    R = np.stack([g[f"p{p}/R"] for p in range(6)])
    if cplx:  # linear operator: the imaginary part filters like the real one
        Q = Q + 0.5j * Q[:, ::-1]
        R = R + 0.5j * R[:, ::-1]
    Qd = torch.from_numpy(Q).to(dev)
    out = filt(Qd)
    got = out.cpu().numpy()
    for p in range(6):
        assert (_rel(got[p].real, R[p].real) < 1e-13).all(), p
        if cplx:
            assert (_rel(got[p].imag, R[p].imag) < 1e-13).all(), p
    flag.raise_if_set()  # clean state: no exception
    # in place, one panel at a time
    one = ExpFilter3D(g["ops/expfilter"], sg[2:3])
    q2 = Qd[2].clone()
    assert one(q2, out=q2) is q2 and torch.equal(q2, out[2])


@pytest.mark.gpu
def test_gpu_nan_flag(built_lib):
    import torch

    from wxfactory_amd.filters import ExpFilter3D, NanFlag

    dev = "cuda:0"
    g = _load(FIXTURES[0])
    sg = [torch.from_numpy(g["p0/metric/sqrtG_new"]).to(dev)]
    flag = NanFlag(dev)
    filt = ExpFilter3D(g["ops/expfilter"], sg, flag)
    Q = torch.from_numpy(g["p0/Q"]).to(dev)
    flag.check(Q)
    flag.raise_if_set()
    bad = Q.clone()
    bad[4, 1, 2, 0, 7] = float("nan")
    flag.check(bad)
    with pytest.raises(ValueError, match="NaN"):
        flag.raise_if_set()
    flag.raise_if_set()  # the flag was cleared
    filt(bad)            # the filter raises it too (the NaN spreads over its element)
    with pytest.raises(ValueError, match="NaN"):
        flag.raise_if_set()
    inf = Q.clone()
    inf[0, 0, 0, 0, 0] = float("inf")  # like numpy.isnan: an infinity alone is not flagged
    flag.check(inf)
    flag.raise_if_set()
    c = torch.complex(Q, torch.zeros_like(Q))
    c[1, 0, 0, 0, 3] = complex(0.0, float("nan"))
    flag.check(c)
    with pytest.raises(ValueError, match="NaN"):
        flag.raise_if_set()


@pytest.mark.gpu
def test_gpu_sponge_2d(built_lib):
    import torch

    from wxfactory_amd.filters import sponge_2d, sponge_profile

    rng = np.random.default_rng(5)
    X3 = rng.uniform(0.0, 19500.0, (6, 7, 25))
    beta = sponge_profile(X3, 19500.0, 9500.0, 5.0)
    Q = rng.uniform(-1.0, 1.0, (4, 6, 7, 25))
    want = Q.copy()
    want[2] = (1.0 / (1.0 + beta * 2.5)) * Q[2]
    got = sponge_2d(torch.from_numpy(Q).to("cuda:0"), torch.from_numpy(beta).to("cuda:0"), 2.5)
    got = got.cpu().numpy()
    assert np.array_equal(got[[0, 1, 3]], want[[0, 1, 3]])
    assert np.abs(got[2] - want[2]).max() <= 4e-16  # 1 + beta*dt is one fused multiply-add on the GPU


@pytest.mark.gpu
def test_gpu_step_loop_filters_and_flags(built_lib):
    """StepLoop = integrator step + filter + NaN check, against the same pieces applied by hand."""
    import torch

    from tests.gpu_util import device_metric
    from tests.util import Golden
    from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter
    from wxfactory_amd.integrators import StepLoop, Tvdrk3
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    dev = "cuda:0"
    g = Golden("callers_euler3d_n3_h3_v2")
    metrics = {p: device_metric(g, p, dev) for p in range(6)}
    rhs = RhsEuler3D({p: Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, metrics[p]) for p in range(6)})
    F = make_filter(0.1, 4, 0.5, np.polynomial.legendre.leggauss(g.n)[0])
    flag = NanFlag(dev)
    filt = ExpFilter3D(F, [metrics[p]["sqrtG"] for p in range(6)])
    loop = StepLoop(Tvdrk3(rhs), filt, flag, check_every=2)
    assert filt.nan_flag is flag
    Q0 = torch.from_numpy(np.stack([g[f"p{p}/Q"] for p in range(6)])).to(dev)
    dt = float(g["meta/dt_rk"])
    Qa = loop.run(Q0, dt, 2)
    plain = Tvdrk3(rhs, pipeline=False)
    Qb = Q0
    for _ in range(2):
        Qb = ExpFilter3D(F, filt.sqrtG)(plain.step(Qb, dt))
    scale = Qb.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    assert ((Qa - Qb).abs() <= 1e-13 * scale).all()
    bad = Q0.clone()
    bad[3, 0, 1, 1, 1, 5] = float("nan")
    loop.step(bad, dt)  # step 3: flag raised on the device, not fetched yet
    with pytest.raises(ValueError, match="NaN"):
        loop.step(Q0, dt)  # step 4: fetched


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["steploop_c21_n4_h2_v3", "steploop_c21_n8_h2_v2"])
@pytest.mark.parametrize("pipeline", [False, True])
def test_gpu_time_loop_matches_reference_run(built_lib, pipeline, name):
    """Five (n = 8: three) steps of the reference's own loop (Tvdrk3.step + apply_filters, simulation.py:147-155) on the Schaer
    mountain case, reproduced with nothing reference-supplied but the initial state: geometry3d metric + sponge,
    pipelined SSP-RK3 stages, the filter kernel with its NaN flag."""
    import torch

    from wxfactory_amd.filters import ExpFilter3D, NanFlag, make_filter
    from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
    from wxfactory_amd.integrators import StepLoop, Tvdrk3
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D
    from wxfactory_amd.synthetic import dfr_ops

    dev = "cuda:0"
    g = _load(name)
    n, H, V, case = (int(g[f"meta/{k}"]) for k in ("n", "H", "V", "case_number"))
    topo = topography_for_case(case, planet_for_case(case)[0])
    metrics, plans = {}, {}
    for p in range(6):
        t = CubedSphere3DTile(n, H, V, p, float(g["meta/ztop"]), case, topo=topo)
        metrics[p] = metric3d_torch(t, dev)
        plans[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metrics[p])
    F = make_filter(1e-3, 4, 0.5, np.polynomial.legendre.leggauss(n)[0])
    assert np.abs(F - g["ops/expfilter"]).max() < 1e-14
    flag = NanFlag(dev)
    rhs = RhsEuler3D(plans)
    rhs.batched = not pipeline  # small tiles: batched stages + stacked filter, or (forced) the fused stage kernels
    loop = StepLoop(Tvdrk3(rhs, pipeline=pipeline), ExpFilter3D(F, [metrics[p]["sqrtG"] for p in range(6)]),
                    flag, check_every=5)
    assert loop.fused == pipeline
    if pipeline:   # n = 8: the stage kernel and its fused filter are the matrix-core instantiation
        assert int(plans[0].lib.wx_euler3d_uses_matrix_cores(plans[0]._h, 1)) == int(n == 8)
    stack = lambda key: np.stack([g[f"p{p}/{key}"] for p in range(6)])  # noqa: E731
    Q0 = torch.from_numpy(stack("Q")).to(dev)
    dt, nsteps = float(g["meta/dt"]), int(g["meta/nsteps"])
    Q1 = loop.step(Q0, dt)
    Qn = loop.run(Q1, dt, nsteps - 1)
    ax = (0, 2, 3, 4, 5)
    for got, key in ((Q1, "Q1"), (Qn, "Qn")):
        ref = stack(key)
        moved = np.abs(ref - stack("Q")).max(axis=ax)
        err = np.abs(got.cpu().numpy() - ref).max(axis=ax)
        assert (err <= 1e-9 * moved + 1e-13 * np.abs(ref).max(axis=ax)).all(), (key, err, moved)
