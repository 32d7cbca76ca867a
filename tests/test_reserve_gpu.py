"""No allocation after plan creation: the stage pipeline's second interface slot and
the prepared JVP's face-value cache are reserved at setup time (wx_euler3d_plan_reserve); without that the evaluation entry
points refuse, with it their FIRST call records into a HIP graph like any later one."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_evaluation_entry_points_refuse_without_reserve(built_lib):
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd import _lib

    g = golden("euler3d_c31p_n3_h4_v2")
    lib = _lib.load()
    plan = make_plan(g, 0)
    dual = plan.twin(torch.complex128, dual=True)
    q = to_dev(g.q(0))
    out = torch.empty_like(q)
    send = torch.zeros((4, plan.edge_count), dtype=torch.float64, device=DEV)
    ptrs = (ctypes.c_void_p * 4)(*[send[e].data_ptr() for e in range(4)])
    assert lib.wx_euler3d_plan_reserved(plan._h) == 0 and lib.wx_euler3d_plan_reserved(dual._h) == 0
    st = lib.wx_euler3d_stage(plan._h, q.data_ptr(), ptrs, None, None, out.data_ptr(), 0.0, 1.0, 1e-3, 0.0, _lib.WX_REGION_ALL, 0,
                              ptrs, 1, None, None)
    assert st == 1 and b"wx_euler3d_plan_reserve" in lib.wx_last_error()
    assert lib.wx_euler3d_extrap_pack_slot(plan._h, q.data_ptr(), ptrs, 1, None) == 1
    assert lib.wx_euler3d_jvp_prepare(dual._h, q.data_ptr(), ptrs, None) == 1 and b"WX_RESERVE_JVP" in lib.wx_last_error()
    assert lib.wx_euler3d_plan_reserve(plan._h, _lib.WX_RESERVE_JVP) == 1           # the cache belongs to dual plans
    assert lib.wx_euler3d_plan_reserve(plan._h, 8) == 1
    _lib.check(lib.wx_euler3d_plan_reserve(plan._h, _lib.WX_RESERVE_STAGE), "reserve")
    _lib.check(lib.wx_euler3d_plan_reserve(dual._h, _lib.WX_RESERVE_STAGE | _lib.WX_RESERVE_JVP), "reserve")
    assert lib.wx_euler3d_plan_reserved(plan._h) == 1 and lib.wx_euler3d_plan_reserved(dual._h) == 3
    assert lib.wx_euler3d_extrap_pack_slot(plan._h, q.data_ptr(), ptrs, 1, None) == 0
    assert lib.wx_euler3d_jvp_prepare(dual._h, q.data_ptr(), ptrs, None) == 0
    torch.cuda.synchronize()
    # the host wrapper says what to do when the first use falls inside a capture
    fresh = make_plan(g, 1)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            with pytest.raises(RuntimeError, match="reserve"):
                fresh.reserve(_lib.WX_RESERVE_STAGE)
    for p in (plan, dual, fresh):
        p.close()


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "own_n8_h3_v2"])
def test_first_calls_record_into_a_graph(built_lib, name):
    """The very first stage of the pipeline and the very first prepared matvec of an RHS object, captured."""
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    plans, Q, v = _sphere(name)
    plain = RhsEuler3D(plans)
    R = plain(Q)
    dt = 1e-3
    want_stage = plain.axpy(Q, None, 0.0, 1.0, dt)
    plain.batched = False
    want_mv = matvec_fun(v.flatten(), 1.0, Q, R, plain, "complex")

    rhs = RhsEuler3D(plans)
    rhs.batched = False
    rhs.reserve(stage=True, jvp=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g_stage, g_mv = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g_stage, stream=side):
            got_stage = rhs.stage(Q, None, 0.0, 1.0, dt)            # first call ever of wx_euler3d_stage on these plans
        assert rhs.jvp_prepare(Q)                                    # (takes a host decision: outside the capture)
        with torch.cuda.graph(g_mv, stream=side):
            got_mv = matvec_fun(v.flatten(), 1.0, Q, R, rhs, "complex")   # first prepared product
    torch.cuda.current_stream().wait_stream(side)
    for gr in (g_stage, g_mv):
        gr.replay()
    torch.cuda.synchronize()
    scale = want_stage.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    assert ((got_stage - want_stage).abs() <= 1e-14 * scale).all()
    assert torch.equal(got_mv, want_mv)
    rhs.jvp_release()


def _sphere(name):
    from tests.gpu_util import make_plan, to_dev

    if name.startswith("own"):
        from wxfactory_amd import synthetic
        from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
        from wxfactory_amd.initial import initial_state
        from wxfactory_amd.rhs_euler3d import Euler3DPlan

        n, H, V = 8, 3, 2
        plans, qs = {}, []
        gen = torch.Generator(device=DEV).manual_seed(4)
        for p in range(6):
            t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
            plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, DEV))
            q = torch.from_numpy(initial_state(t)).to(DEV)
            qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=DEV, dtype=q.dtype) - 0.5)))
        Q = torch.stack(qs)
        v = (torch.rand(Q.shape, generator=gen, device=DEV, dtype=Q.dtype) - 0.5) * 1e-3 * Q.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
        return plans, Q, v
    g = golden(name)
    plans = {p: make_plan(g, p) for p in range(6)}
    Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
    v = to_dev(np.stack([g[f"p{p}/V"] for p in range(6)]))
    return plans, Q, v
