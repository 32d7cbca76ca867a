"""The halo exchange behind the C ABI (csrc/exchange.hip: wx_comm_*, wx_exchange_*, wx_euler3d_rhs_overlapped) on one GPU:
a one-rank RCCL communicator of the library's own - no torch.distributed anywhere in this file - with the exchange in
loopback mode, so that every edge message travels through ncclSend / ncclRecv on the communication stream while the INTERIOR
launch runs (reference: process_topology.py:269-386, 564-606 inside rhs/rhs.py:88-118).  Results must equal the aliasing
path (halos read straight from the neighbours' send slots) bit for bit, eagerly and replayed from a HIP graph that was
captured WITH the fork / join in it."""
import numpy as np
import pytest
import torch

from tests.util import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def comm():
    from wxfactory_amd.exchange import RcclComm

    c = RcclComm(0, 1, device=DEV)
    assert c.version >= 20000
    yield c
    torch.cuda.synchronize()
    c.close()


def _setup(name, comm):
    from tests.gpu_util import make_plan, to_dev
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

    if name.startswith("own"):
        # the matrix-core kernels (n = 8) on this package's own geometry and initial state: 3 x 3 x 2 elements per panel,
        # so that the INTERIOR launch has an element to work on (the reference's n = 8 fixtures hold two panels only)
        from wxfactory_amd import synthetic
        from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch
        from wxfactory_amd.initial import initial_state

        n, H, V = 8, 3, 2
        g, plans, qs = None, {}, []
        gen = torch.Generator(device=DEV).manual_seed(4)
        for p in range(6):
            t = CubedSphere3DTile(n, H, V, p, 10000.0, 31)
            plans[p] = Euler3DPlan(n, H, V, 31, p, synthetic.dfr_ops(n), metric3d_torch(t, DEV))
            q = torch.from_numpy(initial_state(t)).to(DEV)
            qs.append(q * (1.0 + 0.01 * (torch.rand(q.shape, generator=gen, device=DEV, dtype=q.dtype) - 0.5)))
        Q = torch.stack(qs)
    else:
        g = golden(name)
        plans = {p: make_plan(g, p) for p in range(6)}
        Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
    assert ex.needs_comm and ex.n_remote_out == 24 and ex.n_remote_in == 24 and ex._native is not None
    return g, plans, ex, Q, RhsEuler3D(plans)


@pytest.mark.parametrize("name", ["euler3d_c31p_n3_h4_v2", "own_n8_h3_v2"])
def test_native_exchange_equals_the_aliasing_path(comm, name):
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g, plans, ex, Q, plain = _setup(name, comm)
    want = plain(Q)
    assert torch.isfinite(want).all()
    for batched in (False, True):   # per-tile launches through ONE call of the C ABI; one launch per phase for all tiles
        rhs = RhsEuler3D(plans, ex, overlap=True)
        rhs.batched = batched
        got = rhs(Q)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (name, batched)
        got = rhs({p: Q[p] for p in range(6)})      # the dictionary form goes through wx_euler3d_rhs_overlapped too
        assert all(torch.equal(got[p], want[p]) for p in range(6))
    # ... and the pieces driven from the host one by one (what PanelRhs does for stage updates, JVPs, timed evaluations)
    rhs = RhsEuler3D(plans, ex, overlap=True)
    rhs.batched = False
    rhs.timed = True
    got = rhs(Q)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    rhs.retrieve_last_times()
    assert len(rhs.timings[-1]) == 9 and rhs.timings[-1][8] > 0.0
    # the messages did travel: every halo of the loopback exchange sits in the receive buffer and holds the neighbour's slot
    for p in range(6):
        for e in range(4):
            q, e2 = ex.topo.neighbor(p, e), ex.topo.landing(p, e)
            assert ex._halo_src[(p, e)][0] == "recv"
            assert torch.equal(ex.halo_view(p, e), ex.send_view(q, e2))
    if g is not None:
        ref = np.stack([g.r(p) for p in range(6)])
        scale = np.abs(ref).max(axis=(0, 2, 3, 4, 5), keepdims=True)
        assert (np.abs(want.cpu().numpy() - ref) <= 1e-9 * scale).all()   # (the tight comparison lives in test_euler3d_gpu.py)


def test_overlapped_exchange_records_into_a_graph(comm):
    """BASELINE config 5 ("hipGraph-captured matvec" over several GPUs) with the OVERLAP kept: R(Q) and the prepared
    complex-step matvec captured while the exchange is forked to its communication stream beside the INTERIOR launch,
    replayed bit-identically.  (torch's all_to_all_single with a deferred wait aborts in hipStreamEndCapture on this stack;
    the event fork / join of csrc/exchange.hip is the shape that records.)"""
    from tests.gpu_util import to_dev
    from wxfactory_amd.graph import GraphedFunction
    from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g, plans, ex, Q, plain = _setup("euler3d_c31p_n3_h4_v2", comm)
    R = plain(Q)
    v = to_dev(np.stack([g[f"p{p}/V"] for p in range(6)]))
    for batched in (False, True):
        gr = RhsEuler3D(plans, ex, overlap=True)
        gr.batched = batched
        g_rhs = GraphedFunction(gr, Q)            # no rhs=: the exchange keeps its forked form inside the capture
        assert not gr.ex.is_inline and gr.ex.needs_comm
        for scale in (1.0, 1.01):
            assert torch.equal(g_rhs(Q * scale), plain(Q * scale)), (batched, scale)
        op = ComplexStepOperator(1.0, Q, R, gr)
        assert gr._jvp_is_prepared(Q) == (not batched)
        g_mv = GraphedFunction(op, v.flatten())
        for scale in (1.0, -0.37):
            assert torch.equal(g_mv((scale * v).flatten()), matvec_fun((scale * v).flatten(), 1.0, Q, R, plain, "complex"))
        gr.jvp_release()
        del g_rhs, g_mv, op, gr
        torch.cuda.synchronize()


def test_stage_pipeline_and_jvp_over_the_native_exchange(comm):
    from tests.gpu_util import to_dev
    from wxfactory_amd.matvec import matvec_fun
    from wxfactory_amd.rhs_euler3d import RhsEuler3D

    g, plans, ex, Q, plain = _setup("euler3d_c31p_n3_h4_v2", comm)
    dt = 1e-3
    piped = RhsEuler3D(plans, ex, overlap=True)
    Q1 = piped.stage(Q, None, 0.0, 1.0, dt)
    Q2 = piped.stage(Q1, Q, 0.75, 0.25, 0.25 * dt)
    Q3 = piped.stage(Q2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
    P1 = plain.axpy(Q, None, 0.0, 1.0, dt)
    P2 = plain.axpy(P1, Q, 0.75, 0.25, 0.25 * dt)
    P3 = plain.axpy(P2, Q, 1.0 / 3.0, 2.0 / 3.0, (2.0 / 3.0) * dt)
    torch.cuda.synchronize()
    scale = P3.abs().amax(dim=(0, 2, 3, 4, 5), keepdim=True)
    assert ((Q3 - P3).abs() <= 1e-14 * scale).all()
    R = plain(Q)
    v = to_dev(np.stack([g[f"p{p}/V"] for p in range(6)]))
    for r_ in (piped, plain):
        r_.batched = False
    j_plain = matvec_fun(v.flatten(), 1.0, Q, R, plain, "complex")
    assert piped.jvp_prepare(Q)
    j_coll = matvec_fun(v.flatten(), 1.0, Q, R, piped, "complex")
    piped.jvp_release()
    torch.cuda.synchronize()
    assert torch.equal(j_plain, j_coll)
    assert piped._ex_tan.backend == "rccl" and piped._ex_tan._native is not None
    # the store that forms a * (A v) + b * z in the product itself (KIOPS' next Krylov vector), over the INTERIOR / BOUNDARY
    # launches of the overlapped evaluation: the same bits as from one launch per tile, coefficients read from device memory
    from wxfactory_amd.matvec import ComplexStepOperator

    coef = torch.tensor([0.75, -1.5], dtype=torch.float64, device=DEV)
    z = torch.randn(Q.numel(), dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    outs = []
    for r_ in (plain, piped):
        op = ComplexStepOperator(1.0, Q, R, r_)
        assert r_.jvp_fuses_store(Q)
        out = torch.full_like(z, float("nan"))
        assert op.axpy_into(v.flatten().contiguous(), out, z, coef[0:1].data_ptr(), coef[1:2].data_ptr())
        outs.append(out)
        r_.jvp_release()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    want = 0.75 * j_plain - 1.5 * z
    assert float((outs[0] - want).abs().max()) <= 1e-14 * float(want.abs().max())
    # ... and the products of the stored vector with two other vectors, as pairs of partial sums per workgroup
    rows = [torch.randn(Q.numel(), dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(s_))
            for s_ in (6, 7)]
    for r_ in (plain, piped):
        for nrows in (1, 2):
            op = ComplexStepOperator(1.0, Q, R, r_)
            out = torch.full_like(z, float("nan"))
            part, count = op.axpy_into(v.flatten().contiguous(), out, z, coef[0:1].data_ptr(), coef[1:2].data_ptr(), rows[:nrows])
            torch.cuda.synchronize()
            assert torch.equal(out, outs[0]) and 0 < 2 * count <= part.numel()
            got = part[: 2 * count].view(count, 2).sum(dim=0)
            for k in range(nrows):
                ref = float(torch.dot(rows[k], out))
                assert abs(float(got[k]) - ref) <= 1e-12 * float(rows[k].norm() * out.norm()), (nrows, k)
            if nrows == 1:
                assert float(got[1]) == 0.0
            r_.jvp_release()


def test_shallow_water_over_the_native_exchange(comm):
    from tests.gpu_util import to_dev
    from tests.test_sw_gpu import _plan
    from tests.util import golden_sw
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.rhs_sw import RhsShallowWater

    g = golden_sw("sw_c5_n4_h3")   # Williamson 5: topography
    plans = {p: _plan(g, p) for p in range(6)}
    ex = PanelExchange(plans[0].edge_count, DEV, rank=0, world_size=1, loopback=True, backend="rccl", comm=comm)
    Q = torch.stack([to_dev(g.q(p)) for p in range(6)])
    want = RhsShallowWater(plans)(Q)
    for batched in (False, True):
        rhs = RhsShallowWater(plans, ex, overlap=True)
        rhs.batched = batched
        got = rhs(Q)
        torch.cuda.synchronize()
        assert torch.equal(got, want), batched


@pytest.mark.parametrize("form", ["forkjoin", "inline"])
def test_plain_c_capture_probe(built_lib, tmp_path, form):
    """tools/rccl_capture_probe.c - plain C on the image's ROCm (HIP 7.2, RCCL 2.27), no torch in the process: the
    library's wx_exchange_start / _wait with the exchange on its own communication stream, captured into a HIP graph and
    replayed with fresh data three times, every halo checked.  (The same program on the HIP 7.0.2 runtime bundled with
    torch dies in hipStreamEndCapture: profiles/r04_capture_crash.md.)"""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc, rocm = shutil.which("gcc"), os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no C toolchain / HIP headers on this box")
    exe, libdir = tmp_path / "probe", os.path.dirname(built_lib)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{root}/include",
                    os.path.join(root, "tools", "rccl_capture_probe.c"), f"-L{libdir}", "-lwxhip", f"-L{rocm}/lib", "-lamdhip64",
                    "-o", str(exe)], check=True, capture_output=True, text=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([libdir, f"{rocm}/lib"]))
    r = subprocess.run([str(exe)] + (["inline"] if form == "inline" else []) + ["multi"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and "PASS" in r.stdout and r.stdout.count("halos correct") == 4, (r.stdout[-1500:], r.stderr[-500:])
