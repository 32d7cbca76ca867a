"""Several REAL ranks on device buffers: 2 and 3 processes share the one GPU of the test box, each owning its run of cube
panels, halos travelling between the processes - through gloo and host copies (PanelExchange(backend="torch") on device
buffers, reduce.py's host-staged reductions), because RCCL refuses two ranks on one device.  Everything of the several-GPU
path except the transport itself runs as it would on N GPUs: tile ownership, the slot layout of the edge buffers, the
INTERIOR / BOUNDARY launches around a travelling exchange, the batched launches of a rank's tiles, the collective decision of
jvp_prepare, the split Krylov vectors of KIOPS with their two reductions per vector.  Checked against the REFERENCE's values:
R of every panel (1e-10), the KIOPS statistics of config/dcmip21.ini's EPI2 step decision for decision on every rank (config 5),
a Rosenbrock-2 + FGMRES step (config 4), SSP-RK3 steps through the stage pipeline, shallow water R(Q) and time loop (configs 2, 3).
(The transport: tests/test_exchange_rccl_gpu.py and tests/test_comm_gpu.py, one-rank communicator in loopback.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
AX = (0, 2, 3, 4, 5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests.gpu_util import make_plan, to_dev
        from tests.util import GOLDEN, golden, var_err, var_max, make_oracle
        from wxfactory_amd.exchange import PanelExchange
        from wxfactory_amd.matvec import matvec_fun
        from wxfactory_amd.panels import panels_of_rank
        from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

        # ---- (1) R(Q) and the complex-step matvec on the reference's DCMIP 3-1 fixture, every panel against the reference
        g = golden("euler3d_c31p_n3_h4_v2")
        mine = panels_of_rank(rank, world)
        plans = {p: make_plan(g, p) for p in mine}
        ex = PanelExchange(plans[mine[0]].edge_count, DEV, rank=rank, world_size=world)   # gloo, host-staged: device buffers
        assert ex.needs_comm and ex.local == mine
        rhs = RhsEuler3D(plans, ex, overlap=True)
        qs = {p: to_dev(g.q(p)) for p in mine}
        for form in ("dict", "stacked"):
            R = rhs(qs) if form == "dict" else dict(zip(mine, rhs(torch.stack([qs[p] for p in mine]))))
            torch.cuda.synchronize()
            for p in mine:
                ref = g.r(p)
                o = make_oracle(g, p)
                want = {}
                o.rhs(g.q(p), g.halo(p), want=want)
                scale = np.maximum(var_max(ref), o.cancel_scale(want))
                err = var_err(R[p].cpu().numpy(), ref)
                assert (err <= 1e-10 * scale).all(), (rank, form, p, err / scale)
        Q = torch.stack([qs[p] for p in mine])
        Rs = rhs(Q)
        v = to_dev(np.stack([g[f"p{p}/V"] for p in mine]))
        dtj = 1.0
        jv = matvec_fun(v.flatten(), dtj, Q, Rs, rhs, "complex").reshape(Q.shape)
        prepared = rhs.jvp_prepare(Q)        # (a collective decision: the same on every rank)
        flags = [None] * world
        dist.all_gather_object(flags, bool(prepared))
        assert len(set(flags)) == 1
        jv2 = matvec_fun(v.flatten(), dtj, Q, Rs, rhs, "complex").reshape(Q.shape)
        rhs.jvp_release()
        torch.cuda.synchronize()
        for i, p in enumerate(mine):
            ref = (g[f"p{p}/Rc"].imag / float(g["meta/eps"])).reshape(jv[i].shape)   # the reference's Im R(Q + i eps V) / eps
            for got in (jv[i], jv2[i]):
                e = np.abs(got.cpu().numpy() - ref).max(axis=(1, 2, 3, 4)) / np.abs(ref).max(axis=(1, 2, 3, 4))
                assert (e <= 1e-9).all(), (rank, p, e)

        # ---- (2) BASELINE config 5: dcmip21.ini's EPI2 + KIOPS step with the Krylov vectors split over the ranks - two
        # reductions per vector, every decision of the adaptive controller from all-reduced numbers: the reference's statistics
        from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch, planet_for_case, topography_for_case
        from wxfactory_amd.integrators import Epi
        from wxfactory_amd.synthetic import dfr_ops

        c = np.load(os.path.join(GOLDEN, "config5_c21_n4_h2_v3.npz"))
        n, H, V, case = (int(c[f"meta/{k}"]) for k in ("n", "H", "V", "case_number"))
        topo = topography_for_case(case, planet_for_case(case)[0])
        plans5 = {}
        for p in mine:
            t = CubedSphere3DTile(n, H, V, p, float(c["meta/ztop"]), case, topo=topo)
            plans5[p] = Euler3DPlan(n, H, V, case, p, dfr_ops(n), metric3d_torch(t, DEV))
        ex5 = PanelExchange(plans5[mine[0]].edge_count, DEV, rank=rank, world_size=world)
        rhs5 = RhsEuler3D(plans5, ex5, overlap=True)
        epi = Epi(2, rhs5, tol=float(c["meta/tolerance"]))
        assert epi.group is None    # (the default process group: gloo here; reduce.py stages device tensors through the host)
        stack = lambda key: np.stack([c[f"p{p}/{key}"] for p in mine])  # noqa: E731
        Q5 = torch.from_numpy(stack("Q")).to(DEV)
        Qn = epi.step(Q5, float(c["meta/dt"]))
        got = [int(epi.solver_info[k]) for k in ("substeps", "rejected", "iterations", "exps", "krylov_size")]
        ref_stats = c["meta/kiops_stats"][0]
        assert got == [int(ref_stats[j]) for j in (0, 1, 2, 3, 5)], (rank, epi.solver_info, ref_stats)
        ref_u = stack("Q1_unfiltered")
        upd = np.abs(ref_u - stack("Q")).max(axis=AX)
        # (the update is measured over the rank's own panels: compare with the largest update anywhere)
        updmax = torch.from_numpy(upd)
        dist.all_reduce(updmax, op=dist.ReduceOp.MAX)
        err = np.abs(Qn.cpu().numpy() - ref_u).max(axis=AX)
        assert (err <= 1e-6 * updmax.numpy() + 1e-13 * np.abs(ref_u).max(axis=AX)).all(), (rank, err / updmax.numpy())
        del epi, rhs5, ex5, plans5

        # ---- (3) BASELINE config 4: one Rosenbrock-2 step, FGMRES with the reference's one-synchronisation Gram-Schmidt
        # (integrators/ros2.py:24-81, solvers/fgmres.py): one reduction per Krylov vector over the ranks, every decision from
        # all-reduced numbers; and an SSP-RK3 step through the stage pipeline, whose INTERIOR / BOUNDARY launches each
        # extrapolate their own elements' output for the next stage
        from tests.gpu_util import device_metric
        from tests.util import Golden
        from wxfactory_amd.integrators import Ros2, Tvdrk3

        cg = Golden("callers_euler3d_n3_h3_v2")
        plans4 = {p: Euler3DPlan(cg.n, cg.H, cg.V, cg.case, p, cg.ops, device_metric(cg, p, DEV)) for p in mine}
        ex4 = PanelExchange(plans4[mine[0]].edge_count, DEV, rank=rank, world_size=world)
        rhs4 = RhsEuler3D(plans4, ex4, overlap=True)
        st4 = lambda key: torch.from_numpy(np.stack([cg[f"p{p}/{key}"] for p in mine])).to(DEV)  # noqa: E731

        def worst(t):   # the largest value of a per-variable array over all ranks
            t = torch.from_numpy(np.ascontiguousarray(t))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.numpy()

        ros = Ros2(rhs4, tol=1e-9, gmres_restart=30)
        Qr = ros.step(st4("Q"), float(cg["meta/dt_jvp"]))
        info = ros.solver_info
        assert info["flag"] == 0 and info["rel_residual"] < 1e-9 and abs(info["iterations"] - 151) <= 2, (rank, info["iterations"])
        ref, q0 = st4("ros2").cpu().numpy(), st4("Q").cpu().numpy()
        upd = worst(np.abs(ref - q0).max(axis=AX))
        err = np.abs(Qr.cpu().numpy() - ref).max(axis=AX)
        assert (err <= 1e-7 * upd).all(), (rank, err / upd)
        its = [None] * world
        dist.all_gather_object(its, int(info["iterations"]))
        assert len(set(its)) == 1
        stepper = Tvdrk3(rhs4)
        assert stepper.pipeline
        Qk = stepper.step(st4("Q"), float(cg["meta/dt_rk"]))
        ref = st4("rk3").cpu().numpy()
        upd = worst(np.abs(ref - q0).max(axis=AX))
        err = np.abs(Qk.cpu().numpy() - ref).max(axis=AX)
        assert (err <= 1e-9 * upd + 1e-13 * np.abs(ref).max(axis=AX)).all(), (rank, err / upd)
        del ros, stepper, rhs4, ex4, plans4

        # ---- (4) shallow water (BASELINE configs 2, 3): R(Q) on Williamson 5 (topography) against the reference, panel by panel,
        # and five SSP-RK3 steps of Williamson 6 at the order of config 3 against the reference's own run
        from tests.test_sw_gpu import _plan, _scale
        from tests.util import golden_sw
        from wxfactory_amd.rhs_sw import RhsShallowWater

        gs = golden_sw("sw_c5_n4_h3")
        plw = {p: _plan(gs, p) for p in mine}
        exw = PanelExchange(plw[mine[0]].edge_count, DEV, rank=rank, world_size=world)
        rw = RhsShallowWater(plw, exw)
        Rw = rw(torch.stack([to_dev(gs.q(p)) for p in mine]))
        torch.cuda.synchronize()
        for i, p in enumerate(mine):
            ref = gs.r(p)
            sc = np.maximum(var_max(ref), _scale(gs, p, False))
            assert (var_err(Rw[i].cpu().numpy(), ref) <= 1e-10 * sc).all(), (rank, p)
        gk = golden_sw("sw_rk3_c6_n8_h3")
        plk = {p: _plan(gk, p) for p in mine}
        exk = PanelExchange(plk[mine[0]].edge_count, DEV, rank=rank, world_size=world)
        stepper = Tvdrk3(RhsShallowWater(plk, exk))
        Qw = torch.stack([to_dev(gk.q(p)) for p in mine])
        dtk, nsteps = float(gk["meta/rk3_dt"]), int(gk["meta/rk3_steps"])
        for _ in range(nsteps):
            Qw = stepper.step(Qw, dtk)
        ref = np.stack([gk[f"p{p}/rk3_n"] for p in mine])
        moved = worst(np.abs(ref - np.stack([gk[f"p{p}/Q"] for p in mine])).max(axis=(0, 2, 3, 4)))
        err = np.abs(Qw.cpu().numpy() - ref).max(axis=(0, 2, 3, 4))
        assert (err <= 1e-8 * moved).all(), (rank, err / moved)

        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", got))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise


@pytest.mark.parametrize("world", [2, 3])
def test_real_ranks_on_device_buffers_reproduce_the_reference(built_lib, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=500) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]
    assert len({tuple(r[2]) for r in res}) == 1   # every rank took the same adaptive decisions


def _tile_worker(rank, world, port, q):
    try:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.euler3d import Euler3DOracle
        from tests.gpu_util import to_dev
        from tests.util import golden, var_err, var_max
        from wxfactory_amd.exchange import PanelExchange
        from wxfactory_amd.panels import tiles_of_rank
        from wxfactory_amd.rhs_euler3d import Euler3DPlan, RhsEuler3D

        g = golden("euler3d_tiles24_n3_h2_v2")   # the reference's own 24-rank run: one tile per rank there
        k = int(g["meta/k"])
        mine = tiles_of_rank(rank, world, 6 * k * k)
        ex = PanelExchange(5 * g.V * g.H * g.n**2, DEV, rank=rank, world_size=world, tiles_per_side=k)
        assert ex.local == mine and ex.needs_comm
        topo = ex.topo
        # (the fixture holds the reference's metric for two tiles only: every tile's metric from this package's own geometry,
        # which tests/test_geometry3d.py pins against the reference's)
        from wxfactory_amd.geometry3d import CubedSphere3DTile, metric3d_torch

        plans = {}
        for t in mine:
            panel, row, col = topo.locate(t)
            tile = CubedSphere3DTile(g.n, g.H, g.V, panel, 10000.0, g.case, row=row, col=col, k=k)
            plans[t] = Euler3DPlan(g.n, g.H, g.V, g.case, panel, g.ops, metric3d_torch(tile, DEV), on_panel_edge=topo.on_panel_edge(t))
        rhs = RhsEuler3D(plans, ex, overlap=True, tiles_per_side=k)
        assert rhs._small_tiles()                    # a rank's six tiles share one launch per phase
        Q = torch.stack([to_dev(g.q(t)) for t in mine])
        R = rhs(Q)
        torch.cuda.synchronize()
        # the scale of the terms that cancel in R, from a tile whose reference metric the fixture holds
        t0 = g.metric_panels()[0]
        o = Euler3DOracle(g.n, g.H, g.V, g.case, g.ops, g.metric(t0), g[f"p{t0}/geom/boundary_sn_new"],
                          g[f"p{t0}/geom/boundary_we_new"], panel=topo.locate(t0)[0], on_panel_edge=topo.on_panel_edge(t0))
        want = {}
        o.rhs(g.q(t0), g.halo(t0), want=want)
        cancel = o.cancel_scale(want)
        for i, t in enumerate(mine):
            ref = g.r(t)
            scale = np.maximum(var_max(ref), cancel)
            err = var_err(R[i].cpu().numpy(), ref)
            assert (err <= 1e-9 * scale).all(), (rank, t, err / scale)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", None))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise


def test_four_real_ranks_on_the_24_tile_layout(built_lib):
    """What 4 (and 8) GPUs run: the sphere cut into 24 tiles, six per rank, one launch per phase for a rank's tiles, halos between
    tiles of different ranks through the exchange, between tiles of one rank by aliasing - against the R the reference computed
    on 24 MPI ranks (1e-10)."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=500) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


@pytest.mark.parametrize("gpus", [2, 4])
def test_bench_program_flow_over_several_ranks(built_lib, gpus):
    """`python bench.py --gpus N` end to end with N real ranks (started by bench.py itself through torch.distributed.run), on
    ONE device (--one-device: every rank on GPU 0, halos through gloo and host copies): the gloo process group, the
    decomposition (whole panels at 2, the 24-tile layout at 4), the timed loop between barriers, the max over ranks, the
    all-reduced checksum, the gathered per-rank phase rows, ONE line on rank 0's stdout - everything of the several-GPU
    benchmark but the RCCL transport.  The checksum must equal the one-rank run's."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--H", "6", "--V", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=400, cwd=root)
    assert r1.returncode == 0, (r1.stdout[-1500:], r1.stderr[-3000:])
    one = json.loads([ln for ln in r1.stdout.splitlines() if ln.strip()][-1])
    rn = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--one-device", "--exchange", "torch"]
                        + common, capture_output=True, text=True, timeout=600, cwd=root)
    assert rn.returncode == 0, (rn.stdout[-1500:], rn.stderr[-4000:])
    out = [ln for ln in rn.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, out[:4]
    line = json.loads(out[0])
    assert line["n_gpus"] == gpus and line["ranks_seen"] == gpus and line["scaling"] == "strong"
    assert line["process_group"].startswith("gloo") and "rehearsal" in line
    assert len(line["per_rank"]) == gpus and sorted(r["rank"] for r in line["per_rank"]) == list(range(gpus))
    assert line["config"]["tiles"] == (6 if gpus == 2 else 24) and line["config"]["tiles_per_gpu"] == line["config"]["tiles"] // gpus
    assert all(r["tiles"] == line["config"]["tiles_per_gpu"] for r in line["per_rank"])
    for k in ("sum", "abs_sum", "max_abs"):
        a, b = np.asarray(line["checksum"][k]), np.asarray(one["checksum"][k])
        assert (np.abs(a - b) <= 1e-12 * np.maximum(np.asarray(one["checksum"]["abs_sum"]), 1e-300)).all(), (k, a, b)
