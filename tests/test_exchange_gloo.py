"""N > 1 path on CPU: the panel exchange over torch.distributed (gloo), world sizes 2, 3 and 6+2.

What is checked is the product's host logic (wxfactory_amd/exchange.py, panels.py): slot layout,
local aliasing, all_to_all split sizes and ordering.  The face payloads come from the CPU oracle
(pack = rotate + flip, oracle/euler3d.py) standing in for the HIP pack kernel, and the result must
equal what the reference's ExchangeRequest.wait() returned on every panel (golden q_itf_{s,n,w,e}).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import EDGE_FIELDS, golden, halo7, make_oracle

FIXTURE = "euler3d_c31p_n3_h4_v2"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.exchange import PanelExchange
        from wxfactory_amd.panels import panels_of_rank

        g = golden(FIXTURE)
        edge = EDGE_FIELDS * g.V * g.H * g.n**2
        ex = PanelExchange(edge, "cpu", rank=rank, world_size=world)
        assert ex.local == panels_of_rank(rank, world)
        for rep in range(2):  # twice: buffers are reused across RHS evaluations
            for p in ex.local:
                o = make_oracle(g, p)
                sends = o.pack_edges(o.extrapolate(g.q(p)))
                for e in range(4):
                    ex.send_view(p, e).copy_(torch.from_numpy(halo7(sends[e]).reshape(-1)))
            ex.start()
            ex.wait()
            for p in ex.local:
                for e in range(4):
                    got = ex.halo_view(p, e).numpy().reshape(EDGE_FIELDS, g.V, g.H, g.n**2)
                    ref = halo7(g.halo(p)[e])
                    err = np.abs(got - ref).max() / np.abs(ref).max()
                    assert err < 1e-13, (rank, p, e, err)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 3, 6, 8])
def test_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def test_single_rank_aliases_everything():
    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.panels import NEIGHBOR, landing_edge

    ex = PanelExchange(10, "cpu", rank=0, world_size=1)
    assert not ex.needs_comm and ex.local == list(range(6))
    for p in range(6):
        for e in range(4):
            ex.send_view(p, e).fill_(10 * p + e)
    for p in range(6):
        for e in range(4):
            q, e2 = NEIGHBOR[p][e], landing_edge(p, e)
            assert ex.halo_view(q, e2).data_ptr() == ex.send_view(p, e).data_ptr()  # zero copy


def test_panel_graph_matches_reference_delivery():
    """landing_edge reproduces the 'lands on neighbour's edge' table (from
    process_topology.py:105-113) and the ownership map leaves ranks >= 6 idle."""
    from wxfactory_amd.panels import landing_edge, owner_of_panels, panels_of_rank

    lands = [[landing_edge(p, e) for e in range(4)] for p in range(6)]
    N, S, W, E = 1, 0, 2, 3
    assert lands == [[N, S, E, W], [E, E, E, W], [S, N, E, W], [W, W, E, W], [N, N, N, N], [S, S, S, S]]
    assert owner_of_panels(1) == [0] * 6
    assert owner_of_panels(4) == [0, 1, 2, 3, 0, 1]  # 6 tiles do not split evenly over 4: round-robin
    assert owner_of_panels(8) == [0, 1, 2, 3, 4, 5]
    assert panels_of_rank(7, 8) == [] and panels_of_rank(1, 2) == [3, 4, 5]  # contiguous runs when even


class _CpuPlan:
    """Test double with the plan interface PanelRhs drives (extrap_pack / rhs / twin), computing with
    the CPU oracle: lets the orchestration (phase order, regions, idle ranks, stacked states) run under
    gloo without a GPU.  Not a fallback: it lives in tests/ only."""

    def __init__(self, g, p):
        self.g, self.p = g, p
        self.o = make_oracle(g, p)
        self.dtype, self.device = torch.float64, torch.device("cpu")
        self.shape = (5, g.V, g.H, g.H, g.n**3)
        self.edge_count = EDGE_FIELDS * g.V * g.H * g.n**2
        self.calls = []

    def extrap_pack(self, q, send):
        self.itf = self.o.extrapolate(q.numpy())
        for e, s in enumerate(self.o.pack_edges(self.itf)):
            send[e].copy_(torch.from_numpy(halo7(s).reshape(-1)))
        self.calls.append("pack")

    def rhs(self, q, halo, out, region=0):
        self.calls.append(("rhs", region))
        if region == 1:  # INTERIOR: nothing to check without halos; BOUNDARY/ALL computes everything
            return
        h5 = [h.numpy().reshape(EDGE_FIELDS, self.g.V, self.g.H, self.g.n**2)[:5] for h in halo]
        out.copy_(torch.from_numpy(self.o.rhs(q.numpy(), h5, itf=self.itf)))


class _CpuJvpPlan(_CpuPlan):
    """... plus the prepared complex-step JVP interface (jvp_prepare / jvp_tangent_pack / jvp_prepared) and twin(), so that
    RhsEuler3D.jvp_prepare / jvp can be driven over gloo: value halos exchanged once, tangent halos per product."""

    def twin(self, dtype, dual=False):
        return self

    def jvp_prepare(self, q, send_val):
        self.itf_val = self.o.extrapolate(q.numpy())
        for e, s_ in enumerate(self.o.pack_edges(self.itf_val)):
            send_val[e].copy_(torch.from_numpy(np.ascontiguousarray(s_).reshape(-1)))
        self.calls.append("prepare")

    def jvp_tangent_pack(self, q, v, eps, send_tan):
        self.itf_c = self.o.extrapolate(q.numpy() + 1j * eps * v.numpy())
        for d in range(3):   # the cached values are what the product uses for the real parts
            assert np.abs(self.itf_c[d].real - self.itf_val[d]).max() <= 1e-13 * np.abs(self.itf_val[d]).max()
        for e, s_ in enumerate(self.o.pack_edges(self.itf_c)):
            send_tan[e].copy_(torch.from_numpy(np.ascontiguousarray(s_.imag).reshape(-1)))
        self.calls.append("tangent_pack")

    def jvp_prepared(self, q, v, eps, halo_val, halo_tan, out, scale, region=0):
        self.calls.append(("jvp", region))
        if region == 1:
            return
        shp = (5, self.g.V, self.g.H, self.g.n**2)
        halos = [hv.numpy().reshape(shp) + 1j * ht.numpy().reshape(shp) for hv, ht in zip(halo_val, halo_tan)]
        R = self.o.rhs(q.numpy() + 1j * eps * v.numpy(), halos, itf=self.itf_c)
        out.copy_(torch.from_numpy(scale * R.imag))


def _jvp_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.panels import panels_of_rank
        from wxfactory_amd.rhs_euler3d import RhsEuler3D

        g = golden("euler3d_c31p_n3_h4_v2")
        mine = panels_of_rank(rank, world)
        plans = {p: _CpuJvpPlan(g, p) for p in mine}
        rhs = RhsEuler3D(plans, rank=rank, world_size=world, device="cpu", edge_count=5 * g.V * g.H * g.n**2)
        rhs.batched = False
        shp = (len(mine), 5, g.V, g.H, g.H, g.n**3)
        Q = torch.from_numpy(np.stack([g.q(p) for p in mine])) if mine else torch.zeros((0,) + shp[1:], dtype=torch.float64)
        V = torch.from_numpy(np.stack([g[f"p{p}/V"] for p in mine])) if mine else torch.zeros_like(Q)
        assert rhs.jvp_prepare(Q)   # collective: value faces exchanged once
        for rep in range(2):        # two products on the prepared state: tangents only
            out = rhs.jvp(Q, V, g.eps, 1.0 / g.eps)
            for i, p in enumerate(mine):
                ref = g.r(p, True).imag / g.eps
                err = np.abs(out[i].numpy() - ref).max(axis=(1, 2, 3, 4)) / np.abs(ref).max(axis=(1, 2, 3, 4))
                assert (err < 1e-9).all(), (rank, p, rep, err)
        for p in mine:
            assert plans[p].calls == ["prepare"] + ["tangent_pack", ("jvp", 1), ("jvp", 2)] * 2, plans[p].calls
        rhs.jvp_release()
        # the production path: Epi -> kiops -> ComplexStepOperator -> matvec_fun, on every rank, idle ones included
        # (they own no tile, take part in every exchange, and must choose the same exchange as the others)
        from wxfactory_amd.matvec import ComplexStepOperator, matvec_fun

        R = torch.zeros_like(Q)
        op = ComplexStepOperator(2.0, Q, R, rhs)          # prepares (collective)
        assert rhs._jvp_lin is not None
        for scale in (1.0, -0.5):
            out = op((scale * V).flatten()).reshape(Q.shape)
            for i, p in enumerate(mine):
                ref = scale * 2.0 * g.r(p, True).imag / g.eps
                assert np.abs(out[i].numpy() - ref).max() <= 1e-9 * np.abs(ref).max(), (rank, p)
        # a declared linearisation state is binding across ranks: another state is an error, not a silent fall-back
        if mine:
            with pytest.raises(RuntimeError, match="jvp_release"):
                matvec_fun(V.flatten(), 2.0, Q.clone(), R, rhs, "complex")
        rhs.jvp_release()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 6, 8])
def test_prepared_jvp_orchestration_over_gloo(world):
    """RhsEuler3D.jvp_prepare + jvp across ranks (whole panels on 2 ranks; 8 ranks, two of them idle): the value halos
    travel once, the tangent halos once per product through their own exchange, interior before boundary, and the
    product equals the reference's complex-step R (CPU test double of the plan, gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_jvp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def _rhs_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.panel_rhs import PanelRhs
        from wxfactory_amd.panels import panels_of_rank

        g = golden(FIXTURE)
        mine = panels_of_rank(rank, world)
        plans = {p: _CpuPlan(g, p) for p in mine}
        rhs = PanelRhs(plans, rank=rank, world_size=world, device="cpu", edge_count=EDGE_FIELDS * g.V * g.H * g.n**2)
        qs = {p: torch.from_numpy(g.q(p).copy()) for p in mine}
        out = rhs(qs)
        for p in mine:
            ref = g.r(p)
            err = np.abs(out[p].numpy() - ref).max(axis=(1, 2, 3, 4)) / np.abs(ref).max(axis=(1, 2, 3, 4))
            assert (err < 1e-10).all(), (rank, p, err)
            assert plans[p].calls == ["pack", ("rhs", 1), ("rhs", 2)]  # overlap ordering of rhs.py:88-118
        if mine:  # stacked-state form
            st = torch.stack([qs[p] for p in mine])
            o2 = rhs(st)
            assert o2.shape == st.shape and all(torch.equal(o2[i], out[p]) for i, p in enumerate(mine))
        else:
            rhs({})  # idle rank takes part in the second collective too
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 6, 8])
def test_panel_rhs_orchestration_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rhs_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def _state_worker(rank, world, port, q, k):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.panels import CubeTopology, tiles_of_rank
        from wxfactory_amd.state import distribute_cube, gather_cube

        Ht, V, n3 = 2, 3, 4
        glob = torch.arange(6 * 5 * V * (k * Ht) * (k * Ht) * n3, dtype=torch.float64).reshape(6, 5, V, k * Ht, k * Ht, n3)
        local = distribute_cube(glob if rank == 0 else None, rank, world, tiles_per_side=k, tile_shape=(5, V, Ht, Ht, n3))
        topo = CubeTopology(k)
        mine = tiles_of_rank(rank, world, topo.ntiles)
        assert local.shape == (len(mine), 5, V, Ht, Ht, n3)
        for i, t in enumerate(mine):   # tile (panel, row, col) = rows along axis -3, columns along axis -2
            p, r, c = topo.locate(t)
            assert torch.equal(local[i], glob[p, :, :, r * Ht:(r + 1) * Ht, c * Ht:(c + 1) * Ht, :])
        back = gather_cube(local, rank, world, tiles_per_side=k)
        assert (back is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(back, glob)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,k", [(2, 1), (8, 1), (4, 2), (8, 2), (6, 3)])
def test_checkpoint_layout_is_rank_count_independent(world, k):
    """gather_cube / distribute_cube (process_topology.py:444-539) for whole panels and for the 24-tile layout the
    benchmark uses on 4 and 8 GPUs; only rank 0 holds the global array (tensor gather / scatter)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_state_worker, args=(r, world, port, q, k)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


# ------------------------------------------------------------------------------------------------
# 6 k^2 tiles (k = 2): the reference's own finer decomposition, pinned by a 24-rank run of the reference
# ------------------------------------------------------------------------------------------------
TILES = "euler3d_tiles24_n3_h2_v2"


def test_tile_topology_matches_reference_process_topology():
    from wxfactory_amd.panels import CubeTopology, tiles_per_side_for

    g = golden(TILES)
    T = CubeTopology(int(g["meta/k"]))
    for t in range(T.ntiles):
        assert [T.neighbor(t, e) for e in range(4)] == list(g[f"p{t}/topo/neighbors"])
        assert T.flips(t) == [bool(x) for x in g[f"p{t}/topo/flip"]]
        assert T.locate(t) == tuple(int(x) for x in g[f"p{t}/topo/panel_row_col"])
    assert [tiles_per_side_for(n) for n in (1, 2, 3, 4, 6, 8, 24)] == [1, 1, 1, 2, 1, 2, 2]


def _tile_oracle(g, t, topo):
    from oracle.euler3d import Euler3DOracle

    return Euler3DOracle(g.n, g.H, g.V, g.case, g.ops, g.metric(t), g[f"p{t}/geom/boundary_sn_new"],
                         g[f"p{t}/geom/boundary_we_new"], panel=topo.locate(t)[0], on_panel_edge=topo.on_panel_edge(t))


def _tile_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from wxfactory_amd.exchange import PanelExchange
        from wxfactory_amd.panels import tiles_of_rank

        g = golden(TILES)
        k = int(g["meta/k"])
        ex = PanelExchange(EDGE_FIELDS * g.V * g.H * g.n**2, "cpu", rank=rank, world_size=world, tiles_per_side=k)
        assert ex.local == tiles_of_rank(rank, world, 6 * k * k)
        for t in ex.local:
            o = _tile_oracle(g, t, ex.topo)
            for e, s in enumerate(o.pack_edges(o.extrapolate(g.q(t)))):
                ex.send_view(t, e).copy_(torch.from_numpy(halo7(s).reshape(-1)))
        ex.start()
        ex.wait()
        for t in ex.local:
            for e in range(4):
                got = ex.halo_view(t, e).numpy().reshape(EDGE_FIELDS, g.V, g.H, g.n**2)
                ref = halo7(g.halo(t)[e])
                assert np.abs(got - ref).max() < 1e-13 * np.abs(ref).max(), (rank, t, e)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [1, 4, 8])
def test_tile_exchange_over_gloo(world):
    """24 tiles over 1, 4 and 8 ranks (6, 3 tiles each): pack (rotation/flip only on panel edges) + routing
    == the halos the reference delivered on 24 MPI ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def test_inline_exchange_is_capture_safe_by_construction(monkeypatch):
    """The stream-ordered ("inline") form of the halo exchange a HIP-graph capture needs (BASELINE config 5 on several
    GPUs): the collective is issued with async_op=False on buffers that never move, no work handle is kept, the views
    the kernels were given keep their addresses, and an RHS object falls back from INTERIOR / BOUNDARY to one launch.
    No process group: the collective itself is replaced by a recorder."""
    import torch.distributed as dist

    from wxfactory_amd.exchange import PanelExchange
    from wxfactory_amd.panel_rhs import PanelRhs

    calls = []

    def fake_all_to_all_single(recv, send, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        calls.append((recv.data_ptr(), recv.numel(), send.data_ptr(), send.numel(), tuple(output_split_sizes),
                      tuple(input_split_sizes), async_op))
        return object() if async_op else None

    monkeypatch.setattr(dist, "all_to_all_single", fake_all_to_all_single)
    for world, rank, k in ((2, 1, 1), (6, 4, 1), (8, 3, 2), (1, 0, 1)):
        mode = {"inline": False}
        ex = PanelExchange(40, "cpu", rank=rank, world_size=world, loopback=(world == 1), tiles_per_side=k, mode=mode)
        assert ex.needs_comm
        ptrs = {(p, e): (ex.send_view(p, e).data_ptr(), ex.halo_view(p, e).data_ptr()) for p in ex.local for e in range(4)}
        del calls[:]
        ex.start()
        assert ex._work is not None and calls[-1][-1] is True      # the overlapping form keeps a handle
        ex._work = None
        mode["inline"] = True
        assert ex.is_inline
        for _ in range(3):
            ex.start()
            assert ex._work is None and calls[-1][-1] is False     # stream-ordered, nothing kept
            ex.wait()                                              # nothing to wait for
        assert len({c[:6] for c in calls}) == 1                    # same buffers, same splits, every time
        assert calls[-1][1] == ex.n_remote_in * 40 and calls[-1][3] == ex.n_remote_out * 40
        assert ptrs == {(p, e): (ex.send_view(p, e).data_ptr(), ex.halo_view(p, e).data_ptr())
                        for p in ex.local for e in range(4)}

    # an RHS object: the shared switch reaches every exchange it owns or creates, and the phase order becomes
    # pack -> exchange -> ALL (recorded through a test double of the plan)
    order = []

    class Plan:
        dtype, shape, device, edge_count = torch.float64, (5, 1, 2, 2, 8), torch.device("cpu"), 40

        def __init__(self, p):
            self.p = p

        def extrap_pack(self, q, send):
            order.append(("pack", self.p))

        def rhs(self, q, halo, out, region):
            order.append(("rhs", self.p, region, halo is not None))
            out.zero_()

        def twin(self, dtype, dual=False):
            return self

    ex = PanelExchange(40, "cpu", rank=0, world_size=2)
    rhs = PanelRhs({p: Plan(p) for p in ex.local}, ex)
    assert ex.mode is rhs.comm_mode
    q = {p: torch.zeros(Plan.shape, dtype=torch.float64) for p in ex.local}
    del calls[:]
    monkeypatch.setattr(ex, "wait", lambda: order.append(("wait",)))
    rhs(q)
    assert [o for o in order if o[0] == "rhs"][0][2] == 1 and calls[-1][-1] is True    # INTERIOR first, async collective
    del order[:]
    rhs.set_inline_exchange(True)
    rhs(q)
    assert calls[-1][-1] is False
    kinds = [o[0] for o in order]
    assert kinds == ["pack"] * len(ex.local) + ["wait"] + ["rhs"] * len(ex.local)
    assert all(o[2] == 0 and o[3] for o in order if o[0] == "rhs")                     # region ALL, halos given
    assert rhs.exchange_for(torch.complex128).is_inline                                # a later-created exchange too


# ---------------------------------------------------------------------------------------------------------
# The bootstrap of the library's communicator and the callers' reductions WITHOUT an NCCL process group
# ---------------------------------------------------------------------------------------------------------
def _id_worker(rank, world, port, store_port, q):
    try:
        from wxfactory_amd import reduce
        from wxfactory_amd.exchange import share_comm_id

        mine = bytes([rank + 1]) * 128          # (what wx_comm_unique_id would have produced on this rank)
        # (a) no process group at all: a TCPStore carries rank 0's id
        store = dist.TCPStore("127.0.0.1", store_port, world, is_master=(rank == 0), timeout=__import__("datetime").timedelta(seconds=60))
        assert not dist.is_initialized() and reduce.world_size(None) == 1
        got = share_comm_id(mine, rank, world, store=store)
        assert got == bytes([1]) * 128 and len(got) == 128, rank
        # (b) a gloo group: broadcast; then the reductions of reduce.py over the same group (the CPU twin of RcclComm.allreduce)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert share_comm_id(mine, rank, world) == bytes([1]) * 128
        assert reduce.world_size(None) == world and not reduce.is_comm(None)
        assert reduce.capturable(None) == (world == 1)
        t = torch.tensor([float(rank + 1), -float(rank)], dtype=torch.float64)
        assert reduce.allreduce(t.clone(), None, "sum").tolist() == [world * (world + 1) / 2, -world * (world - 1) / 2]
        assert reduce.allreduce(t.clone(), None, "max").tolist() == [float(world), 0.0]
        assert reduce.allreduce(t.clone(), None, "min").tolist() == [1.0, -float(world - 1)]

        class Comm:     # the duck type reduce.py accepts for the library's communicator
            calls = 0

            def allreduce(self, x, op):
                Comm.calls += 1
                dist.all_reduce(x, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
                return x

        c = Comm()
        c.world = world
        assert reduce.is_comm(c) and reduce.world_size(c) == world and reduce.capturable(c)
        assert reduce.allreduce(t.clone(), c, "sum")[0] == world * (world + 1) / 2 and Comm.calls == 1
        from wxfactory_amd.filters import NanFlag

        flag = NanFlag("cpu", group=c)
        flag.flag[0] = 1 if rank == world - 1 else 0
        try:
            flag.raise_if_set()
            raised = False
        except ValueError:
            raised = True
        assert raised and Comm.calls == 2      # every rank raises (simulation.py:399-408), through the communicator
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 3])
def test_id_bootstrap_over_a_store(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, store_port = _free_port(), _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, store_port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def test_device_buffers_never_travel_on_a_torch_nccl_group(monkeypatch):
    """backend "torch" with device buffers: gloo = staged through host copies (bench.py's independent route), anything else
    is refused - the data path of several GPUs is the library's own communicator."""
    import inspect

    from wxfactory_amd import exchange

    src = inspect.getsource(exchange.PanelExchange.start)
    assert 'dist.get_backend(self.group) != "gloo"' in src and "backend='rccl'" in src
    import re

    pkg = os.path.dirname(exchange.__file__)
    root = os.path.dirname(pkg)
    for name in ["bench.py"] + [os.path.join("benchlib", f) for f in os.listdir(os.path.join(root, "benchlib")) if f.endswith(".py")]:
        assert not re.search(r"init_process_group\(\s*[\"']nccl", open(os.path.join(root, name)).read()), name
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            assert not re.search(r"init_process_group\(\s*[\"']nccl", open(os.path.join(pkg, name)).read()), name
