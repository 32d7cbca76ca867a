"""bench.py's tile -> rank map for the driver's N = 1, 2, 4, 6, 8 runs, against CubeTopology
(reference: process_topology.py:69-94 tiling, :259-261 delivery rule).  Host logic only."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
from wxfactory_amd.exchange import PanelExchange  # noqa: E402
from wxfactory_amd.panels import CubeTopology  # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
def test_every_tile_has_one_owner_and_the_exchange_tables_match(world):
    H = 60
    k, owner = bench.decomposition(world, H)
    topo = CubeTopology(k)
    assert len(owner) == topo.ntiles == 6 * k * k and H % k == 0
    assert k == (1 if world in (1, 2, 3, 6) else 2)
    # equal work per active rank, every rank of the driver's runs active
    counts = [owner.count(r) for r in range(world)]
    assert sum(counts) == topo.ntiles and min(counts) == max(counts) > 0
    ec = 7
    ex = [PanelExchange(ec, "cpu", rank=r, world_size=world, tiles_per_side=k) for r in range(world)]
    for r in range(world):
        # what r sends to s is what s expects from r (all_to_all_single split sizes)
        for s in range(world):
            assert ex[r].send_splits[s] == ex[s].recv_splits[r]
        for t in ex[r].local:
            assert owner[t] == r
            for e in range(4):
                q, e2 = topo.neighbor(t, e), topo.landing(t, e)
                assert topo.neighbor(q, e2) == t
                kind, _ = ex[r]._halo_src[(t, e)]
                # a halo aliases the sender's slot exactly when the neighbour lives on this rank
                assert (kind == "send") == (owner[q] == r)
    # message order inside a rank pair: sender's slot i is receiver's slot i (sorted by destination tile, edge)
    for r in range(world):
        for s in range(world):
            if r == s:
                continue
            sent = sorted((topo.neighbor(t, e), topo.landing(t, e)) for t in ex[r].local for e in range(4)
                          if owner[topo.neighbor(t, e)] == s)
            recv = sorted((q, e2) for q in ex[s].local for e2 in range(4) if owner[topo.neighbor(q, e2)] == r)
            assert sent == recv


def test_plain_invocation_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus 6` (no torchrun around it) must start six ranks itself: direct children with torchrun's
    environment, from a parent that has not imported torch (profiles/r06_n6_rehearsal.md: the box allows six processes on a
    card, an elastic agent would be the seventh)."""
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None):
            started.append((cmd, env))

        def poll(self):
            return 0

        def terminate(self):
            pass

    import subprocess

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "6", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert len(started) == 6
    ports = set()
    for r, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and os.path.basename(cmd[1]) == "bench.py" and cmd[2:] == ["--gpus", "6", "--steps", "3"]
        assert "torch.distributed.run" not in cmd
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "6"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1
    # the launcher itself pulls in neither torch nor the package
    src = open(os.path.join(ROOT, "benchlib", "launch.py")).read()
    assert "import torch" not in src and "wxfactory_amd" not in src
    top = open(os.path.join(ROOT, "bench.py")).read()
    assert "\nimport torch" not in top
