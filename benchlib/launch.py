"""Start one rank per GPU for `python bench.py --gpus N` typed as is.

No torch in this process and no elastic agent between it and the ranks: a GPU box of this pool allows six processes on a card,
and `python -m torch.distributed.run --nproc-per-node 6` makes seven (profiles/r06_n6_rehearsal.md: the 6-rank rehearsal of
round 5 was killed by that guard, silently).  The ranks get what torch.distributed.run would set - RANK, LOCAL_RANK, WORLD_SIZE,
LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT - and bench.py reads them as it does under the driver's own torchrun."""
import os
import socket
import subprocess
import sys
import time


def spawn_ranks(n, argv, script):
    """Run `script argv` as n ranks; returns the first non-zero exit code (the other ranks are ended by their PIDs), else 0."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:   # one rank failed: the others would wait for it in a collective
                    q.terminate()
        time.sleep(0.05)
    return rc
