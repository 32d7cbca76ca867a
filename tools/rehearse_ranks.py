#!/usr/bin/env python3
"""The north star's decomposition - six ranks, one cube panel each - through the whole program on the ONE-GPU box.

    python tools/rehearse_ranks.py [--world 6] [--out gpurun_out/rehearsal.json]

The box allows six processes on its card (profiles/r06_n6_rehearsal.md), and `pytest -m gpu` is one of them: a 6-rank run
cannot start from inside the test suite.  This program imports nothing that opens the GPU and starts
  (1) tests/test_multirank_gpu.py's worker as `world` REAL ranks on device buffers (R(Q) and the complex-step product of every
      panel against the reference's fixtures, dcmip21.ini's EPI2 + KIOPS step with the reference's statistics, a Rosenbrock-2 +
      FGMRES step, SSP-RK3 steps, shallow water) - what test_real_ranks_on_device_buffers_reproduce_the_reference runs at 2 and 3;
  (2) `bench.py --gpus world --one-device --exchange torch` at the small size of test_bench_program_flow_over_several_ranks, and
      the same at one rank: the two lines' checksums must agree to 1e-12.
Its report (JSON) is kept under profiles/.  Halos travel through gloo and host copies: RCCL refuses two ranks on one device -
the transport itself is what only the driver's multi-GPU run exercises."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_MAIN = """
import sys, json, queue
sys.path.insert(0, {root!r})
from tests.test_multirank_gpu import _worker
class Q:
    def put(self, item):
        json.dump([item[0], item[1], item[2]], open({out!r}.format(item[0]), "w"))
_worker({rank}, {world}, {port}, Q())
"""


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def real_ranks(world, scratch):
    port = free_port()
    out = os.path.join(scratch, "rank{}.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", RANK_MAIN.format(root=ROOT, out=out, rank=r, world=world, port=port)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    res = []
    for r in range(world):
        try:
            res.append(json.load(open(out.format(r))))
        except OSError:
            res.append([r, "no result file; exit code %s; tail: %s" % (procs[r].returncode, logs[r][-1500:]), None])
    ok = all(x[1] == "ok" for x in res) and all(p.returncode == 0 for p in procs)
    same = len({json.dumps(x[2]) for x in res}) == 1
    return {"world": world, "ok": ok, "every_rank_took_the_same_adaptive_decisions": same, "seconds": round(time.time() - t0, 1),
            "kiops_statistics": res[0][2], "failures": [x for x in res if x[1] != "ok"]}


def bench_flow(world):
    common = ["--H", "6", "--V", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    bench = os.path.join(ROOT, "bench.py")
    one = subprocess.run([sys.executable, bench] + common, capture_output=True, text=True, timeout=600, cwd=ROOT)
    many = subprocess.run([sys.executable, bench, "--gpus", str(world), "--one-device", "--exchange", "torch"] + common,
                          capture_output=True, text=True, timeout=900, cwd=ROOT)
    rep = {"world": world, "rc_one": one.returncode, "rc_many": many.returncode}
    if one.returncode or many.returncode:
        rep["stderr_tail"] = (one.stderr[-1000:], many.stderr[-2000:])
        rep["ok"] = False
        return rep
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][-1])
    lines = [ln for ln in many.stdout.splitlines() if ln.strip()]
    b = json.loads(lines[-1])
    worst = 0.0
    for k in ("sum", "abs_sum", "max_abs"):
        for x, y, s in zip(b["checksum"][k], a["checksum"][k], a["checksum"]["abs_sum"]):
            worst = max(worst, abs(x - y) / max(s, 1e-300))
    rep.update({"lines_on_stdout": len(lines), "ranks_seen": b["ranks_seen"], "tiles": b["config"]["tiles"],
                "tiles_per_gpu": b["config"]["tiles_per_gpu"], "per_rank": b["per_rank"], "exchange": b["config"]["exchange"],
                "checksum_vs_one_rank": worst, "ms_per_step": b["ms_per_step"],
                "ok": len(lines) == 1 and b["ranks_seen"] == world and worst <= 1e-12})
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rehearsal.json"))
    a = ap.parse_args()
    scratch = os.path.join(os.path.dirname(os.path.abspath(a.out)), "rehearsal_scratch")
    os.makedirs(scratch, exist_ok=True)
    rep = {"what": __doc__.split("\n\n")[0], "gpu_processes": a.world, "real_ranks": real_ranks(a.world, scratch),
           "bench_flow": bench_flow(a.world)}
    rep["ok"] = rep["real_ranks"]["ok"] and rep["bench_flow"]["ok"]
    json.dump(rep, open(a.out, "w"), indent=1)
    print(json.dumps({k: rep[k] for k in ("ok",)}), rep["real_ranks"]["ok"], rep["bench_flow"].get("ok"))
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
