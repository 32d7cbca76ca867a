#!/bin/bash
# round 4: stream arrangements of the twelve launches of an N = 1 evaluation; loopback rehearsal with the exchange on the origin stream
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/tailbench.py > gpurun_out/r04_tailbench.log 2>&1; echo "tailbench: $?"; tail -n 14 gpurun_out/r04_tailbench.log
timeout -k 10 300 python3 bench.py --loopback --exchange rccl --no-extras --no-cpu-baseline > gpurun_out/r04_loop_rccl.json 2> gpurun_out/r04_loop_rccl.err && echo "loopback rccl ok" && \
timeout -k 10 300 python3 bench.py --loopback --exchange torch --no-extras --no-cpu-baseline > gpurun_out/r04_loop_torch.json 2> gpurun_out/r04_loop_torch.err && echo "loopback torch ok" && \
timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r04_alias.json 2> gpurun_out/r04_alias.err && echo "aliasing ok"
python3 - <<'PY'
import json
for f in ("r04_loop_rccl","r04_loop_torch","r04_alias"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["per_rank"][0], d["roofline"]["frac"], d["roofline"]["sweep_frac"], d.get("exchange"))
    except Exception as e: print(f, "failed", e)
PY
