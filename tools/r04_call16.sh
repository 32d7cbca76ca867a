#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --loopback --exchange rccl --no-extras --no-cpu-baseline > gpurun_out/r04_loop_rccl.json 2> gpurun_out/r04_loop_rccl.err && echo "loopback rccl ok" && \
timeout -k 10 300 python3 bench.py --loopback --exchange torch --no-extras --no-cpu-baseline > gpurun_out/r04_loop_torch.json 2> gpurun_out/r04_loop_torch.err && echo "loopback torch ok"
tail -n 5 gpurun_out/r04_loop_rccl.err
python3 - <<'PY'
import json
for f in ("r04_loop_rccl","r04_loop_torch"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["sweep_frac"], d["config"]["exchange"], d["rccl_version"])
    except Exception as e: print(f, "failed", e)
PY
