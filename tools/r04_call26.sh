#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r04_b3.json 2> gpurun_out/r04_b3.err; echo "bench: $?"; tail -n 3 gpurun_out/r04_b3.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_b3.json") if l.startswith("{")][-1])
r=d["roofline"]; print(d["ms_per_step"], r["frac"], r["sweep_frac"], r["sweep"]["target"]["met"]); print(r["ceiling"]); print(r["on_measured_traffic"])
PY
