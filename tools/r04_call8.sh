#!/bin/bash
# round 4, GPU call 8: the whole GPU suite on the current sources, then the default bench line
set -o pipefail
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite1.log 2>&1; rc=$?; echo "gpu suite: $rc"; tail -n 25 gpurun_out/r04_gpu_suite1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_v1.json 2> gpurun_out/r04_bench_v1.err; echo "bench: $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_v1.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "K2", r["launch_ms"], "frac", r["frac"], "sweep", r["sweep_frac"], "target", r["sweep"]["target"])
print("ceiling", r.get("ceiling"), r.get("on_measured_traffic"))
print("checksum", d["checksum"]["sum"], d["checksum"]["max_abs"])
print("column", {k: v for k, v in d["extra"]["euler_column_metric"].items() if k != "note"})
print("callers", {k: v for k, v in d["extra"]["euler_callers"].items() if k.endswith("_ms")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["flavours"]["cpp_openmp_sum_factorised"]["algorithmic_GBps_per_process"])
PY
exit 0
