#!/bin/bash
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 300 python tools/copybench.py > gpurun_out/r04_copybench.log 2>&1; echo "copybench: $?"; sort -t: -k2 -n gpurun_out/r04_copybench.log | tail -n 8
timeout -k 10 600 python - > gpurun_out/r04_cpu_baseline.log 2>&1 <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
r = bench.cpu_baseline(8, 8, 20250824, 60)
print(json.dumps(r))
PY
echo "cpu baseline: $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_cpu_baseline.log").read().strip().splitlines()[-1])
print(d["value"], d["flavours"]["cpp_openmp_sum_factorised"])
PY
exit 0
