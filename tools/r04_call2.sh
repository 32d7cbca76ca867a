#!/bin/bash
# round 4, GPU call 2: which stack crashes hipStreamEndCapture with a forked RCCL exchange in the capture?
set -o pipefail
mkdir -p gpurun_out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
P=tools/_bin/rccl_capture_probe
# the same probe binary on torch's bundled runtime (HIP 7.0.2 + RCCL 2.26.6), no torch in the process
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$TL timeout -k 10 120 $P > gpurun_out/r04_probe_forkjoin_torchlibs.log 2>&1; rc=$?; echo "probe fork/join on torch's libs: $rc"
tail -n 4 gpurun_out/r04_probe_forkjoin_torchlibs.log
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$TL timeout -k 10 120 $P inline > gpurun_out/r04_probe_inline_torchlibs.log 2>&1; echo "probe inline on torch's libs: $?"
tail -n 2 gpurun_out/r04_probe_inline_torchlibs.log
if [ $rc -ne 0 ]; then
  timeout -k 10 300 rocgdb -batch -ex "set env LD_LIBRARY_PATH $PWD/wxfactory_amd/lib:$TL" -ex run -ex bt --args $P > gpurun_out/r04_probe_forkjoin_torchlibs_gdb.log 2>&1; echo "rocgdb: $?"
  grep -n "SIG\|^#" gpurun_out/r04_probe_forkjoin_torchlibs_gdb.log | head -40
fi
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 600 python -m pytest tests/test_exchange_rccl_gpu.py tests/test_c_abi_example_gpu.py -x -q -m gpu --deselect tests/test_exchange_rccl_gpu.py::test_overlapped_exchange_records_into_a_graph > gpurun_out/r04_rccl_tests.log 2>&1; rc=$?; echo "pytest rccl: $rc"; tail -n 15 gpurun_out/r04_rccl_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --loopback --exchange rccl --no-extras --no-cpu-baseline > gpurun_out/r04_loopback_rccl.json 2> gpurun_out/r04_loopback_rccl.err; echo "bench loopback rccl: $?"
timeout -k 10 300 python bench.py --loopback --exchange torch --no-extras --no-cpu-baseline > gpurun_out/r04_loopback_torch.json 2> gpurun_out/r04_loopback_torch.err; echo "bench loopback torch: $?"
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err; echo "bench n1: $?"
python - <<'PY'
import json
for f in ("r04_loopback_rccl", "r04_loopback_torch", "r04_bench_n1"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 4), d["roofline"]["launch_ms"], d["roofline"].get("extrap_kernel_launch_ms"), d["config"].get("exchange"), d["per_rank"])
    except Exception as e:
        print(f, "unreadable", e)
PY
exit 0
