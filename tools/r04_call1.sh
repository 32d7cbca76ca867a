#!/bin/bash
# round 4, GPU call 1: the native RCCL exchange - capture probe, parity tests, C example, loopback bench, then the torch
# async-capture crash under rocgdb (last: it is expected to abort)
set -o pipefail
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
P=tools/_bin/rccl_capture_probe
timeout -k 10 120 $P > gpurun_out/r04_probe_forkjoin_global.log 2>&1; echo "probe fork/join global: $?"
timeout -k 10 120 $P threadlocal > gpurun_out/r04_probe_forkjoin_threadlocal.log 2>&1; echo "probe fork/join threadlocal: $?"
timeout -k 10 120 $P inline > gpurun_out/r04_probe_inline_global.log 2>&1; echo "probe inline: $?"
for f in gpurun_out/r04_probe_*.log; do tail -n 2 $f; done
timeout -k 10 600 python -m pytest tests/test_exchange_rccl_gpu.py tests/test_c_abi_example_gpu.py -x -q -m gpu > gpurun_out/r04_rccl_tests.log 2>&1; rc=$?; echo "pytest rccl: $rc"; tail -15 gpurun_out/r04_rccl_tests.log
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
# the crash r03 hid behind async_op=False: torch's all_to_all_single with a deferred wait inside a capture, under the debugger
PROBE_ASYNC=1 timeout -k 10 300 rocgdb -batch -ex run -ex bt -ex "info threads" --args python tools/graphcoll_probe.py > gpurun_out/r04_torch_async_capture_gdb.log 2>&1; echo "rocgdb torch async capture: $?"
grep -n "SIGABRT\|SIGSEGV\|#0\|#1 \|#2 \|#3 \|#4 \|#5 \|#6 \|#7 \|#8 \|#9 \|#1[0-9] " gpurun_out/r04_torch_async_capture_gdb.log | head -40
exit 0
