#!/bin/bash
# round 4, GPU call 3: the capture crash inside a torch process, under the debugger (expected to stop at a signal)
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 400 rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "info sharedlibrary" --args python -m pytest "tests/test_exchange_rccl_gpu.py::test_overlapped_exchange_records_into_a_graph" -x -q -m gpu > gpurun_out/r04_capture_crash_gdb.log 2>&1
echo "rocgdb: $?"
grep -n "SIG\|^#" gpurun_out/r04_capture_crash_gdb.log | head -60
exit 0
