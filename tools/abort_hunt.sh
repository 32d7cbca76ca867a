#!/bin/bash
# One-off diagnosis (round 4): tests/test_euler3d_gpu.py::test_rccl_exchange_path_on_one_gpu ends the process with SIGABRT about once
# in ten runs.  Run it alone under rocgdb a few times, stop at the first signal, keep every thread's backtrace.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/abort_hunt; mkdir -p "$OUT"
export LD_LIBRARY_PATH=$ROOT/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH NCCL_DEBUG=WARN WX_RCCL_TEST_CHILD=1
cd "$ROOT"
for i in $(seq 1 ${1:-8}); do
  timeout -k 10 180 rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 SIGUSR2 SIGCHLD nostop noprint pass" -ex run \
      -ex "thread apply all bt 30" --args python3 -X faulthandler -m pytest tests/test_euler3d_gpu.py -x -q -s -m gpu -k rccl_exchange_path \
      > "$OUT/run_$i.log" 2>&1
  echo "run $i: rc=$? $(grep -c 'received signal' "$OUT/run_$i.log") signal(s); $(grep -E 'passed|failed' "$OUT/run_$i.log" | tail -n 1)"
  if grep -q "received signal" "$OUT/run_$i.log"; then break; fi
done
