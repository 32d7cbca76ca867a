#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
ROOT=$PWD
timeout -k 10 900 python -m pytest tests/test_sw_gpu.py tests/test_column_metric_gpu.py -x -q -m gpu > gpurun_out/r04_sw_tests2.log 2>&1; rc=$?; echo "pytest sw+column: $rc"; tail -n 12 gpurun_out/r04_sw_tests2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/swbench.py > gpurun_out/r04_swbench3.log 2>&1; echo "swbench: $?"; grep -v amdgpu.ids gpurun_out/r04_swbench3.log | tail -n 7
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_sw_stats2 -- python3 $ROOT/tools/swbench.py > $ROOT/gpurun_out/r04_swbench_profiled2.log 2>&1); echo "swbench stats: $?"
find gpurun_out/r04_sw_stats2 -name "*kernel_stats.csv" -exec head -n 8 {} \; | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*agent_info.csv" -delete
exit 0
