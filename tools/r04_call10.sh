#!/bin/bash
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 900 python -m pytest tests/test_sw_gpu.py tests/test_exchange_rccl_gpu.py tests/test_reserve_gpu.py -x -q -m gpu > gpurun_out/r04_sw_tests.log 2>&1; rc=$?; echo "pytest sw: $rc"; tail -n 12 gpurun_out/r04_sw_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/swbench.py > gpurun_out/r04_swbench.log 2>&1; echo "swbench: $?"; grep -v "amdgpu.ids" gpurun_out/r04_swbench.log | tail -n 8
exit 0
