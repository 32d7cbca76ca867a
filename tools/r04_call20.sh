#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_euler3d_gpu.py tests/test_n8_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_exchange_rccl_gpu.py tests/test_column_metric_gpu.py -x -q -m gpu > gpurun_out/r04_grid3_tests.log 2>&1; rc=$?; echo "tests: $rc"; tail -n 6 gpurun_out/r04_grid3_tests.log
[ $rc -eq 0 ] || exit 1
{ for r in 1 2 3; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --reps 30 libwxhip_base.so libwxhip.so; done; echo "== jvp"; timeout -k 10 200 python3 tools/jvpkbench.py --reps 20; WXHIP_LIB=$PWD/wxfactory_amd/lib/libwxhip_base.so timeout -k 10 200 python3 tools/jvpkbench.py --reps 20; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_grid3_ab.log
cat gpurun_out/r04_grid3_ab.log | cut -c1-150
