#!/bin/bash
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 300 python tools/kbench.py --column --reps 30 libwxhip_base.so libwxhip.so libwxhip_base.so libwxhip.so > gpurun_out/r04_column_diet_ab.log 2>&1; echo "kbench column: $?"; grep -v amdgpu.ids gpurun_out/r04_column_diet_ab.log
timeout -k 10 600 python -m pytest tests/test_column_metric_gpu.py -x -q -m gpu > gpurun_out/r04_column_tests.log 2>&1; echo "pytest column: $?"; tail -n 6 gpurun_out/r04_column_tests.log
exit 0
