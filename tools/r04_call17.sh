#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_sw_gpu.py -x -q -m gpu > gpurun_out/r04_sw_tests.log 2>&1; echo "sw tests: $?"; tail -n 15 gpurun_out/r04_sw_tests.log
