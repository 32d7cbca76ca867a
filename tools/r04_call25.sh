#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite2.log 2>&1; rc=$?; echo "gpu suite: $rc"; grep -v "^  File \"/usr\|^Extension modules" gpurun_out/r04_gpu_suite2.log | head -70 | cut -c1-250
