#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_euler3d_gpu.py -x -q -m gpu -k "rccl_exchange_path" > gpurun_out/r04_dbg.log 2>&1; echo "rc: $?"; grep -v "^  File \"/usr" gpurun_out/r04_dbg.log | head -60 | cut -c1-220
