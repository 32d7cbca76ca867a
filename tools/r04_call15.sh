#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/tailbench.py > gpurun_out/r04_tailbench2.log 2>&1; echo "tailbench: $?"; tail -n 14 gpurun_out/r04_tailbench2.log
