#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ TRUE_METRIC=1 timeout -k 10 400 python3 tools/stepbench.py; TRUE_METRIC=1 timeout -k 10 400 python3 tools/stepbench.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stepbench.log; cat gpurun_out/r04_stepbench.log
