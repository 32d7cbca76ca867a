#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ for r in 1 2 3 4; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --reps 30 libwxhip_base.so libwxhip.so libwxhip_k2g1.so; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_grid3_ab2.log
cat gpurun_out/r04_grid3_ab2.log | cut -c1-120
