#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ for r in 1 2 3; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --column --reps 30 libwxhip.so libwxhip_pf.so; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pf_ab.log
cut -c1-150 gpurun_out/r04_pf_ab.log
