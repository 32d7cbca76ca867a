#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ echo "== general kernel"; timeout -k 10 200 python3 tools/kstamps.py; echo "== column kernel"; timeout -k 10 200 python3 tools/kstamps.py --column; } > gpurun_out/r04_kstamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_kstamps.log
