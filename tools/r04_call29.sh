#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ for r in 1 2 3; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --reps 30 libwxhip_prev.so libwxhip.so; done
  for r in 1 2; do echo "== stepbench prev"; WXHIP_LIB=$PWD/wxfactory_amd/lib/libwxhip_prev.so TRUE_METRIC=1 timeout -k 10 300 python3 tools/stepbench.py | tail -n 1; echo "== stepbench new"; TRUE_METRIC=1 timeout -k 10 300 python3 tools/stepbench.py | tail -n 1; done
  for r in 1 2; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --column --reps 30 libwxhip_prev.so libwxhip.so; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_ops_ab.log
cut -c1-150 gpurun_out/r04_ops_ab.log
