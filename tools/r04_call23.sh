#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite2.log 2>&1; rc=$?; echo "gpu suite: $rc"; tail -n 5 gpurun_out/r04_gpu_suite2.log
[ $rc -eq 0 ] || exit 1
{ for r in 1 2; do timeout -k 10 200 python3 tools/kbench.py --rot-zero --column --reps 30 libwxhip_base.so libwxhip.so; done
  echo "== swbench base"; WXHIP_LIB=$PWD/wxfactory_amd/lib/libwxhip_base.so timeout -k 10 200 python3 tools/swbench.py
  echo "== swbench new"; timeout -k 10 200 python3 tools/swbench.py
  echo "== matrixbench base"; WXHIP_LIB=$PWD/wxfactory_amd/lib/libwxhip_base.so timeout -k 10 300 python3 tools/matrixbench.py
  echo "== matrixbench new"; timeout -k 10 300 python3 tools/matrixbench.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_fastdiv_ab.log
cut -c1-175 gpurun_out/r04_fastdiv_ab.log
