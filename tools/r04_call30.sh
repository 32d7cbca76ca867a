#!/bin/bash
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{ for r in 1 2 3; do echo "prev"; WXHIP_LIB=$PWD/wxfactory_amd/lib/libwxhip_prev.so timeout -k 10 200 python3 tools/jvpkbench.py --reps 20; echo "new"; timeout -k 10 200 python3 tools/jvpkbench.py --reps 20; done
  timeout -k 10 600 python3 -m pytest tests/test_n8_kernels_gpu.py tests/test_column_metric_gpu.py tests/test_euler3d_gpu.py -x -q -m gpu 2>&1 | tail -n 3
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_jvp_ops_ab.log
cut -c1-170 gpurun_out/r04_jvp_ops_ab.log
