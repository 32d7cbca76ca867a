#!/bin/bash
# development A/B: LLVM AMDGPU scheduler strategies (WX_HIPCC_EXTRA="-mllvm -amdgpu-sched-strategy=..." python -m wxfactory_amd.build --out=libwxhip_<name>.so)
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
L="libwxhip.so libwxhip_max-ilp.so libwxhip_max-memory-clause.so libwxhip_iterative-minreg.so"
{ for r in 1 2; do timeout -k 10 300 python3 tools/kbench.py --rot-zero --reps 30 $L; done
  for l in $L; do echo "== jvp $l"; WXHIP_LIB=$PWD/wxfactory_amd/lib/$l timeout -k 10 200 python3 tools/jvpkbench.py --reps 20; done
  for l in $L; do echo "== step $l"; WXHIP_LIB=$PWD/wxfactory_amd/lib/$l TRUE_METRIC=1 timeout -k 10 300 python3 tools/stepbench.py | tail -n 1; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_sched_ab.log
cut -c1-160 gpurun_out/r04_sched_ab.log
