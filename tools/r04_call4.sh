#!/bin/bash
mkdir -p gpurun_out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
P=tools/_bin/rccl_capture_probe
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$TL timeout -k 10 120 $P multi > gpurun_out/r04_probe_multi_torchlibs.log 2>&1; echo "probe multi on torch's libs: $?"; tail -n 3 gpurun_out/r04_probe_multi_torchlibs.log
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib timeout -k 10 120 $P multi > gpurun_out/r04_probe_multi_rocm72.log 2>&1; echo "probe multi on /opt/rocm: $?"; tail -n 3 gpurun_out/r04_probe_multi_rocm72.log
for s in a b ab abc; do
  PROBE_STEPS=$s timeout -k 10 200 python tools/capture_probe_torch.py > gpurun_out/r04_capture_probe_torch_$s.log 2>&1; echo "torch capture probe steps=$s: $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r04_capture_probe_torch_$s.log | tail -n 6
done
exit 0
