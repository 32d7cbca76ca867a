#!/bin/bash
mkdir -p gpurun_out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
P=tools/_bin/rccl_capture_probe
for v in "ec=360" "prio" "prio ec=360"; do
  tag=$(echo $v | tr ' =' '__')
  LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$TL timeout -k 10 120 $P $v > gpurun_out/r04_probe_${tag}_torchlibs.log 2>&1; echo "probe [$v] on torch's libs: $?"; tail -n 2 gpurun_out/r04_probe_${tag}_torchlibs.log
done
# the torch probe with the library's streams replaced by raw HIP streams (no torch pool stream in the capture but the origin)
PROBE_STEPS=a PROBE_RAW_COMM_STREAM=1 timeout -k 10 200 python tools/capture_probe_torch.py > gpurun_out/r04_capture_probe_torch_rawstream.log 2>&1; echo "torch capture probe, raw comm stream: $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r04_capture_probe_torch_rawstream.log | tail -n 4
PROBE_STEPS=a NCCL_DEBUG=INFO timeout -k 10 200 python tools/capture_probe_torch.py > gpurun_out/r04_capture_probe_torch_nccldebug.log 2>&1; echo "torch capture probe, NCCL_DEBUG: $?"
env | grep -i "nccl\|rccl\|hsa\|hip\|rocm" > gpurun_out/r04_env.txt
exit 0
