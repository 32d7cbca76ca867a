#!/bin/bash
mkdir -p gpurun_out
for v in raw bare empty kernel interior; do
  PROBE_VARIANT=$v timeout -k 10 200 python tools/capture_probe_torch2.py > gpurun_out/r04_capture_probe2_$v.log 2>&1; echo "variant $v: $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r04_capture_probe2_$v.log | tail -n 3
done
exit 0
