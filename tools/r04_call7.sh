#!/bin/bash
mkdir -p gpurun_out/tl
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
P=tools/_bin/rccl_capture_probe
# the C probe on the runtime torch bundles (HIP 7.0.2 + RCCL 2.26.6), no torch in the process: versioned names -> torch's files
ln -sf $TL/libamdhip64.so gpurun_out/tl/libamdhip64.so.7; ln -sf $TL/librccl.so gpurun_out/tl/librccl.so.1
ln -sf $TL/libhsa-runtime64.so gpurun_out/tl/libhsa-runtime64.so.1; ln -sf $TL/librocprofiler-register.so gpurun_out/tl/librocprofiler-register.so.0
ln -sf $TL/libroctx64.so gpurun_out/tl/libroctx64.so.4; ln -sf $TL/librocm_smi64.so gpurun_out/tl/librocm_smi64.so.1
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$PWD/gpurun_out/tl:$TL timeout -k 10 120 $P > gpurun_out/r04_probe_forkjoin_hip702.log 2>&1; echo "C probe fork/join on HIP 7.0.2 + RCCL 2.26.6: $?"; grep -n "HIP version\|RCCL version" gpurun_out/r04_probe_forkjoin_hip702.log; tail -n 2 gpurun_out/r04_probe_forkjoin_hip702.log
LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:$PWD/gpurun_out/tl:$TL timeout -k 10 120 $P inline > gpurun_out/r04_probe_inline_hip702.log 2>&1; echo "C probe inline on HIP 7.0.2 + RCCL 2.26.6: $?"; tail -n 1 gpurun_out/r04_probe_inline_hip702.log
rm -rf gpurun_out/tl
PROBE_VARIANT=mirror timeout -k 10 200 python tools/capture_probe_torch2.py > gpurun_out/r04_capture_probe2_mirror.log 2>&1; echo "torch variant mirror: $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r04_capture_probe2_mirror.log | tail -n 3
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout -k 10 600 python -m pytest tests/test_exchange_rccl_gpu.py tests/test_c_abi_example_gpu.py -x -q -m gpu > gpurun_out/r04_rccl_tests.log 2>&1; rc=$?; echo "pytest rccl: $rc"; tail -n 15 gpurun_out/r04_rccl_tests.log
exit 0
