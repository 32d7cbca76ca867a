#!/bin/bash
# development A/B of K1 (extrapolation kernel): workgroups per CU capped through unused dynamic LDS, plain / non-temporal stores
# needs libwxhip_k1exp.so (-DWX_K1_EXPERIMENT) and libwxhip_k1nt.so (-DWX_K1_EXPERIMENT -DWX_K1_NT_STORE)
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
for rep in 1 2; do
for lib in libwxhip_k1exp.so libwxhip_k1nt.so; do
  for dyn in 0 17000 30000 57000 100000; do
    echo -n "dyn_lds=$dyn "; WX_K1_DYN_LDS=$dyn timeout -k 10 120 python3 tools/kbench.py --rot-zero --reps 30 $lib | grep -v amdgpu.ids
  done
done
done
