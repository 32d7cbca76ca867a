#!/bin/bash
# SQ / MFMA counter passes over one E7 panel (K1 + K2) for one library variant (development tool).
#   tools/sqprof.sh <lib.so> <out_dir> [kbench args]
# Each pass is its own rocprofv3 run with --kernel-trace only (gpurun refuses --pmc with the other trace domains);
# the program (python3) comes directly after "--".
set -e
LIB=$1; OUT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export WXHIP_LIB=$ROOT/wxfactory_amd/lib/$LIB
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64" \
  "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/tools/kbench.py" --child --reps 5 "$@" > "$ROOT/$OUT/pass$i.log" 2>&1) || { echo "pass $i failed"; tail -5 "$ROOT/$OUT/pass$i.log"; }
done
