#!/bin/bash
# Kernel stats + HBM traffic + SQ counter passes over any of the tools/*bench.py programs (development tool):
#   tools/profile_prog.sh <out_dir under gpurun_out/> <tools/prog.py> [prog args]
# Every pass is its own rocprofv3 run with --kernel-trace only beside --pmc (gpurun refuses --pmc with the other trace
# domains); the program (python3) comes directly after "--".  Summaries: tools/pmc_summary.py, tools/sq_summary.py.
set -e
OUT=$1; PROG=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
D=$ROOT/gpurun_out/$OUT
mkdir -p "$D/sq"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -- python3 "$ROOT/$PROG" "$@" > "$D/stats.log" 2>&1) || { echo "stats pass failed"; tail -5 "$D/stats.log"; }
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$D/pmc_$c" -- python3 "$ROOT/$PROG" "$@" > "$D/pmc_$c.log" 2>&1) || { echo "$c pass failed"; tail -5 "$D/pmc_$c.log"; }
done
python3 "$ROOT/tools/pmc_summary.py" "$D/pmc_FETCH_SIZE" "$D/pmc_WRITE_SIZE" "$D/pmc_summary.json" > "$D/pmc_summary.txt" || true
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64" \
  "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$D/sq/pass$i" -- python3 "$ROOT/$PROG" "$@" > "$D/sq/pass$i.log" 2>&1) || { echo "sq pass $i failed"; tail -5 "$D/sq/pass$i.log"; }
done
python3 "$ROOT/tools/sq_summary.py" "$D/sq" "$D/sq_counters.json" > "$D/sq_counters.txt" 2>&1 || true
find "$D" -name "*kernel_stats.csv" | head -3
cat "$D/pmc_summary.txt"; tail -40 "$D/sq_counters.txt"
