#!/bin/bash
# One counter pass over tools/kbench.py for several library variants (development tool):
#   tools/sqone.sh "<counters>" <out_dir under gpurun_out> lib1.so lib2.so ... [-- kbench args]
set -e
CNT=$1; OUT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for lib in "${LIBS[@]}"; do
  D=$ROOT/gpurun_out/$OUT/${lib%.so}
  mkdir -p "$D"
  (cd /tmp && WXHIP_LIB=$ROOT/wxfactory_amd/lib/$lib rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$D" -- python3 "$ROOT/tools/kbench.py" --child --reps 5 "$@" > "$D/log.txt" 2>&1) || { echo "$lib failed"; tail -3 "$D/log.txt"; }
  echo "== $lib"; python3 "$ROOT/tools/sq_summary.py" "$D" "$D/summary.json" | grep -A12 "euler_rhs_kernel" | head -14
done
