#!/bin/bash
# A/B of the XCD strip-tile element order (WX_XCD_TILE builds): kbench times, then FETCH_SIZE of K2 per variant.
# usage (on the GPU box): bash tools/xcdtile_ab.sh lib1.so lib2.so ...
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/xcdtile; mkdir -p "$OUT"
export LD_LIBRARY_PATH=$ROOT/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH TMPDIR=/tmp
for r in 1 2 3; do timeout -k 10 300 python3 tools/kbench.py --rot-zero --reps 30 "$@" || exit 1; done | tee "$OUT/kbench.log"
for lib in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && WXHIP_LIB=$ROOT/wxfactory_amd/lib/$lib timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "wx::" --output-format csv -d "$OUT/${lib}_$c" -- python3 "$ROOT/tools/kbench.py" --child --reps 5 --rot-zero > "$OUT/${lib}_$c.log" 2>&1) || { echo "pmc $lib $c failed"; exit 1; }
  done
  python3 tools/pmc_summary.py "$OUT/${lib}_FETCH_SIZE" "$OUT/${lib}_WRITE_SIZE" "$OUT/${lib}_summary.json" > "$OUT/${lib}_summary.txt" || true
  echo "== $lib"; cat "$OUT/${lib}_summary.txt"
done
