#!/bin/bash
# Collect the evidence bench.py and DESIGN.md quote, on the GPU box (development tool):
#   tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>/...
# 1. bench.py as the driver runs it; 2. the same command under rocprofv3 --kernel-trace --stats; 3. separate --pmc
# FETCH_SIZE / WRITE_SIZE passes over one E7 panel (27-field and rot-zero metric) + tools/pmc_summary.py;
# 4. kernel stats of the shallow-water S7 workload.  Each profiler run has the program right after "--".
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py > "$OUT/bench.json.log" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras > "$OUT/bench_profiled.json.log" 2> "$OUT/stats.err")
for mode in rotzero full; do
  extra=""; [ $mode = rotzero ] && extra="--rot-zero"
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_${mode}_fetch" -- python3 "$ROOT/tools/kbench.py" --child --reps 5 $extra > "$OUT/pmc_${mode}_fetch.log" 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_${mode}_write" -- python3 "$ROOT/tools/kbench.py" --child --reps 5 $extra > "$OUT/pmc_${mode}_write.log" 2>&1)
  python3 tools/pmc_summary.py "$OUT/pmc_${mode}_fetch" "$OUT/pmc_${mode}_write" "$OUT/pmc_${mode}_summary.json" > "$OUT/pmc_${mode}_summary.txt"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sw_stats" -- python3 "$ROOT/tools/swbench.py" > "$OUT/swbench.log" 2> "$OUT/sw_stats.err") || echo "swbench profile failed"
find "$OUT" -name "*kernel_stats.csv" | head
