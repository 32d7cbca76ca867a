#!/bin/bash
# Collect the evidence bench.py and DESIGN.md quote, on the GPU box (development tool):
#   WX_COMMIT=<short hash> tools/collect_profiles.sh <tag> [--pmc-only | --no-pmc | --stats-only]   -> gpurun_out/<tag>/...
# 1. separate --pmc FETCH_SIZE / WRITE_SIZE passes over one E7 panel, K1 + K2 (27-field and rot-zero metric) and the JVP
#    kernels, + tools/pmc_summary.py (bench.py quotes roofline.traffic from the K2 summaries once they are copied to
#    profiles/ - so run this BEFORE the bench whose line is to carry the figure);
# 2. bench.py as the driver runs it; 3. the same command under rocprofv3 --kernel-trace --stats with the extras, so that
#    every hot kernel (RHS, JVP, tangent extrapolation, filter, Krylov vector kernels, shallow water) has a stats row;
# 4. SQ / MFMA counter passes of K2 and of the JVP kernel; 5. the reference's benchmark matrix, shallow water S7.
# Each profiler run has the program (python3) right after "--".
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
pmc() {  # pmc <name> <program> [args]: FETCH_SIZE and WRITE_SIZE passes + summary
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    # counters only for this library's kernels: with the counter service attached to EVERY dispatch, rocprofv3 died in
    # the launch of a torch elementwise kernel of the geometry setup (tools/matrixbench.py, nine times in round 3)
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "wx::" --output-format csv -d "$OUT/pmc_${name}_$c" -- python3 "$@" > "$OUT/pmc_${name}_$c.log" 2>&1) || echo "pmc $name $c failed"
  done
  python3 tools/pmc_summary.py "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE" "$OUT/pmc_${name}_summary.json" > "$OUT/pmc_${name}_summary.txt" || true
}
if [ "$2" != "--stats-only" ] && [ "$2" != "--no-pmc" ]; then
pmc rotzero "$ROOT/tools/kbench.py" --child --reps 5 --rot-zero
pmc full "$ROOT/tools/kbench.py" --child --reps 5
pmc jvp "$ROOT/tools/jvpkbench.py" --reps 5
pmc sw "$ROOT/tools/swbench.py"
pmc n4 "$ROOT/tools/matrixbench.py" --orders 4 --reps 5     # the program the n = 4 / 6 times come from
pmc n6 "$ROOT/tools/matrixbench.py" --orders 6 --reps 5
echo "== pmc done"; cat "$OUT"/pmc_rotzero_summary.txt "$OUT"/pmc_jvp_summary.txt
# the bench below quotes roofline.traffic from profiles/<round>_pmc_*_summary.json of THIS tree (hash-checked): put them there
R=${TAG%%_*}
for k in rotzero full; do cp "$OUT/pmc_${k}_summary.json" "$ROOT/profiles/${R}_pmc_${k}_summary.json" 2>/dev/null || true; done
fi
if [ "$2" = "--stats-only" ]; then
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras > "$OUT/bench_profiled.json.log" 2> "$OUT/stats.err") || echo "bench stats failed"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_extras" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_extras_profiled.json.log" 2> "$OUT/stats_extras.err") || echo "bench extras stats failed"
elif [ "$2" != "--pmc-only" ]; then
  python3 bench.py > "$OUT/bench.json.log" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
  # (a) the headline alone: the K2 / K1 averages of this stats file are the ones to hold against the HIP-event figures
  #     of the line the same run printed; (b) with the extras: a stats row for every other hot kernel (JVP, tangent
  #     extrapolation, stage / filter, Krylov vector kernels, shallow water, the benchmark matrix)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras > "$OUT/bench_profiled.json.log" 2> "$OUT/stats.err") || echo "bench stats failed"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_extras" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_extras_profiled.json.log" 2> "$OUT/stats_extras.err") || echo "bench extras stats failed"
  sq() {  # sq <name> <program> [args]
    local name=$1; shift; local i=0
    for set in \
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
      "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64" \
      "GRBM_GUI_ACTIVE" ; do
      i=$((i+1))
      (cd /tmp && rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "wx::" --output-format csv -d "$OUT/sq_$name/pass$i" -- python3 "$@" > "$OUT/sq_${name}_pass$i.log" 2>&1) || echo "sq $name pass $i failed"
    done
    python3 tools/sq_summary.py "$OUT/sq_$name" "$OUT/sq_${name}_counters.json" > "$OUT/sq_${name}_counters.txt" 2>&1 || true
  }
  sq k2 "$ROOT/tools/kbench.py" --child --reps 5 --rot-zero
  sq jvp "$ROOT/tools/jvpkbench.py" --reps 5
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/matrix_stats" -- python3 "$ROOT/tools/matrixbench.py" --reps 20 > "$OUT/matrixbench_profiled.log" 2>&1) || echo "matrix stats failed"
  python3 tools/matrixbench.py > "$OUT/matrixbench.log" 2>&1 || true
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sw_stats" -- python3 "$ROOT/tools/swbench.py" > "$OUT/swbench_profiled.log" 2>&1) || echo "swbench profile failed"
  python3 tools/swbench.py > "$OUT/swbench.log" 2>&1 || true
fi
# keep what is quoted (stats, counter tables, summaries); drop the bulky raw traces: gpurun copies back at most 64 MiB
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT/sq_k2" "$OUT/sq_jvp" -name "*counter_collection.csv" -delete 2>/dev/null || true
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
find "$OUT" -name "*kernel_stats.csv" | head
