#!/bin/bash
# Copy what tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/ (development tool):
#   tools/install_evidence.sh r04_v7 [old_tag_to_remove]
set -e
TAG=$1; OLD=$2; R=${TAG%%_*}; O=gpurun_out/$TAG; P=profiles
[ -n "$OLD" ] && rm -f $P/${OLD}_*
for k in rotzero full jvp sw n4 n6; do cp $O/pmc_${k}_summary.json $P/${R}_pmc_${k}_summary.json; cp $O/pmc_${k}_summary.txt $P/${R}_pmc_${k}_summary.txt; done
for k in rotzero jvp; do for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc_${k}_$c -name "*counter_collection.csv" | head -n 1); [ -n "$f" ] && cp "$f" $P/${R}_pmc_${k}_${c}_counter_collection.csv; done; done
cp $O/bench.json.log $P/${TAG}_bench.json.log; cp $O/bench_profiled.json.log $P/${TAG}_bench_profiled.json.log; cp $O/bench_extras_profiled.json.log $P/${TAG}_bench_extras_profiled.json.log
cp $(find $O/stats -name "*kernel_stats.csv") $P/${TAG}_bench_kernel_stats.csv; cp $(find $O/stats_extras -name "*kernel_stats.csv") $P/${TAG}_bench_extras_kernel_stats.csv
cp $O/sq_k2_counters.json $P/${TAG}_k2_sq_counters.json; cp $O/sq_jvp_counters.json $P/${TAG}_jvp_sq_counters.json
cp $O/matrixbench.log $P/${TAG}_matrixbench.log; cp $(find $O/matrix_stats -name "*kernel_stats.csv") $P/${TAG}_matrixbench_kernel_stats.csv
cp $O/swbench.log $P/${TAG}_swbench.log; cp $(find $O/sw_stats -name "*kernel_stats.csv") $P/${TAG}_swbench_kernel_stats.csv
cp $P/${TAG}_bench.json.log $P/${R}_final_bench.json.log
ls $P | grep ${TAG}
