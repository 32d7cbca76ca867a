#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
ROOT=$PWD
# (1) ceiling of "vertical faces stay on chip": product vs diagnostic build 5, one box, one call, twice
timeout -k 10 300 python tools/kbench.py --rot-zero --reps 30 libwxhip.so libwxhip_novert.so libwxhip.so libwxhip_novert.so > gpurun_out/r04_vertical_faces_ceiling.log 2>&1; echo "kbench novert: $?"; grep -v amdgpu.ids gpurun_out/r04_vertical_faces_ceiling.log
# (2) shallow water: graph-replayed steps, then kernel stats of the same program
timeout -k 10 300 python tools/swbench.py > gpurun_out/r04_swbench2.log 2>&1; echo "swbench: $?"; grep -v amdgpu.ids gpurun_out/r04_swbench2.log | tail -n 4
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_sw_stats -- python3 $ROOT/tools/swbench.py > $ROOT/gpurun_out/r04_swbench_profiled.log 2>&1); echo "swbench stats: $?"
find gpurun_out/r04_sw_stats -name "*kernel_stats.csv" -exec head -n 12 {} \;
# (3) counters over the program the n = 4 / 6 times come from: collection restricted to this library's kernels
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "wx::" --output-format csv -d $ROOT/gpurun_out/r04_pmc_n4_$c -- python3 $ROOT/tools/matrixbench.py --orders 4 --reps 5 > $ROOT/gpurun_out/r04_pmc_n4_$c.log 2>&1); echo "matrixbench n=4 pmc $c: $?"
done
tail -n 3 gpurun_out/r04_pmc_n4_FETCH_SIZE.log
python3 tools/pmc_summary.py gpurun_out/r04_pmc_n4_FETCH_SIZE gpurun_out/r04_pmc_n4_WRITE_SIZE gpurun_out/r04_n4_pmc_summary.json > gpurun_out/r04_n4_pmc_summary.txt 2>&1; tail -n 12 gpurun_out/r04_n4_pmc_summary.txt
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*agent_info.csv" -delete
exit 0
