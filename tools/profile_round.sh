#!/bin/bash
# One round's profile evidence for bench.py (run on the GPU box from the repo root):
#   bash tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_stats/
# 2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes              -> gpurun_out/<tag>_pmc_*/
# 3. the per-kernel HBM traffic table + JSON                         -> gpurun_out/<tag>_traffic.{txt,json}
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out
# (--legs '': the C3 leg launches the same chain kernel on 4096 frames, which would mix two launch sizes into one average)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py --no-cpu-baseline --legs '' > $out/${tag}_stats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/${tag}_stats.log; exit 1; }
# the other configurations' kernels (C3, C4, C5 legs), few C2 steps
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_legs_stats -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/${tag}_legs_stats.log 2>&1 || { echo "legs stats pass failed"; tail -5 $out/${tag}_legs_stats.log; exit 1; }
grep -h '^{"metric' $out/${tag}_legs_stats.log > $out/${tag}_bench_legs_line_profiled.json
f=$(find $out/${tag}_legs_stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_legs_kernel_stats.csv
grep -h '^{"metric' $out/${tag}_stats.log > $out/${tag}_bench_line_profiled.json   # bench.py's own line from the profiled process
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs '' > $out/${tag}_pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/${tag}_pmc_$c.log; exit 1; }
done
python3 $root/tools/pmc_summary.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE --json $out/${tag}_traffic.json > $out/${tag}_traffic.txt
cat $out/${tag}_traffic.txt
f=$(find $out/${tag}_stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_kernel_stats.csv && cat $out/${tag}_kernel_stats.csv

# 4. the FFT round trip (C5): kernel times and HBM traffic of the two-pass kernels (tools/bench_fft.py: 16 planes of 4096x4096)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_fft_stats -- python3 $root/tools/bench_fft.py > $out/${tag}_fft_stats.log 2>&1 || { echo "fft stats pass failed"; tail -5 $out/${tag}_fft_stats.log; exit 1; }
f=$(find $out/${tag}_fft_stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_fft_kernel_stats.csv && grep -E "Name|fft" $out/${tag}_fft_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_fft_pmc_$c -- python3 $root/tools/bench_fft.py > $out/${tag}_fft_pmc_$c.log 2>&1 || { echo "fft pass $c failed"; tail -5 $out/${tag}_fft_pmc_$c.log; exit 1; }
done
python3 $root/tools/pmc_summary.py $out/${tag}_fft_pmc_FETCH_SIZE $out/${tag}_fft_pmc_WRITE_SIZE | grep -E "fft" > $out/${tag}_fft_traffic.txt
cat $out/${tag}_fft_traffic.txt
