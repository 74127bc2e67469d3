#!/bin/bash
# rocprofv3 kernel stats of tools/bench_fft.py (run on the GPU box from the repo root): bash tools/fft_stats.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_fftstats -- python3 $root/tools/bench_fft.py > $root/gpurun_out/${tag}_fftstats.log 2>&1 || { echo failed; tail -5 $root/gpurun_out/${tag}_fftstats.log; exit 1; }
f=$(find $root/gpurun_out/${tag}_fftstats -name '*kernel_stats.csv' | head -1)
grep -E "Name|fft_" "$f" | cut -c1-200
