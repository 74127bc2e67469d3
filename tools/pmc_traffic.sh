#!/bin/bash
# Collects HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for the
# filter kernels of bench.py.  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_traffic.sh <tag> [ENV=VAL ...]     -> gpurun_out/pmc_<tag>_{fetch,write}/
set -u
tag=$1; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_${c}
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs '' > $out.log 2>&1 || { echo "pass $c failed"; tail -5 $out.log; exit 1; }
done
python3 $root/tools/pmc_summary.py $root/gpurun_out/pmc_${tag}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_WRITE_SIZE
