#!/bin/bash
# HBM traffic (separate --pmc passes) of one filter kernel over tools/run_filter.py.
# Usage (on the GPU box, from the repo root): bash tools/pmc_filter.sh <tag> <filter> [ENV=VAL ...]
set -u
tag=$1; filt=$2; shift 2
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_${c}
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out -- python3 $root/tools/run_filter.py $filt 3 16 > $out.log 2>&1 || { echo "pass $c failed"; tail -5 $out.log; exit 1; }
done
python3 $root/tools/pmc_summary.py $root/gpurun_out/pmc_${tag}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_WRITE_SIZE | grep -v synth
