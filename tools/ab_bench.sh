#!/bin/bash
# A/B: bench.py with the in-tree library and with tools/_var/lib<tag>.so, alternating, on one box.
# Usage (inside gpurun): bash tools/ab_bench.sh <tag> [rounds]
tag=$1; n=${2:-2}
for i in $(seq $n); do
  for v in "" $tag; do
    if [ -n "$v" ]; then export MI_IMAGETOOLS_LIB=$PWD/tools/_var/lib$v.so; else unset MI_IMAGETOOLS_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-base}', round(d['value']), d['roofline']['avg_launch_ms'])" || exit 1
  done
done
