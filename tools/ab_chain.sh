#!/bin/bash
# Times the fused chain (tools/sweep_chain.py, bench workload) with the in-tree library and with every
# tools/_var/lib<tag>.so named on the command line, alternating, on one box.
# Usage (inside gpurun): bash tools/ab_chain.sh <tag> [<tag> ...]
for i in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export MI_IMAGETOOLS_LIB=$PWD/tools/_var/lib$v.so; else unset MI_IMAGETOOLS_LIB; fi
    printf "%-10s " "${v:-base}"
    timeout -k 10 100 python tools/sweep_chain.py MI_CHAIN_ROWS_PER_BAND=- 2>/dev/null | tail -1 || exit 1
  done
done
