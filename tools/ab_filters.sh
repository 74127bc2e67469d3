#!/bin/bash
# tools/time_filters.py with the in-tree library and with every tools/_var/lib<tag>.so named, alternating.
# Usage (inside gpurun): FILTERS="sobel mean3" bash tools/ab_filters.sh <tag> [<tag> ...]
for i in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export MI_IMAGETOOLS_LIB=$PWD/tools/_var/lib$v.so; else unset MI_IMAGETOOLS_LIB; fi
    printf "%-8s " "${v:-base}"
    timeout -k 10 150 python tools/time_filters.py $FILTERS 2>/dev/null | tail -1 || exit 1
  done
done
