#!/bin/bash
# Builds the current working tree into tools/_var/lib<tag>.so (an A/B partner for
# MI_IMAGETOOLS_LIB=...), leaving the in-tree library as it was.
# Usage: bash tools/make_variant.sh <tag>
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/cudaimageprocessing_amd/libmi_imagetools.so
mkdir -p $root/tools/_var
[ -f $lib ] && cp -p $lib /tmp/_keep_lib.so
python3 $root/cudaimageprocessing_amd/build.py --force > /dev/null
cp $lib $root/tools/_var/lib$tag.so
[ -f /tmp/_keep_lib.so ] && cp -p /tmp/_keep_lib.so $lib   # (its old timestamp: the in-tree library stays "stale" if the sources changed)
echo tools/_var/lib$tag.so
