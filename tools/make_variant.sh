#!/bin/bash
# Builds the current working tree into tools/_var/lib<tag>.so (an A/B partner for
# MI_IMAGETOOLS_LIB=...), leaving the in-tree library as it was.
# Usage: bash tools/make_variant.sh <tag>
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/cudaimageprocessing_amd/libmi_imagetools.so
mkdir -p $root/tools/_var
[ -f $lib ] && cp $lib /tmp/_keep_lib.so
python3 $root/cudaimageprocessing_amd/build.py --force > /dev/null
cp $lib $root/tools/_var/lib$tag.so
[ -f /tmp/_keep_lib.so ] && cp /tmp/_keep_lib.so $lib && touch $lib
echo tools/_var/lib$tag.so
