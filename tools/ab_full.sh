#!/bin/bash
# Local helper: full variant build of the library with extra flags -> tools/ab/lib_<name>.so (sources copied to /tmp, the tree is untouched)
# usage: tools/ab_full.sh <name> "<flags>"
set -e
name=$1; flags=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
d=/tmp/abfull_$name
rm -rf $d; mkdir -p $d/gpismap_amd $d/tools $d/include
cp -r $root/gpismap_amd/csrc $d/gpismap_amd/csrc
cp -r $root/include/. $d/include/
cp -r $root/tools/experiments $d/tools/experiments
rm -f $d/gpismap_amd/csrc/*.o
make -s -C $d/gpismap_amd/csrc -j8 EXTRA="$flags" 2>&1 | grep -E "error|warning: .*spill" || true
mkdir -p $root/tools/ab
cp $d/gpismap_amd/libgpismap_amd.so $root/tools/ab/lib_$name.so
echo "built tools/ab/lib_$name.so"
