#!/bin/bash
# Local helper for A/B runs: tools/ab_build.sh <name> "<extra flags>" <file.hip> [...]  ->  tools/ab/lib_<name>.so
# Recompiles only the named kernel files with the extra flags and links them with the current objects of gpismap_amd/csrc.
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../gpismap_amd/csrc"
make -s -j8
mkdir -p /tmp/ab_$name ../../tools/ab
objs=""
for o in *.o; do
  base=${o%.o}; keep=1
  for f in "$@"; do [ "$f" = "$base.hip" ] && keep=0; done
  if [ $keep = 1 ]; then objs="$objs $o"; fi
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-variable -Wno-unused-result -I. -I../../include $flags -c $f -o /tmp/ab_$name/${f%.hip}.o
  objs="$objs /tmp/ab_$name/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_$name.so $objs
echo "built tools/ab/lib_$name.so"
