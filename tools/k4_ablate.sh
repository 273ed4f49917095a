#!/bin/bash
# Builds shape variants of K4 (ongpis_test.hip with -DK4_QS / -DK4_MINW / -DK4_NSLOT / ...) into .ab/libk4_<tag>.so and runs
# tools/k4_bench.py against each (GPISMAP_AMD_LIB selects the library).  Usage: tools/k4_ablate.sh build | run [N...]
set -e
cd "$(dirname "$0")/.."
CS=gpismap_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$CS -Iinclude"
VARIANTS=${VARIANTS:-"q1s2:-DK4_QS=1,-DK4_MINW=4,-DK4_NSLOT=2 q1s2f:-DK4_QS=1,-DK4_MINW=4,-DK4_NSLOT=2,-DK4_SYNC_BARRIER=0 q2s2:-DK4_NSLOT=2"}
if [ "$1" = build ]; then
  mkdir -p .ab
  make -s -j8 -C $CS
  for v in $VARIANTS; do
    tag=${v%%:*}; def=${v#*:}
    def=${def//,/ }
    for f in ongpis_test.hip map_query.hip; do /opt/rocm/bin/hipcc $FLAGS $def -c $CS/$f -o .ab/${f%.hip}_$tag.o & done
    /opt/rocm/bin/hipcc $FLAGS $def -x hip -c $CS/ongpis_store.cpp -o .ab/ongpis_store_$tag.o &
    wait
    objs=$(ls $CS/*.o | grep -v -E "ongpis_test.o|map_query.o|ongpis_store.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o .ab/libk4_$tag.so $objs .ab/ongpis_test_$tag.o .ab/map_query_$tag.o .ab/ongpis_store_$tag.o
  done
else
  shift
  for v in $VARIANTS; do
    tag=${v%%:*}
    for n in "${@:-240}"; do
      echo -n "$tag: "; GPISMAP_AMD_LIB=$PWD/.ab/libk4_$tag.so timeout 120 python tools/k4_bench.py $n 64 8192 3 2>&1 | tail -1
    done
  done
fi
