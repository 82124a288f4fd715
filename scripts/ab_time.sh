#!/bin/bash
# A/B device timing of several builds of the library on ONE box: scripts/ab_time.sh "<cfg> [<cfg> ...]" lib1.so lib2.so ...   (two rounds, interleaved)
cfg=$1; shift
export PPR_DIFFPHYS_ANY_ABI=1
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    PPR_DIFFPHYS_LIB=$lib python scripts/gpu_time.py $cfg 2>&1 | grep TIMING
  done
done
