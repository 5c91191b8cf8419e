#!/bin/bash
# GPU box: same-call A/B timing of library variants on the bench workload shape.
# usage: gpu_ab.sh <size> <block> <quality> lib1 lib2 ...   (each library is timed twice, interleaved)
size=$1; block=$2; q=$3; shift 3
for rep in 1 2; do
  for lib in "$@"; do python tools/time_lib.py $lib $size $block $q 2 2>&1 | tail -2 | tr "\n" " "; echo; done
done
