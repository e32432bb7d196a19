#!/bin/bash
# A/B of two builds of the library on the same box: kernel ms per view, alternating, three passes.
# usage: ab_libs.sh <libA.so> <libB.so> [workload] [views]
set -u
cd ${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; WL=${3:-cfg2_1080p_512c_b8}; VIEWS=${4:-V0,V1,V2}
for pass in 1 2 3; do
  for L in $A $B; do
    echo -n "$(basename $L): "; VRT_HIP_LIB=$PWD/$L python tools/variant_sweep.py $WL 0 ${FRAMES:-300} $VIEWS 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c48-
  done
done
