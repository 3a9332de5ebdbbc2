#!/bin/bash
# A/B of builds of the Siegel sixteen-lanes forward units (parked trailing block): tools/park_ab.sh <out> <lib>...
O=$1; shift; LIBS="$@"
: > $O
for model in upper bounded; do
  for n in 9 10 11 12 13 14 15 16; do
    python tools/fwd_ab.py $model $n 5000 262144 $LIBS 2>&1 | grep -v amdgpu.ids >> $O
  done
done
