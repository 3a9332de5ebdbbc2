#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <unit.hip>[,<unit2.hip>...] <extra hipcc flags...>
# compiles the named translation unit(s) with the extra flags and links them with the product's other objects into
# build_ab/<name>.so (the product library is not touched).  Timed against each other by tools/fwd_ab.py.
set -e
NAME=$1; UNITS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/sympa_amd/csrc
mkdir -p $ROOT/build_ab/$NAME
OBJS=""
SKIP=""
for U in ${UNITS//,/ }; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c -o $ROOT/build_ab/$NAME/${U%.hip}.o $CSRC/$U &
  SKIP="$SKIP ${U%.hip}.o"
done
wait
for O in $CSRC/*.o; do
  B=$(basename $O)
  if [[ " $SKIP " == *" $B "* ]]; then OBJS="$OBJS $ROOT/build_ab/$NAME/$B"; else OBJS="$OBJS $O"; fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $ROOT/build_ab/$NAME.so $OBJS
echo built build_ab/$NAME.so
