#!/bin/bash
# time variant builds of the split backward: bash tools/split_variants.sh build_ab/a.so build_ab/b.so ...   ("" = product)
for v in "" "$@"; do
  SYMPA_SELFCHECK=0 SYMPA_HIP_LIB=${v:+$PWD/$v} python3 tools/bwd_split_ab.py --dims 8 --models upper --split-only $SPLIT_AB_ARGS 2>&1 | grep "split "
done
