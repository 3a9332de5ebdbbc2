#!/bin/bash
# A/B of builds of the spd forward unit on one box: tools/spd_ab.sh <out> <n> <variant.so>...   ("product" = the in-tree library)
O=$1; N=$2; shift 2
for v in "$@"; do
  echo "== $v n=$N" >> $O
  if [ "$v" = product ]; then python tools/spd_time.py $N >> $O 2>&1; else SYMPA_HIP_LIB=$v python tools/spd_time.py $N >> $O 2>&1; fi
done
