#!/bin/bash
# qpbench (fixed ADMM iteration count, eps = 0) for every tools/micro/libht_<tag>.bin given: ab_many.sh "B list" tag1 tag2 ...   (run on the GPU box)
cd "$(dirname "$0")/.."
BL=$1; shift
for tag in "$@"; do
  echo "== $tag"
  QPB_LIB=tools/micro/libht_$tag.bin python tools/qpbench.py $BL 2>&1 | grep -v amdgpu.ids
done
