#!/bin/bash
# builds tools/micro/libht_<tag>.bin with the extra compiler arguments given per variant: ht_build.sh tag1 "-DX=1" tag2 "-DX=2" ...
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  (/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Xclang -target-feature -Xclang -load-store-opt -falign-loops=64 $defs \
     -o tools/micro/libht_$tag.bin mpc_motion_planner_amd/csrc/mpcmp.hip 2>/tmp/ht_$tag.err; echo built $tag rc=$?) &
done
wait
