#!/bin/bash
# A/B timing of the working tree's k_qp2 against HEAD at a fixed ADMM iteration count (tools/ablate.py, best of 6):
#   bash tools/ab.sh            (builds both libraries here, runs on the GPU box through gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
TMP=$(mktemp -d)
git show HEAD:mpc_motion_planner_amd/csrc/qp_kernel_v2.hpp > "$TMP/qp_kernel_v2.hpp"
for f in mpc_motion_planner_amd/csrc/*.hpp mpc_motion_planner_amd/csrc/mpcmp.hip; do [ "$(basename $f)" = qp_kernel_v2.hpp ] || cp "$f" "$TMP/"; done
mkdir -p "$TMP/../include_ab" && true
( cd "$TMP" && sed -i 's|"../../include/mpcmp.h"|"'"$ROOT"'/include/mpcmp.h"|' mpcmp.hip *.hpp && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -load-store-opt -falign-loops=64 -o "$ROOT/tools/micro/libabl0.bin" mpcmp.hip ) &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -load-store-opt -falign-loops=64 -o tools/micro/libabl11.bin mpc_motion_planner_amd/csrc/mpcmp.hip &
wait
rm -rf "$TMP"
timeout 3000 /usr/local/graft/bin/gpurun --timeout 900 -- 'python tools/ablate.py 0 11 0 11 0 11 0 11 2>&1 | tail -8' 2>&1 | tail -8
