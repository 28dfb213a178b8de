#!/bin/bash
# resource usage (VGPRs, scratch, occupancy) of the kernels whose mangled name matches $1 (default: all QP kernels)
cd "$(dirname "$0")/../mpc_motion_planner_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -load-store-opt -falign-loops=64 -DMPCMP_SPLIT_N25 ${EXTRA:-} \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/kres.o mpcmp.hip 2>&1 | grep -A9 "Function Name: .*${1:-k_qp}" | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill" | sed 's/.*remark: *//'
