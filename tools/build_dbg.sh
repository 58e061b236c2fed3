#!/bin/bash
# debug build of the engine (per-section cycle table of the LDS one-workgroup kernels): same flags as __graft_entry__.build()
# plus -DJSLP_DEBUG_WGLDS; use with JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so (tools/gpu_round.sh wgt)
set -e
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DJSLP_DEBUG_WGLDS -Iinclude \
    -o build/libjslp_hip_dbg.so jslpsolver_amd/csrc/jslp_hip.hip "$@" 2>&1 | grep -E " error:|fatal" || true
ls -la build/libjslp_hip_dbg.so
