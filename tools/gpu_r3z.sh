#!/bin/bash
# round 3: copy-on-write single-node call, stale rows compacted before slot 0 is made whole again
out=gpurun_out/r03_z2; mkdir -p $out
export TMPDIR=/tmp
for v in 1 0; do echo "== single-node JSLP_NODE_COW_SINGLE=$v"; JSLP_NODE_COW_SINGLE=$v timeout 200 python tools/wglds_timing.py single 2>&1 | grep -v "^{" | tail -2; done
for v in 1 0; do echo "== shim Monster_II sequential JSLP_NODE_COW_SINGLE=$v"; JSLP_NODE_COW_SINGLE=$v SHIM_RUNS=8 timeout 300 node tools/shim_profile.js Monster_II 2>&1 | tail -3 | cut -c1-330; done
echo "== batch rate"; timeout 120 python tools/wglds_timing.py rate 2>&1 | tail -2
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_pool_and_extras.py tests/test_node_dropin.py tests/test_speculative_bnb.py tests/test_incremental_bnb.py tests/test_enhanced_bnb.py tests/test_fuzz_services.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log | cut -c1-300
