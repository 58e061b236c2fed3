#!/bin/bash
# round 3, call Z: copy-on-write start for the single-node call
out=gpurun_out/r03_z; mkdir -p $out
export TMPDIR=/tmp
for v in 1 0; do echo "== single-node JSLP_NODE_COW_SINGLE=$v"; JSLP_NODE_COW_SINGLE=$v timeout 200 python tools/wglds_timing.py single 2>&1 | grep -v "^{" | tail -3; done
for v in 1 0; do echo "== shim Monster_II sequential JSLP_NODE_COW_SINGLE=$v"; JSLP_NODE_COW_SINGLE=$v SHIM_RUNS=8 timeout 300 node tools/shim_profile.js Monster_II 2>&1 | tail -3 | cut -c1-330; done
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_wide_goldens.py > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log | cut -c1-300
