#!/bin/bash
# round 3, call P: pipelined phase 1 in the lean kernel
out=gpurun_out/r03_p; mkdir -p $out
export TMPDIR=/tmp
echo "== sanity 200/500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 200 2>&1 | tail -4; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_wide_goldens.py tests/test_edge_cases.py tests/test_cycle_goldens.py tests/test_pool_and_extras.py -m gpu -x -q > $out/pytest_subset.log 2>&1; echo "tests rc=$?"; tail -4 $out/pytest_subset.log | cut -c1-300
echo "== shim Monster_II"; SHIM_DEFAULTS=1 SHIM_RUNS=10 timeout 120 node tools/shim_profile.js Monster_II 2>&1 | tail -3 | cut -c1-500
