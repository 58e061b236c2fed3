#!/bin/bash
# round 3, call I: pricing and ratio test share their barriers (4 barriers per pivot)
out=gpurun_out/r03_i; mkdir -p $out
export TMPDIR=/tmp
echo "== sanity 200/500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 200 2>&1 | tail -4; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -5
echo "== parity subset"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_wide_goldens.py tests/test_edge_cases.py -m gpu -x -q > $out/pytest_subset.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest_subset.log
