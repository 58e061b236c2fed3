#!/bin/bash
# round 3, call B: intra-XCD hand-off micro-benchmark; lean kernel with the LDS-free summary (S) and the parallel column-0 step (N)
out=gpurun_out/r03_b; mkdir -p $out
export TMPDIR=/tmp
echo "== xcd handoff bench"; timeout 120 build/xcd_handoff_bench 256 2000 2>&1 | tee $out/xcd_handoff_bench.log
echo "== sanity 200/500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
echo "== parity subset"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_wide_goldens.py tests/test_edge_cases.py -m gpu -x -q > $out/pytest_subset.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest_subset.log
