#!/bin/bash
# round 3, call C: two-level hand-off micro-benchmark; lean kernel with one update pass
out=gpurun_out/r03_c; mkdir -p $out
export TMPDIR=/tmp
echo "== xcd handoff bench"; timeout 120 build/xcd_handoff_bench 256 2000 2>&1 | tee $out/xcd_handoff_bench.log
echo "== sanity 500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
