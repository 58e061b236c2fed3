#!/bin/bash
# round 3, call D: granule spacing in the chip-wide all-gather; lean kernel with the parallel pending-cell evaluation
out=gpurun_out/r03_d; mkdir -p $out
export TMPDIR=/tmp
for st in 1 8 32 512; do timeout 120 build/xcd_handoff_bench 256 2000 $st 2>&1 | grep "flavour 4\|flavour 0\|flavour 6" | cut -c1-150 | tee -a $out/xcd_handoff_stride.log; done
echo "== sanity 500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
