#!/bin/bash
# round 3, call E: one cache line per summary granule; pivot-column entries prefetched for the update pass; 512-lane geometry
out=gpurun_out/r03_e; mkdir -p $out
export TMPDIR=/tmp
echo "== sanity 500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean stride 64"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== 2000 lean stride 128"; JSLP_HIP_LIBRARY=build/libjslp_hip_s128.so timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== 2000 lean stride 16"; JSLP_HIP_LIBRARY=build/libjslp_hip_s16.so timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== 2000 lean 512 lanes x 4 columns"; JSLP_RES_CPT=4 timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
