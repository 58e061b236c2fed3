#!/bin/bash
# round 3, call A: first contact of the lean resident kernel with the GPU (short timeouts first: a hang must not eat the box)
out=gpurun_out/r03_a; mkdir -p $out
export TMPDIR=/tmp
echo "== sanity n=200/500 (lean)"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 200 2>&1 | tail -4
JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
echo "== 2000 lean";   timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== 2000 general (JSLP_RES_LEAN=0)"; JSLP_RES_LEAN=0 timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== 2000 lean, early poll"; JSLP_HIP_LIBRARY=build/libjslp_hip_ep.so timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
echo "== phase timing general"; JSLP_RES_LEAN=0 JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
echo "== tests"; timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 $out/pytest_gpu.log
