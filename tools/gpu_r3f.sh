#!/bin/bash
# round 3, call F: replicated summary granules (readers per line: 256 / REPL)
out=gpurun_out/r03_f; mkdir -p $out
export TMPDIR=/tmp
echo "== sanity 500"; JSLP_FORCE_PATH=resident timeout 120 python tools/dense_lp_times.py 500 2>&1 | tail -4
for v in "" r16 r32 r1; do
  echo "== 2000 lean REPL ${v:-8 (default)}"
  if [ -z "$v" ]; then timeout 200 python tools/dense_lp_times.py 2000 2>&1 | grep "3a"; else JSLP_HIP_LIBRARY=build/libjslp_hip_$v.so timeout 200 python tools/dense_lp_times.py 2000 2>&1 | grep "3a"; fi
done
echo "== phase timing lean"; JSLP_HIP_LIBRARY=build/libjslp_hip_resdbg.so timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | tail -6
