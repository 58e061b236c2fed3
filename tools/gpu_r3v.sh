#!/bin/bash
# round 3, call V: which of the two changes (one-barrier pricing / ratio test overlapped with the update) breaks fuzz instance 8
out=gpurun_out/r03_v; mkdir -p $out
export TMPDIR=/tmp
for lib in jslpsolver_amd/csrc/libjslp_hip.so build/libjslp_nospec.so build/libjslp_noovl.so; do
  echo "== $lib"; FUZZ_ONLY=4,8,16,24,28,36 JSLP_HIP_LIBRARY=$lib timeout 300 python tools/fuzz_resident.py check 2>&1 | tail -3 | cut -c1-400
  JSLP_HIP_LIBRARY=$lib timeout 300 python tools/dense_lp_times.py 2000 2>&1 | grep "3a" | cut -c1-200
done
echo "== phase timing"; JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 300 python tools/resident_phase_timing.py 2000 > $out/phase.log 2>&1; tail -8 $out/phase.log | cut -c1-400
