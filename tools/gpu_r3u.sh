#!/bin/bash
# round 3, call U: one-barrier pricing + ratio test overlapped with the row update in the lean resident kernel
out=gpurun_out/r03_u; mkdir -p $out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_wide_goldens.py tests/test_cycle_goldens.py tests/test_gpu_parity.py -m gpu -q -x -k "resident or config3 or golden or cycle or wide or soft or dense or optional" > $out/pytest.log 2>&1; echo "rc=$?"; tail -4 $out/pytest.log | cut -c1-300
echo "== fuzz"; timeout 600 python tools/fuzz_resident.py check > $out/fuzz.log 2>&1; tail -2 $out/fuzz.log
echo "== times"; timeout 600 python tools/dense_lp_times.py 2000 > $out/times.log 2>&1; tail -12 $out/times.log
echo "== phase timing"; JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 300 python tools/resident_phase_timing.py 2000 > $out/phase.log 2>&1; tail -8 $out/phase.log | cut -c1-400
