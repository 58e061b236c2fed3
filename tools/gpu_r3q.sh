#!/bin/bash
# round 3: flakiness check -- the whole GPU suite twice, then the resident-kernel test files once more with wave 0 of every workgroup
# made late at every row fetch (JSLP_TEST_RESIDENT_LATE_WAVE0=1 for the whole process), then the fuzz set the same way
out=gpurun_out/r03_q2; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_$i.log 2>&1; echo "full suite run $i rc=$?"; tail -1 $out/pytest_$i.log; done
JSLP_TEST_RESIDENT_LATE_WAVE0=1 timeout 1500 python -m pytest tests/test_wide_goldens.py tests/test_cycle_goldens.py tests/test_gpu_parity.py tests/test_edge_cases.py -m gpu -q > $out/pytest_late.log 2>&1; echo "late-wave suite rc=$?"; tail -1 $out/pytest_late.log
JSLP_TEST_RESIDENT_LATE_WAVE0=1 timeout 900 python tools/fuzz_resident.py check > $out/fuzz_late.log 2>&1; echo "late-wave fuzz rc=$?"; tail -1 $out/fuzz_late.log
