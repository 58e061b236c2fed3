#!/bin/bash
# round 3: resident-kernel check (fuzz, times, goldens)
out=gpurun_out/r03_w; mkdir -p $out
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_resident.py check > $out/fuzz.log 2>&1; tail -3 $out/fuzz.log | cut -c1-300
echo "== times"; timeout 600 python tools/dense_lp_times.py 2000 > $out/times.log 2>&1; tail -4 $out/times.log
echo "== tall/wide with the cycle check"; for sh in "4000 2000" "3000 3000" "2000 4000"; do CHECK_CYCLES=1 REPEATS=2 timeout 300 python tools/tall_one.py $sh 2>&1 | tail -1 | cut -c1-200; done
echo "== tests"; timeout 900 python -m pytest tests/test_wide_goldens.py tests/test_cycle_goldens.py tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_pool_and_extras.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log | cut -c1-300
