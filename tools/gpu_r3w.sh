#!/bin/bash
# round 3: resident-kernel check (fuzz, times, goldens)
out=gpurun_out/r03_w; mkdir -p $out
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_resident.py check > $out/fuzz.log 2>&1; tail -3 $out/fuzz.log | cut -c1-300
echo "== times"; timeout 600 python tools/dense_lp_times.py 2000 > $out/times.log 2>&1; tail -4 $out/times.log
echo "== tests"; timeout 900 python -m pytest tests/test_wide_goldens.py tests/test_cycle_goldens.py tests/test_gpu_parity.py tests/test_edge_cases.py -m gpu -q -x -k "resident or config3 or golden or cycle or wide or soft or dense or optional or abort or hand" > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log | cut -c1-300
