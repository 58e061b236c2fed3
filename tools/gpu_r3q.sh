#!/bin/bash
# round 3, call Q: optional objectives in the lean resident kernel; fp32 sweep
out=gpurun_out/r03_q; mkdir -p $out
export TMPDIR=/tmp
echo "== 2000"; timeout 200 python tools/dense_lp_times.py 2000 2>&1 | tail -4
echo "== tests (wide goldens, parity, cycles)"; timeout 1200 python -m pytest tests/test_wide_goldens.py tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_cycle_goldens.py -m gpu -q > $out/pytest_subset.log 2>&1; echo "tests rc=$?"; tail -12 $out/pytest_subset.log | cut -c1-300
echo "== fp32 sweep"; timeout 600 python tools/fp32_sweep.py $out/fp32_sweep.md > $out/fp32_sweep.log 2>&1; echo "fp32 rc=$?"; tail -3 $out/fp32_sweep.log | cut -c1-300
