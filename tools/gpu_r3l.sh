#!/bin/bash
# round 3, call L: full GPU suite on the new default policies, then the per-workload profile rows + PMC of the headline
out=gpurun_out/r03_l; mkdir -p $out
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -8 $out/pytest_gpu.log | cut -c1-300
bash tools/gpu_round.sh r03_l profw
bash tools/gpu_round.sh r03_l pmc
