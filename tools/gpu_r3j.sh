#!/bin/bash
# round 3, call J: tall / wide geometries of the lean kernel against the streaming kernels; full GPU suite; bench line
out=gpurun_out/r03_j; mkdir -p $out
export TMPDIR=/tmp
for shape in "4000 2000" "2500 2000" "3000 3000" "2000 4000"; do
  echo "== $shape default path"; timeout 200 python tools/tall_one.py $shape 2>&1 | tail -1
  echo "== $shape resident (JSLP_RES_WIDE_TALL=1)"; JSLP_RES_WIDE_TALL=1 timeout 200 python tools/tall_one.py $shape 2>&1 | tail -1
done 2>&1 | tee $out/tall_wide.log
echo "== tests"; timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 2 > $out/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $out/bench.log | cut -c1-3000; tail -3 $out/bench.log | cut -c1-300
