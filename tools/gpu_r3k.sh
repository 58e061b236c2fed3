#!/bin/bash
export TMPDIR=/tmp
for s in "3000 3000"; do
  echo "== $s default, 2 repeats"; REPEATS=2 timeout 200 python tools/trace_compare.py check $s 2>&1 | tail -2
  echo "== $s general build, 2 repeats"; REPEATS=2 JSLP_RES_LEAN=0 JSLP_RES_WIDE_TALL=1 timeout 200 python tools/trace_compare.py check $s 2>&1 | tail -2
  echo "== tall_one"; JSLP_RES_WIDE_TALL=1 timeout 200 python tools/tall_one.py 3000 3000 2>&1 | tail -1
done
