#!/bin/bash
export TMPDIR=/tmp
echo "== shipping build"; for i in 1 2; do timeout 120 python tools/wglds_timing.py rate 2>&1 | tail -1; done
WATCHED=1 timeout 120 python tools/wglds_timing.py rate 2>&1 | tail -1
echo "== 64-VGPR build with the lane test recomputed"; for i in 1 2 3; do JSLP_HIP_LIBRARY=build/libjslp_hip_w8.so timeout 120 python tools/wglds_timing.py rate 2>&1 | tail -1; done
WATCHED=1 JSLP_HIP_LIBRARY=build/libjslp_hip_w8.so timeout 120 python tools/wglds_timing.py rate 2>&1 | tail -1
JSLP_HIP_LIBRARY=build/libjslp_hip_w8.so timeout 300 python -m pytest tests/test_pool_and_extras.py -m gpu -q -k "large_batch or watched_batch" 2>&1 | tail -2
JSLP_HIP_LIBRARY=build/libjslp_hip_w8.so timeout 120 python tools/queue_check.py 2>&1 | tail -2
