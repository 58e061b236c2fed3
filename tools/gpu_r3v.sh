#!/bin/bash
export TMPDIR=/tmp
for lib in jslpsolver_amd/csrc/libjslp_hip.so build/libjslp_nospec.so; do echo "== $lib"; JSLP_HIP_LIBRARY=$lib timeout 300 python tools/dense_lp_times.py 2000 2>&1 | grep "3a.*False" | cut -c1-190
for sh in "4000 2000" "2000 4000"; do JSLP_HIP_LIBRARY=$lib REPEATS=2 timeout 300 python tools/tall_one.py $sh 2>&1 | tail -1 | cut -c1-200; done; done
JSLP_HIP_LIBRARY=build/libjslp_nospec.so timeout 300 python tools/resident_stress.py 600 3000 40 2>&1 | tail -1
