#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/dense_lp_times.py 2000 2>&1 | grep "3a\|3b.*False" | cut -c1-190
for sh in "4000 2000" "3000 3000" "2000 4000"; do REPEATS=2 timeout 300 python tools/tall_one.py $sh 2>&1 | tail -1 | cut -c1-200; done
timeout 500 python tools/resident_stress.py 600 3000 80 2>&1 | tail -2 | cut -c1-300
timeout 500 python tools/resident_stress.py 4000 2000 8 2>&1 | tail -2 | cut -c1-300
timeout 600 python tools/fuzz_resident.py check 2>&1 | tail -1 | cut -c1-200
