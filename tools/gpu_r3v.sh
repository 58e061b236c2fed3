#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/dense_lp_times.py 2000 2>&1 | grep "3a\|3b.*False" | cut -c1-190
timeout 500 python tools/resident_stress.py 600 3000 120 2>&1 | tail -3 | cut -c1-300
timeout 500 python tools/resident_stress.py 4000 2000 12 2>&1 | tail -3 | cut -c1-300
timeout 300 python tools/resident_stress.py 1200 2100 30 2>&1 | tail -2 | cut -c1-300
