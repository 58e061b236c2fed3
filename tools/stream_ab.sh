#!/bin/bash
# Same-box A/B of builds of the streaming kernels (k_pivot_fused<1|2|3>): whole-solve pivot rate of the three verified dense integer LPs beyond the register
# file, host clock, every solve checked against its known answer.  DEV_LIBS="build/libA.so build/libB.so" (built on the CPU box); `shipped` always runs; REPS rounds.
for rep in $(seq 1 ${REPS:-2}); do
for l in shipped ${DEV_LIBS:-}; do
  L="JSLP_HIP_LIBRARY=$l"; [ $l = shipped ] && L="JSLP_AB_NONE=1"
  for w in stream_5001x2001 stream_5001x3001 stream_3001x5001; do
    env $L timeout 200 python tools/pmc_workload.py $w 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$l', d['key'], round(d['whole_solve_units_per_s']), 'pivots/s', round(1e6 / d['whole_solve_units_per_s'], 2), 'us/pivot')"
  done
done
done
