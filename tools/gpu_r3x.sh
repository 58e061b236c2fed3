#!/bin/bash
# round 3, call X: fewer, fatter workgroups for the 2001 x 2001 tableau (less fabric traffic per pivot)
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python tools/dense_lp_times.py 2000 2>&1 | grep "3a.*False" | cut -c1-200; }
run JSLP_X=0
run JSLP_RES_GEOM=3
run JSLP_RES_GEOM=3 JSLP_RES_RPB=12
run JSLP_RES_GEOM=3 JSLP_RES_RPB=16
run JSLP_RES_GEOM=4 JSLP_RES_RPB=12
run JSLP_RES_GEOM=2
