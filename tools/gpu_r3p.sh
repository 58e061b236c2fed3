#!/bin/bash
# round 3: V8 CPU profile of Solve(Monster_II) through the reference host + binding (default options), top self-time functions
export TMPDIR=/tmp
rm -rf /tmp/prof; SHIM_DEFAULTS=1 SHIM_RUNS=200 node --cpu-prof --cpu-prof-dir=/tmp/prof tools/shim_profile.js Monster_II 2>&1 | tail -2 | cut -c1-300
python tools/js_profile.py $(ls /tmp/prof/*.cpuprofile | head -1) 40
