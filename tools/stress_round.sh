#!/bin/bash
# >= 5 M pivots of tools/resident_stress.py over the five register-resident geometries, both pipelines, the cycle check, unrestricted
# variables: every run against its KNOWN answer (exit status != 0 on any difference or resident abort).  usage: tools/stress_round.sh <out file>
# STRESS_DIV=n divides every run count by n (the chaos build sleeps thousands of cycles per pivot)
out=${1:-gpurun_out/resident_stress.txt}; mkdir -p $(dirname $out); : > $out; rc=0; div=${STRESS_DIV:-1}
run() { local m=$1 n=$2 r=$(( ($3 + div - 1) / div )); shift 3; timeout 900 python tools/resident_stress.py $m $n $r "$@" >> $out 2>&1 || rc=1; tail -1 $out; }
run 2000 2000 110                      # <1024,2,8> lean phase 2: 1.4 M pivots
run 2000 2000 30 --check               # ... with the cycle check
run 4000 2000 40                       # <512,4,16>
run 3000 3000 24                       # <512,6,12>
run 2000 4000 50                       # <512,8,8>
run 600 3000 250                       # partial-line writers (the shapes that went wrong before the release fence)
run 1200 2100 120
run 2000 2000 300 --kind lp            # phase-1 pipeline (config 3b)
run 1000 1000 300 --kind int2p         # phase 1 + phase 2, headline geometry
run 2100 300 200 --kind int2p          # fused phase 1 -> resident phase 2 (tall)
JSLP_FORCE_PATH=resident run 1200 2100 100 --unr 3   # lean build with unrestricted variables, wide geometry
run 1000 1000 100 --unr 3              # ... headline geometry
python3 - "$out" <<'PY'
import re, sys
tot = 0; bad = 0
for l in open(sys.argv[1]):
    m = re.search(r"(\d+) pivots, path \S+: (\d+) differ", l)
    if m: tot += int(m.group(1)); bad += int(m.group(2))
print("TOTAL: %d pivots, %d runs differing from their known answer" % (tot, bad))
PY
exit $rc
