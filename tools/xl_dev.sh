#!/bin/bash
# development loop of the XCD-local kernel on the GPU box: parity tests + in-kernel phase timing with the fast dev build
# (hipcc ... -DJSLP_DEV_XL_ONLY -DJSLP_DEBUG_RESIDENT -o build/libjslp_hip_xldev.so)
tag=${1:-xldev}; out=gpurun_out/$tag; mkdir -p $out
export JSLP_HIP_LIBRARY=build/libjslp_hip_xldev.so JSLP_XL=1
timeout 600 python -m pytest tests/test_xcd_local.py -m gpu -q -x -k "not opt_in" > $out/xl.log 2>&1 < /dev/null; echo "xl rc=$?"; tail -12 $out/xl.log
for n in 500 1000; do echo "== xl $n"; JSLP_FORCE_PATH=xl timeout 120 python tools/resident_phase_timing.py $n 2>&1 | grep -v wg100; done > $out/phase_timing.txt 2>&1
cat $out/phase_timing.txt
timeout 300 python tools/xl_times.py --mode > $out/xl_times.json 2> $out/xl_times.err < /dev/null; python - <<PY
import json
try:
    rows = json.loads([l for l in open("$out/xl_times.json") if l.startswith("[")][-1])
    for r in rows: print("%-44s check=%-5s %-12s pivots %5d  wall %.2f us/pivot  kernel %s" % (r["case"], r["check"], r["path"], r["pivots"], r["wall_us_per_pivot"], r["kernel_us_per_pivot"]))
except Exception as e:
    print("xl_times failed", e); print(open("$out/xl_times.err").read()[-1500:])
PY
