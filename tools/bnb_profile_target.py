#!/usr/bin/env python3
"""rocprofv3 target: sequential branch-and-bound of one fixture (default Monster_II), a few solves, so that the kernel
trace shows the per-node launch costs of a REAL tree walk (one node per engine call) next to the wall time.
usage: tools/bnb_profile_target.py [fixture name] [solves]"""
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Solve, _capi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Monster_II"
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz"), "rt") as fh:
    g = json.load(fh)
lib = _capi.load_hip()
Solve(g["model"], lib=lib)
t0 = time.perf_counter()
for _ in range(solves):
    r = Solve(g["model"], lib=lib, full=True)
ms = 1e3 * (time.perf_counter() - t0) / solves
print(json.dumps({"fixture": name, "ms_per_solve": ms, "relaxations": r["iter"] + 1, "pivots": len(r["pivots"])}))
