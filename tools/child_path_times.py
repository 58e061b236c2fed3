#!/usr/bin/env python3
"""Which launch shape should ONE branch-and-bound child use?  Runs Solve(model) end to end for a few fixtures with the
single-child threshold (JSLP_WG_CELLS_CHILD, cells = row capacity x padded width) forced low (children through the
chip-wide resident kernel) and high (children in one workgroup).  One subprocess per setting: the knob is read once.
usage: tools/child_path_times.py [out.md]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import gzip, json, os, sys, time
sys.path.insert(0, %r)
from jslpsolver_amd import Solve, _capi
lib = _capi.load_hip()
out = {}
for name in ("Monster_II", "Vendor_Selection", "Knapsack_1", "StockCuttingProblem"):
    with gzip.open(os.path.join(%r, "tests", "golden", "fixtures", name + ".json.gz"), "rt") as fh:
        g = json.load(fh)
    Solve(g["model"], lib=lib)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = Solve(g["model"], lib=lib, full=True); ts.append(1e3 * (time.perf_counter() - t0))
    out[name] = {"ms": sorted(ts)[2], "result": r["result"]["result"], "iter": r["iter"], "pivots": len(r["pivots"])}
print(json.dumps(out))
""" % (ROOT, ROOT)


def main(out_path=None):
    rows = {}
    for label, cells in (("one workgroup per child", 1 << 40), ("default threshold", None), ("chip-wide (resident) per child", 0)):
        env = dict(os.environ)
        if cells is not None:
            env["JSLP_WG_CELLS_CHILD"] = str(cells)
        out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=900)
        try:
            rows[label] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            rows[label] = {"error": (out.stderr or out.stdout)[-300:]}
    names = ["Monster_II", "Vendor_Selection", "Knapsack_1", "StockCuttingProblem"]
    lines = ["| single-child launch shape | " + " | ".join("%s (ms)" % n for n in names) + " |", "|---|" + "---|" * len(names)]
    for label, r in rows.items():
        if "error" in r:
            lines.append("| %s | %s |" % (label, r["error"].replace("\n", " ")))
        else:
            lines.append("| %s | " % label + " | ".join("%.1f (%d relaxations, %d pivots)" % (r[n]["ms"], r[n]["iter"], r[n]["pivots"]) for n in names) + " |")
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        open(out_path, "w").write(text)


if __name__ == "__main__":
    main(*sys.argv[1:2])
