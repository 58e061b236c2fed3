#!/usr/bin/env python3
"""SURVEY.md 8d config 5 ("default vs options.useIncremental") and config 4 under the incremental service: wall time of
Solve(model) through the Python host + HIP engine (device-resident checkpoints), through the reference's own host + N-API
addon + HIP engine, and the unpatched reference on the CPU (node).  usage: tools/incremental_times.py [out.md]"""
import copy
import gzip
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Solve, _capi  # noqa: E402

NODE_SCRIPT = r"""
const fs=require('fs'),path=require('path'),zlib=require('zlib');
const root=process.argv[1], mode=process.argv[2], file=process.argv[3], options=JSON.parse(process.argv[4]);
const solver=require(path.join(root,'oracle/_ref/src/solver.js')).default;
if(mode==='gpu'){const T=require(path.join(root,'oracle/_ref/src/tableau/tableau.js')).default;
 const {SlackVariable}=require(path.join(root,'oracle/_ref/src/expressions.js'));
 const gpu=require(path.join(root,'host/gpu-tableau.js'));gpu.loadEngine(process.env.JSLP_HIP_LIBRARY?{library:path.resolve(process.env.JSLP_HIP_LIBRARY)}:{});gpu.install(T,{SlackVariable,solver,speculate:0});}
const g=JSON.parse(zlib.gunzipSync(fs.readFileSync(file)).toString());
const run=()=>{const m=JSON.parse(JSON.stringify(g.model));m.options=Object.assign({},m.options||{},options);delete m.options.timeout;
 const t0=process.hrtime.bigint();const r=solver.Solve(m,undefined,true);const ms=Number(process.hrtime.bigint()-t0)/1e6;
 const it=r._tableau.branchAndCutIterations;const res=solver.buildSimplifiedResult(r).result;
 if(mode==='gpu'){require(path.join(root,'host/gpu-tableau.js')).release(r._tableau);}
 return [ms,res,it];};
for(let i=0;i<10;i++)run();const a=[];for(let i=0;i<9;i++)a.push(run());a.sort((x,y)=>x[0]-y[0]);console.log(JSON.stringify({ms:a[4][0],result:a[4][1],iterations:a[4][2]}));
"""

POLICIES = [
    ("default service (every node from the root)", {}),
    ("useIncremental (hybrid + pseudocost)", {"useIncremental": True}),
    ("useIncremental, depth-first + most-fractional", {"useIncremental": True, "nodeSelection": "depth-first", "branching": "most-fractional"}),
    ("default service + useMIRCuts", {"useMIRCuts": True}),
    ("useIncremental + useMIRCuts", {"useIncremental": True, "useMIRCuts": True}),
]


def node(mode, path, options):
    out = subprocess.run(["node", "-e", NODE_SCRIPT, ROOT, mode, path, json.dumps(options)], capture_output=True, text=True, timeout=1800)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        return {"ms": None, "result": (out.stderr or out.stdout)[-200:], "iterations": None}


NAMES = ("Vendor_Selection", "Monster_II", "Knapsack_1")


def main(out_path=None):
    # node measurements first, while this process has no HIP context yet (a second live context on the GPU roughly
    # doubles the per-call latency of the child process)
    node_results = {}
    for name in NAMES:
        path = os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz")
        for label, options in POLICIES:
            node_results[(name, label)] = (node("gpu", path, options), node("cpu", path, options))
    lib = _capi.load_hip()
    lines = ["| model | service | relaxations | pivots | checkpoints | Python host + HIP (ms) | reference host + N-API + HIP (ms) | reference on CPU, node 12 (ms) | result (py / shim / ref) |",
             "|---|---|---|---|---|---|---|---|---|"]
    for name in NAMES:
        path = os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz")
        with gzip.open(path, "rt") as fh:
            g = json.load(fh)
        for label, options in POLICIES:
            model = copy.deepcopy(g["model"])
            model["options"] = dict(model.get("options") or {}, **options)
            model["options"].pop("timeout", None)
            Solve(model, lib=lib)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = Solve(model, lib=lib, full=True)
                ts.append(1e3 * (time.perf_counter() - t0))
            shim, ref = node_results[(name, label)]
            lines.append("| %s | %s | %d | %d | %d | %.1f | %s | %s | %s / %s / %s |" % (
                name, label, out["iter"], len(out["pivots"]), out["checkpoints"], sorted(ts)[2],
                "%.1f" % shim["ms"] if shim["ms"] else "n/a", "%.1f" % ref["ms"] if ref["ms"] else "n/a",
                out["result"]["result"], shim["result"], ref["result"]))
            if shim["iterations"] is not None and (shim["iterations"] != out["iter"] or ref["iterations"] != out["iter"]):
                lines.append("| | MISMATCH relaxations: py %s shim %s ref %s | | | | | | | |" % (out["iter"], shim["iterations"], ref["iterations"]))
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:2])
