#!/usr/bin/env python3
"""End-to-end wall times of BASELINE.json's configs on this box: Solve(model) through the Python host + HIP engine,
through the reference's own host + N-API addon + HIP engine (node), and the unpatched reference (node, CPU).
Writes a markdown table.  usage: tools/config_times.py [out.md]"""
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
const root=process.argv[1], mode=process.argv[2], file=process.argv[3];
const solver=require(path.join(root,'oracle/_ref/src/solver.js')).default;
if(mode!=='cpu'){const T=require(path.join(root,'oracle/_ref/src/tableau/tableau.js')).default;
 const {SlackVariable}=require(path.join(root,'oracle/_ref/src/expressions.js'));
 const gpu=require(path.join(root,'host/gpu-tableau.js'));gpu.loadEngine({});gpu.install(T,{SlackVariable,solver,speculate:mode==='gpu16'?16:1});}
const g=JSON.parse(zlib.gunzipSync(fs.readFileSync(file)).toString());
const run=()=>{const m=JSON.parse(JSON.stringify(g.model));const t0=process.hrtime.bigint();const r=solver.Solve(m);return [Number(process.hrtime.bigint()-t0)/1e6,r.result];};
for(let i=0;i<10;i++)run();const a=[];for(let i=0;i<9;i++)a.push(run()[0]);a.sort((x,y)=>x-y);console.log(JSON.stringify({ms:a[4],result:run()[1]}));
"""


def node(mode, path):
    out = subprocess.run(["node", "-e", NODE_SCRIPT, ROOT, mode, path], capture_output=True, text=True, timeout=900)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        return {"ms": None, "result": (out.stderr or out.stdout)[-200:]}


CASES = (("1 Berlin Airlift (fixture variant)", "Berlin_Air_Lift_Problem"), ("2 Monster LP", "Monster_Problem"),
         ("4 Monster_II MIP", "Monster_II"), ("5 Vendor Selection", "Vendor_Selection"), ("LargeFarmMIP", "LargeFarmMIP"))


def main(out_path=None):
    # the node columns first, while this process has no HIP context yet: a second live context on the GPU roughly
    # doubles the per-call latency of the child process
    node_results = {}
    for _label, name in CASES:
        path = os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz")
        node_results[name] = (node("cpu", path), node("gpu", path), node("gpu16", path))
    lib = _capi.load_hip()
    rows = []
    for label, name in CASES:
        path = os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz")
        with gzip.open(path, "rt") as fh:
            g = json.load(fh)
        model = g["model"]
        times = {}
        for spec in (1, 16):
            Solve(model, lib=lib, speculate=spec)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                r = Solve(model, lib=lib, speculate=spec)
                ts.append(1e3 * (time.perf_counter() - t0))
            times[spec] = (sorted(ts)[2], r["result"])
        ref, shim, shim16 = node_results[name]
        rows.append((label, g["tableau"]["height"], g["tableau"]["width"], g["nPivots"], len(g["simplexCalls"]), times[1], times[16], shim, shim16, ref))
    lines = ["| config | tableau | pivots | LP relaxations | Python host + HIP (ms) | same, 16-node speculative batches (ms) | reference host + N-API + HIP, install(..., {speculate: 1}) (ms) | same with the default 16-node speculative batches (ms) | reference TS on CPU, node 12 (ms) | result |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    fmt = lambda x: "%.1f" % x["ms"] if x["ms"] else "n/a"
    for label, h, w, p, n, t1, t16, shim, shim16, ref in rows:
        lines.append("| %s | %dx%d | %d | %d | %.1f | %.1f | %s | %s | %s | %s / %s / %s / %s |" % (
            label, h, w, p, n, t1[0], t16[0], fmt(shim), fmt(shim16), fmt(ref), t1[1], shim["result"], shim16["result"], ref["result"]))
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:2])
