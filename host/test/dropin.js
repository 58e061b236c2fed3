// Drop-in proof: the REFERENCE's own host (oracle/_ref = its type-erased sources: model parsing, presolve,
// branch-and-bound, result assembly) with host/gpu-tableau.js installed on its Tableau class, run over the golden
// fixtures; every result object must equal the one the unpatched reference produced (same keys, same order,
// same doubles) and the pivot sequence must have the same digest.
//   node host/test/dropin.js <engine library> <golden fixtures dir> [name filter]
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const root = path.join(__dirname, "..", "..");
const refSrc = path.join(root, "oracle", "_ref", "src");
const solver = require(path.join(refSrc, "solver.js")).default;
const Tableau = require(path.join(refSrc, "tableau", "tableau.js")).default;
const { SlackVariable } = require(path.join(refSrc, "expressions.js"));
const gpu = require(path.join(root, "host", "gpu-tableau.js"));

const library = path.resolve(process.argv[2]);
const dir = process.argv[3];
const filter = process.argv[4] || "";
const backend = gpu.loadEngine({ library });
gpu.install(Tableau, { SlackVariable });

function num(x) {
    if (typeof x !== "number") return x;
    if (Number.isFinite(x)) return Object.is(x, -0) ? "-0" : x;
    return String(x);
}
function digest(trace) {
    let h = 2166136261 | 0;
    for (let i = 0; i < trace.length; i++) h = Math.imul(h ^ trace[i], 16777619);
    return (h >>> 0).toString(16);
}

let pass = 0, fail = 0, onGpu = 0;
for (const f of fs.readdirSync(dir).filter((x) => x.endsWith(".json.gz") && x.includes(filter)).sort()) {
    const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(dir, f))).toString());
    if (!g.model) continue;
    const solution = solver.Solve(JSON.parse(JSON.stringify(g.model)), undefined, true);
    const res = solver.buildSimplifiedResult(solution);
    const got = {};
    for (const k of Object.keys(res)) got[k] = num(res[k]);
    const bad = [];
    if (JSON.stringify(Object.keys(res)) !== JSON.stringify(g.resultKeys)) bad.push("keys");
    if (JSON.stringify(got) !== JSON.stringify(g.result)) bad.push("values");
    const trace = gpu.pivotTrace(solution._tableau);
    if (trace) {
        onGpu += 1;
        if (trace.length / 2 !== g.nPivots) bad.push("pivot count " + trace.length / 2 + " != " + g.nPivots);
        else if (digest(trace) !== g.pivotDigest) bad.push("pivot digest");
        if (solution.iter !== undefined && solution.iter !== g.final.branchAndCutIterations) bad.push("B&B iterations");
        gpu.release(solution._tableau);
    }
    if (bad.length) { fail += 1; console.log("FAIL", f, bad.join("; ")); } else { pass += 1; }
}
console.log(JSON.stringify({ backend, pass, fail, solved_on_engine: onGpu }));
process.exit(fail === 0 && pass > 0 ? 0 : 1);
