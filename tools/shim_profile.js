// Where does a Solve() through the reference host + N-API binding spend its time?  (wall clock inside the addon vs
// inside Model.solve vs the whole Solve; V8 CPU profile of the rest when run with --cpu-prof)
//   node tools/shim_profile.js <fixture name> [engine library]
"use strict";
const fs = require("fs"), path = require("path"), zlib = require("zlib");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
const M = require(path.join(root, "oracle/_ref/src/model.js")).default;
const gpu = require(path.join(root, "host/gpu-tableau.js"));
gpu.loadEngine(process.argv[3] ? { library: path.resolve(process.argv[3]) } : {});
const addon = require(path.join(root, "addon/jslp_napi.node"));
let inAddon = 0, calls = 0;
let perFn = {};
for (const k of Object.keys(addon)) {
    const f = addon[k];
    if (typeof f !== "function") continue;
    addon[k] = function () {
        const t0 = process.hrtime.bigint();
        try { return f.apply(this, arguments); } finally {
            const ms = Number(process.hrtime.bigint() - t0) / 1e6;
            inAddon += ms; calls += 1;
            const e = perFn[k] || (perFn[k] = [0, 0]);
            e[0] += ms; e[1] += 1;
        }
    };
}
// SHIM_DEFAULTS=1: the binding's default options (size policy, 16-node speculative batches); else the engine path whatever the
// size, one node at a time
if (process.env.SHIM_DEFAULTS === "1") gpu.install(T, { SlackVariable, solver });
else gpu.install(T, { SlackVariable, solver, minCells: 0, speculate: 0 });
const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root, "tests/golden/fixtures", process.argv[2] + ".json.gz"))).toString());
const origSolve = M.prototype.solve;
let tSolve = 0;
M.prototype.solve = function () { const t0 = process.hrtime.bigint(); try { return origSolve.apply(this, arguments); } finally { tSolve += Number(process.hrtime.bigint() - t0) / 1e6; } };
for (let i = 0; i < (Number(process.env.SHIM_RUNS) || 8); i++) {
    inAddon = 0; tSolve = 0; calls = 0; perFn = {};
    const m = JSON.parse(JSON.stringify(g.model));
    const t0 = process.hrtime.bigint();
    solver.Solve(m);
    const ms = Number(process.hrtime.bigint() - t0) / 1e6;
    console.log("total", ms.toFixed(2), "ms; Model.solve", tSolve.toFixed(2), "; inside the addon", inAddon.toFixed(2), "over", calls, "calls;",
        Object.keys(perFn).map((k) => k + " " + perFn[k][0].toFixed(3) + "/" + perFn[k][1]).join(", "));
}
