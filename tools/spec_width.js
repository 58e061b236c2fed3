// Solve(Monster_II) through the reference host + binding for several speculation widths (install(..., {speculate: w})):
// JIT-warm median / min of 15, relaxBatch calls per solve.   node tools/spec_width.js [fixture] [w ...]
"use strict";
const fs = require("fs"), path = require("path"), zlib = require("zlib");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
const gpu = require(path.join(root, "host/gpu-tableau.js"));
gpu.loadEngine(process.env.JSLP_ENGINE_LIBRARY ? { library: path.resolve(process.env.JSLP_ENGINE_LIBRARY) } : {});
const name = process.argv[2] || "Monster_II";
const widths = process.argv.slice(3).map(Number);
const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root, "tests/golden/fixtures", name + ".json.gz"))).toString());
for (const w of widths.length ? widths : [1, 2, 4, 8, 16, 32, 64]) {
    const uninstall = gpu.install(T, { SlackVariable, solver, speculate: w, minCells: 0 });
    const run = () => { const m = JSON.parse(JSON.stringify(g.model)); const t0 = process.hrtime.bigint(); const r = solver.Solve(m); return [Number(process.hrtime.bigint() - t0) / 1e6, r.result]; };
    for (let i = 0; i < 12; i++) run();
    const a = [];
    let res;
    for (let i = 0; i < 15; i++) { const x = run(); a.push(x[0]); res = x[1]; }
    a.sort((x, y) => x - y);
    console.log("speculate", w, "median", a[7].toFixed(2), "ms  min", a[0].toFixed(2), "ms  result", res, res === g.result.result ? "ok" : "WRONG");
    uninstall();
}
