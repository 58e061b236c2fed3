// Solve(model) through the reference host + binding with the speculative service's lookahead set to each of argv[3..] (extra nodes
// per batch: children of cached, not yet committed nodes); JIT-warm median of 21, result checked against the reference's golden.
//   node tools/spec_lookahead.js Monster_II 0 16 32
"use strict";
const fs = require("fs"), path = require("path"), zlib = require("zlib");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
const gpu = require(path.join(root, "host/gpu-tableau.js"));
gpu.loadEngine(process.env.JSLP_LIBRARY ? { library: process.env.JSLP_LIBRARY } : {});
const name = process.argv[2] || "Monster_II";
const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root, "tests/golden/fixtures", name + ".json.gz"))).toString());
const med = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
let uninstall = null;
for (const la of process.argv.slice(3).map(Number)) {
    if (uninstall) uninstall();
    uninstall = gpu.install(T, { SlackVariable, solver, lookahead: la, minCells: process.env.MIN_CELLS !== undefined ? Number(process.env.MIN_CELLS) : undefined });
    const run = () => { const m = JSON.parse(JSON.stringify(g.model)); const t0 = process.hrtime.bigint(); const r = solver.Solve(m, undefined, true); const ms = Number(process.hrtime.bigint() - t0) / 1e6; const st = r._tableau.__gpuSpeculativeStats; const res = solver.buildSimplifiedResult(r).result; gpu.release(r._tableau); return [ms, res, st]; };
    for (let i = 0; i < 8; i++) run();
    const a = []; let last;
    for (let i = 0; i < 21; i++) { last = run(); a.push(last[0]); }
    if (last[1] !== g.result.result) { console.log("WRONG ANSWER", last[1], g.result.result); process.exit(1); }
    console.log(JSON.stringify({ model: name, lookahead: la, median_ms: med(a), min_ms: Math.min(...a), stats: last[2] }));
}
