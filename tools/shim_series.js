// per-Solve wall times of the reference host + binding over many consecutive solves (JIT / GC warm-up curve)
//   node tools/shim_series.js <fixture> [speculate] [count]
"use strict";
const fs = require("fs"), path = require("path"), zlib = require("zlib");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const mode = process.argv[3] || "1";
if (mode !== "cpu") {
    const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
    const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
    const gpu = require(path.join(root, "host/gpu-tableau.js"));
    gpu.loadEngine(process.env.JSLP_HIP_LIBRARY ? { library: path.resolve(process.env.JSLP_HIP_LIBRARY) } : {});
    gpu.install(T, { SlackVariable, solver, speculate: Number(mode) });
}
const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root, "tests/golden/fixtures", process.argv[2] + ".json.gz"))).toString());
const n = Number(process.argv[4] || 20);
const out = [];
for (let i = 0; i < n; i++) {
    const m = JSON.parse(JSON.stringify(g.model));
    const t0 = process.hrtime.bigint();
    solver.Solve(m);
    out.push((Number(process.hrtime.bigint() - t0) / 1e6).toFixed(1));
}
console.log(process.argv[2], "mode", mode, out.join(" "));
