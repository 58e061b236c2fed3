// Where does the engine start to pay?  End-to-end solver.Solve(model) of the reference's own generators at growing sizes:
// the unpatched reference (CPU) against the reference host + binding (engine for every tableau: minCells = 0).
//   node tools/mincells_sweep.js [engine library]      -> markdown table (the default of install()'s minCells comes from it)
"use strict";
const path = require("path");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
const gen = require(path.join(root, "oracle/_ref/src/test-utils/problem-generator.js"));
const gpu = require(path.join(root, "host/gpu-tableau.js"));

const cases = [];
const sizes = process.env.JSLP_SWEEP_SIZES ? JSON.parse(process.env.JSLP_SWEEP_SIZES) : [[20, 15], [40, 30], [80, 60], [120, 90], [160, 120], [240, 180], [320, 240], [480, 360], [640, 480]];
for (const [n, m] of sizes) {
    cases.push({ kind: "LP", n, m, model: gen.generateResourceAllocation({ seed: 7, numVariables: n, numConstraints: m, density: 0.8 }) });
    if (n <= (Number(process.env.JSLP_SWEEP_MAX_MIP) || 160)) {  // bigger random MIPs spend minutes in the CPU tree: not what this sweep is after
        const mip = gen.generateKnapsack({ seed: 7, numVariables: n });
        cases.push({ kind: "MIP (knapsack)", n, m: 1, model: mip });
    }
}
function time(model) {
    for (let i = 0; i < 4; i++) solver.Solve(JSON.parse(JSON.stringify(model)));
    const a = [];
    let r;
    for (let i = 0; i < 7; i++) {
        const mm = JSON.parse(JSON.stringify(model));
        const t0 = process.hrtime.bigint();
        r = solver.Solve(mm);
        a.push(Number(process.hrtime.bigint() - t0) / 1e6);
    }
    a.sort((x, y) => x - y);
    return [a[3], r.result, r.feasible];
}
for (const c of cases) { const [ms, res, feas] = time(c.model); c.cpu = ms; c.res = res; c.feas = feas; }
gpu.loadEngine(process.argv[2] ? { library: path.resolve(process.argv[2]) } : {});
gpu.install(T, { SlackVariable, solver, minCells: 0, speculate: 0 });
console.log("| model | tableau cells | reference on CPU (ms) | reference host + engine (ms) | engine / CPU | same result |");
console.log("|---|---|---|---|---|---|");
for (const c of cases) {
    const [ms, res, feas] = time(c.model);
    const cells = (c.m + 1 + (c.kind === "LP" ? 0 : c.n)) * (c.n + 1);  // knapsack: one `x <= 1` row per binary variable
    console.log("| " + c.kind + " " + c.n + " x " + c.m + " | " + cells + " | " + c.cpu.toFixed(2) + " | " + ms.toFixed(2) + " | " + (ms / c.cpu).toFixed(2) + " | " +
        (res === c.res && feas === c.feas) + " |");
}
