// Where does the engine start to pay for an LP -- by AREA or by WORK?  End-to-end solver.Solve(model) of the reference's own
// generator over sizes x densities: the unpatched reference (CPU) against the reference host + binding with everything on the
// engine (minCells = 0), with the structural non-zero count host/gpu-tableau.js's default policy routes on, and the time the
// engine path spends inside each binding call (addon.timings()).  The default of install()'s minNnz comes from this table.
//   node tools/policy_sweep.js [engine library]      -> markdown (profiles/r05_policy_sweep.md)
"use strict";
const path = require("path");
const root = path.join(__dirname, "..");
const solver = require(path.join(root, "oracle/_ref/src/solver.js")).default;
const T = require(path.join(root, "oracle/_ref/src/tableau/tableau.js")).default;
const { SlackVariable } = require(path.join(root, "oracle/_ref/src/expressions.js"));
const gen = require(path.join(root, "oracle/_ref/src/test-utils/problem-generator.js"));
const gpu = require(path.join(root, "host/gpu-tableau.js"));
const fs = require("fs"), zlib = require("zlib");

const cases = [];
const sizes = process.env.JSLP_SWEEP_SIZES ? JSON.parse(process.env.JSLP_SWEEP_SIZES)
    : [[60, 45], [100, 75], [140, 105], [200, 150], [300, 225], [450, 340]];
const densities = process.env.JSLP_SWEEP_DENSITIES ? JSON.parse(process.env.JSLP_SWEEP_DENSITIES) : [0.8, 0.2, 0.05];
function nnzOf(model) {  // what structuralNnz() counts from the Model: one per term, right-hand side and cost
    let n = Object.keys(model.constraints).length;
    for (const v of Object.values(model.variables)) for (const k of Object.keys(v)) if (v[k] !== 0) n += 1;
    return n;
}
for (const [n, m] of sizes) for (const d of densities) {
    if (n * m * d < 150) continue;  // (nearly empty models: the generator leaves variables without a single coefficient)
    if (d < 0.5 && n * m > (Number(process.env.JSLP_SWEEP_MAX_SPARSE_CELLS) || 70000)) continue;  // (the CPU leg of the sparser big ones takes minutes per solve)
    const model = gen.generateResourceAllocation({ seed: 7, numVariables: n, numConstraints: m, density: d });
    cases.push({ label: "LP " + n + " x " + m + " @ " + d, cells: (n + 1) * (m + 1), nnz: nnzOf(model), model });
}
for (const f of ["Monster_Problem"]) {  // BASELINE config 2
    const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root, "tests/golden/fixtures", f + ".json.gz"))).toString());
    cases.push({ label: "fixture " + f, cells: g.tableau.height * g.tableau.width, nnz: nnzOf(g.model), model: g.model });
}
const med = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
function time(model, reps) {
    // bounded: a sparse-ish 300 x 225 LP takes the reference 20-40 s PER SOLVE on the CPU (thousands of pivots on a tableau that fills in).
    // A first solve that takes more than half a second is the sample; otherwise two more warm-ups, then the median of <= reps (<= ~2 s).
    const t00 = process.hrtime.bigint();
    let r = solver.Solve(JSON.parse(JSON.stringify(model)));
    const first = Number(process.hrtime.bigint() - t00) / 1e6;
    if (first > 500) return [first, r.result, r.feasible];
    for (let i = 0; i < 2; i++) solver.Solve(JSON.parse(JSON.stringify(model)));
    const a = [];
    let spent = 0;
    for (let i = 0; i < reps; i++) {
        const mm = JSON.parse(JSON.stringify(model));
        const t0 = process.hrtime.bigint();
        r = solver.Solve(mm);
        a.push(Number(process.hrtime.bigint() - t0) / 1e6);
        spent += a[a.length - 1];
        if (a.length >= 3 && spent > 2000) break;
    }
    return [med(a), r.result, r.feasible];
}
for (const c of cases) { const [ms, res, feas] = time(c.model, 9); c.cpu = ms; c.res = res; c.feas = feas; process.stderr.write("cpu " + c.label + " " + ms.toFixed(2) + " ms\n"); }
gpu.loadEngine(process.argv[2] ? { library: path.resolve(process.argv[2]) } : {});
const addon = require(path.join(root, "addon/jslp_napi.node"));
let uninstall = gpu.install(T, { SlackVariable, solver, minCells: 0, speculate: 0 });
for (const c of cases) {
    time(c.model, 3);
    addon.timings(true);
    const [ms, res, feas] = time(c.model, 9);
    const t = addon.timings(true);
    const runs = (t.create || [0, 1])[1];
    const per = (names) => names.reduce((s, k) => s + (t[k] ? t[k][0] : 0), 0) / Math.max(1, runs);
    const phases = [per(["create", "hostMatrix"]), per(["upload", "setOptionalObjectives"]), per(["simplex"]), per(["readRhs", "getOptionalObjectives"]), per(["detach", "destroy"])];
    c.engine = ms;
    c.same = res === c.res && feas === c.feas;
    c.phases = phases;
}
// what the DEFAULT policy does with each (install() without minCells): where it ends up, whether it got there through the CPU time budget of
// a deferred LP (host/gpu-tableau.js `deferrable`), and what the Solve() costs that way
uninstall();
uninstall = gpu.install(T, { SlackVariable, solver });
for (const c of cases) {
    const before = gpu.stats.deferredToEngine;
    const solution = solver.Solve(JSON.parse(JSON.stringify(c.model)), undefined, true);
    c.onEngine = gpu.pivotTrace(solution._tableau) !== null;
    c.viaBudget = gpu.stats.deferredToEngine > before;
    c.nnz = gpu.structuralNnz(solution._tableau);  // (exactly what the policy counted)
    gpu.release(solution._tableau);
    const [ms, res, feas] = time(c.model, 9);
    c.defMs = ms;
    c.same = c.same && res === c.res && feas === c.feas;
}
console.log("| model | cells | structural nnz | reference on CPU (ms) | everything on the engine, minCells 0 (ms) | engine / CPU | inside the binding, engine run (ms: create+pin / upload / simplex / read-back / release) | DEFAULT policy: where | DEFAULT policy (ms) | default / best of the two | same result |");
console.log("|---|---|---|---|---|---|---|---|---|---|---|");
for (const c of cases) {
    const best = Math.min(c.cpu, c.engine);
    console.log("| " + c.label + " | " + c.cells + " | " + c.nnz + " | " + c.cpu.toFixed(2) + " | " + c.engine.toFixed(2) + " | " + (c.engine / c.cpu).toFixed(2) + " | " +
        c.phases.map((x) => x.toFixed(3)).join(" / ") + " | " + (c.onEngine ? (c.viaBudget ? "CPU for the budget, then engine" : "engine") : "CPU") + " | " + c.defMs.toFixed(2) + " | " +
        (c.defMs / best).toFixed(2) + " | " + c.same + " |");
}
