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
// reference results for the B&B strategy options BEFORE the binding is installed (these models are not in the goldens)
const strategyFiles = ["Knapsack_1", "Integer_Wood_Shop_Problem", "Monster_II", "Sudoku4x4", "Integer_Sports_Complex_Problem"];
const strategies = [{ nodeSelection: "best-first" }, { nodeSelection: "depth-first" }, { nodeSelection: "hybrid", branching: "pseudocost" },
    { branching: "most-fractional" }, { branching: "strong" }, { useIncremental: true }];
function loadGolden(d, f) { return JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(d, f))).toString()); }
const strategyBase = {};
const strategyDir = path.join(root, "tests", "golden", "fixtures");
if (!filter && dir.indexOf("fixtures") >= 0) {
    for (const f of strategyFiles) for (const v of strategies) {
        const m = JSON.parse(JSON.stringify(loadGolden(strategyDir, f + ".json.gz").model));
        m.options = Object.assign({}, m.options || {}, v);
        strategyBase[f + JSON.stringify(v)] = JSON.stringify(solver.Solve(m));
    }
}
// a dense LP of the reference's own generator for the size-policy check below, solved by the unpatched reference
const denseLpGolden = (() => {
    const gen = require(path.join(refSrc, "test-utils", "problem-generator.js"));
    const model = gen.generateResourceAllocation({ seed: 7, numVariables: 160, numConstraints: 120, density: 0.8 });
    const solution = solver.Solve(JSON.parse(JSON.stringify(model)), undefined, true);
    return { model, final: { feasible: solution.feasible }, result: { result: solver.buildSimplifiedResult(solution).result } };
})();
// the post-solve editing API (dynamic-modification.ts through Model.updateRightHandSide / updateCost /
// updateConstraintCoefficient / smallerThan / removeConstraint, then model.solve() again): same sequence on the unpatched
// reference first, then under the binding, where every edit brings the tableau home and the next solve uploads it again
function editScript(file) {
    const out = [];
    const snap = (sol) => JSON.stringify([sol.feasible, sol.bounded, num(sol.evaluation), sol.generateSolutionSet()]);
    solver.Solve(JSON.parse(JSON.stringify(loadGolden(strategyDir, file + ".json.gz").model)), undefined, true);
    const model = solver.lastSolvedModel;
    const c0 = model.constraints[0], c1 = model.constraints[model.constraints.length - 1];
    const v0 = model.variables[0], v1 = model.variables[model.variables.length - 1];
    model.updateRightHandSide(c0, c0.isUpperBound ? 3 : -3);
    out.push(snap(model.solve()));
    model.setCost(v1.cost + 1.5, v1);
    out.push(snap(model.solve()));
    model.updateConstraintCoefficient(c1, v0, 0.25);
    out.push(snap(model.solve()));
    const extra = model.smallerThan(1e6);
    extra.addTerm(1, v0);
    extra.addTerm(2, v1);
    out.push(snap(model.solve()));
    model.removeConstraint(extra);
    out.push(snap(model.solve()));
    return out;
}
// a Model INSTANCE handed to Solve() (main.ts:127-134) stays the caller's: solve, solve again, edit, solve again -- all through
// solver.Solve(instance) with the simplified result, i.e. the path that hands engines back when it built the Model itself
function instanceScript(file) {
    const def = JSON.parse(JSON.stringify(loadGolden(strategyDir, file + ".json.gz").model));
    const model = new solver.Model(undefined, undefined, solver.selectBranchAndCutService(def)).loadJson(def);
    const out = [];
    out.push(JSON.stringify(solver.Solve(model)));
    out.push(JSON.stringify(solver.Solve(model)));
    const c0 = model.constraints[0];
    model.updateRightHandSide(c0, c0.isUpperBound ? 2 : -2);
    out.push(JSON.stringify(solver.Solve(model)));
    out.push(JSON.stringify(model.solve().generateSolutionSet()));
    return out;
}
const editFiles = ["Berlin_Air_Lift_Problem", "Wiki_1", "Monster_Problem", "Shift_Work_Problem"];
const instanceBase = {};
if (!filter && dir.indexOf("fixtures") >= 0) for (const f of editFiles) instanceBase[f] = instanceScript(f);
const editBase = {};
if (!filter && dir.indexOf("fixtures") >= 0) for (const f of editFiles) editBase[f] = editScript(f);
let uninstall = gpu.install(Tableau, { SlackVariable, solver, minCells: 0, speculate: 0 }); // parity runs: EVERY tableau on the engine, the reference's own one-node-at-a-time services

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
// options.nodeSelection / options.branching (enhanced service) run over the same overridden methods and must reproduce
// the unpatched reference
let strategyOk = 0;
for (const key of Object.keys(strategyBase)) {
    const f = key.slice(0, key.indexOf("{"));
    const v = JSON.parse(key.slice(key.indexOf("{")));
    const m = JSON.parse(JSON.stringify(loadGolden(strategyDir, f + ".json.gz").model));
    m.options = Object.assign({}, m.options || {}, v);
    const r = JSON.stringify(solver.Solve(m));
    if (r === strategyBase[key]) strategyOk += 1;
    else { fail += 1; console.log("FAIL strategy", f, JSON.stringify(v)); }
}
// options.useIncremental: host/gpu-incremental-service.js over device checkpoints against the reference's own incremental
// service (tests/golden/incremental.json.gz): same pivots, same number of relaxations, same result
let incrementalOk = 0, checkpointsTaken = 0;
const incFile = path.join(root, "tests", "golden", "incremental.json.gz");
if (!filter && dir.indexOf("fixtures") >= 0 && fs.existsSync(incFile)) {
    const cases = JSON.parse(zlib.gunzipSync(fs.readFileSync(incFile)).toString());
    for (const c of cases) {
        if (c.iterations <= 1 && !(c.options.nodeSelection === "depth-first")) continue; // single-relaxation cases: a sample
        const g = loadGolden(path.join(root, "tests", "golden"), c.file);
        const m = JSON.parse(JSON.stringify(g.model));
        m.options = Object.assign({}, c.options);
        delete m.options.timeout; // wall-clock limits are not replayable; no golden run was ended by one
        const solution = solver.Solve(m, undefined, true);
        const res = solver.buildSimplifiedResult(solution);
        const got = {};
        for (const k of Object.keys(res)) got[k] = num(res[k]);
        const bad = [];
        if (JSON.stringify(Object.keys(res)) !== JSON.stringify(c.resultKeys)) bad.push("keys");
        if (JSON.stringify(got) !== JSON.stringify(c.result)) bad.push("values");
        const trace = gpu.pivotTrace(solution._tableau);
        if (!trace) bad.push("not on the engine");
        else {
            if (trace.length / 2 !== c.nPivots) bad.push("pivot count " + trace.length / 2 + " != " + c.nPivots);
            else if (digest(trace) !== c.pivotDigest) bad.push("pivot digest");
            if (solution._tableau.branchAndCutIterations !== c.iterations) bad.push("B&B iterations");
            checkpointsTaken += solution._tableau.__gpuCheckpoints || 0;
            gpu.release(solution._tableau);
        }
        if (bad.length) { fail += 1; console.log("FAIL incremental", c.file, JSON.stringify(c.options), bad.join("; ")); }
        else incrementalOk += 1;
    }
}
// options.useMIRCuts: Tableau.applyMIRCuts() built on the device, under the default, enhanced and incremental services,
// against the reference (tests/golden/mir.json.gz)
let mirOk = 0;
const mirFile = path.join(root, "tests", "golden", "mir.json.gz");
if (!filter && dir.indexOf("fixtures") >= 0 && fs.existsSync(mirFile)) {
    const doc = JSON.parse(zlib.gunzipSync(fs.readFileSync(mirFile)).toString());
    for (const c of doc.cases) {
        if (c.mirCuts === 0 || c.nPivots > 20000) continue;
        const g = loadGolden(path.join(root, "tests", "golden"), c.file);
        const m = JSON.parse(JSON.stringify(g.model));
        m.options = Object.assign({}, c.options);
        const solution = solver.Solve(m, undefined, true);
        const res = solver.buildSimplifiedResult(solution);
        const got = {};
        for (const k of Object.keys(res)) got[k] = num(res[k]);
        const bad = [];
        if (JSON.stringify(Object.keys(res)) !== JSON.stringify(c.resultKeys)) bad.push("keys");
        if (JSON.stringify(got) !== JSON.stringify(c.result)) bad.push("values");
        const trace = gpu.pivotTrace(solution._tableau);
        if (!trace) bad.push("not on the engine");
        else {
            if (trace.length / 2 !== c.nPivots) bad.push("pivot count " + trace.length / 2 + " != " + c.nPivots);
            else if (digest(trace) !== c.pivotDigest) bad.push("pivot digest");
            if (solution._tableau.branchAndCutIterations !== c.iterations) bad.push("B&B iterations");
            gpu.release(solution._tableau);
        }
        if (bad.length) { fail += 1; console.log("FAIL mir", c.file, JSON.stringify(c.options), bad.join("; ")); }
        else mirOk += 1;
    }
}
// install(..., { speculate: 16 }): the default policy with 16-node speculative batches (host/gpu-speculative-service.js)
// must return the sequential run's result object and relaxation count on every integer fixture
// -- and with the round-4 lookahead (children of cached nodes evaluated ahead of the walk): same results, same iteration counts,
//    and the lookahead really ran (some node of some tree was evaluated before the walk built it)
let speculativeOk = 0, lookaheadOk = 0, lookaheadNodes = 0;
for (const lookahead of [0, 16]) if (!filter && dir.indexOf("fixtures") >= 0) {
    uninstall();
    uninstall = gpu.install(Tableau, { SlackVariable, solver, speculate: 16, minCells: 0, lookahead });
    for (const f of fs.readdirSync(dir).filter((x) => x.endsWith(".json.gz")).sort()) {
        const g = loadGolden(dir, f);
        if (!g.model || !g.tableau || g.tableau.integerVarIndexes.length === 0) continue;
        const o = g.model.options || {};
        if (o.nodeSelection || o.branching || o.useMIRCuts || o.useIncremental) continue;
        const m = JSON.parse(JSON.stringify(g.model));
        if (m.options) delete m.options.timeout; // wall-clock limits are not replayable
        const solution = solver.Solve(m, undefined, true);
        const res = solver.buildSimplifiedResult(solution);
        const got = {};
        for (const k of Object.keys(res)) got[k] = num(res[k]);
        const bad = [];
        if (!(solution._tableau.branchAndCutService && solution._tableau.branchAndCutService.__gpuSpeculative)) bad.push("service not injected");
        if (JSON.stringify(Object.keys(res)) !== JSON.stringify(g.resultKeys)) bad.push("keys");
        if (JSON.stringify(got) !== JSON.stringify(g.result)) bad.push("values");
        if (solution._tableau.branchAndCutIterations !== g.final.branchAndCutIterations) bad.push("B&B iterations " + solution._tableau.branchAndCutIterations + " != " + g.final.branchAndCutIterations);
        const sp = solution._tableau.__gpuSpeculativeStats;
        if (lookahead && sp) lookaheadNodes += sp.lookaheadNodes;
        gpu.release(solution._tableau);
        if (bad.length) { fail += 1; console.log("FAIL speculative", lookahead ? "(lookahead)" : "", f, bad.join("; ")); }
        else if (lookahead) lookaheadOk += 1;
        else speculativeOk += 1;
    }
}
// random models (tests/golden/fuzz_*.jsonl.gz: the reference's generators under seven service policies; soft constraints,
// equalities, ranges, unrestricted variables) through the reference host + binding, INCLUDING the ones the reference's
// presolve touches: result object, pivot count / digest, relaxation count
let fuzzOk = 0;
if (!filter && dir.indexOf("fixtures") >= 0) {
    uninstall();
    uninstall = gpu.install(Tableau, { SlackVariable, solver, minCells: 0, speculate: 0 });
    for (const name of ["fuzz_services.jsonl.gz", "fuzz_soft.jsonl.gz"]) {
        const file = path.join(root, "tests", "golden", name);
        if (!fs.existsSync(file)) continue;
        for (const line of zlib.gunzipSync(fs.readFileSync(file)).toString().split("\n")) {
            if (!line.startsWith("{")) continue;
            const c = JSON.parse(line);
            const solution = solver.Solve(JSON.parse(JSON.stringify(c.model)), undefined, true);
            const res = solver.buildSimplifiedResult(solution);
            const got = {};
            for (const k of Object.keys(res)) got[k] = num(res[k]);
            const bad = [];
            if (JSON.stringify(Object.keys(res)) !== JSON.stringify(c.keys)) bad.push("keys");
            if (JSON.stringify(got) !== JSON.stringify(c.result)) bad.push("values");
            const trace = solution._tableau ? gpu.pivotTrace(solution._tableau) : null;
            if (trace) {
                if (trace.length / 2 !== c.nPivots) bad.push("pivot count " + trace.length / 2 + " != " + c.nPivots);
                else if (digest(trace) !== c.digest) bad.push("pivot digest");
                if (c.iter !== null && solution._tableau.branchAndCutIterations !== undefined && solution._tableau.branchAndCutIterations !== c.iter) bad.push("B&B iterations");
                gpu.release(solution._tableau);
            }
            if (bad.length) { fail += 1; console.log("FAIL fuzz", name, c.gen, c.seed, JSON.stringify(c.model.options || {}), bad.join("; ")); }
            else fuzzOk += 1;
        }
    }
}
// DETECTED cycles (tests/golden/cycles: reference runs that end in "Cycle in phase N"): same result object and the same three
// model.messages ("Cycle in phase N", "Start :s", "Length :l" -- simplex.ts:86-88 / 313-315) through the binding
let cycleOk = 0;
if (!filter && dir.indexOf("fixtures") >= 0) {
    const cdir = path.join(root, "tests", "golden", "cycles");
    for (const f of (fs.existsSync(cdir) ? fs.readdirSync(cdir) : []).filter((x) => x.endsWith(".json.gz") && !x.startsWith("embedded")).sort()) {
        const g = loadGolden(cdir, f);
        if (!g.model) continue;  // (recorded without its model: rebuilt by tests/test_cycle_goldens.py, not here)
        const solution = solver.Solve(JSON.parse(JSON.stringify(g.model)), undefined, true);
        const res = solver.buildSimplifiedResult(solution);
        const got = {};
        for (const k of Object.keys(res)) got[k] = num(res[k]);
        const msgs = solver.lastSolvedModel.messages;
        const trace = gpu.pivotTrace(solution._tableau);
        const bad = [];
        if (JSON.stringify(got) !== JSON.stringify(g.result)) bad.push("result");
        if (JSON.stringify(msgs) !== JSON.stringify(g.messages)) bad.push("messages " + JSON.stringify(msgs) + " != " + JSON.stringify(g.messages));
        if (!trace || digest(trace) !== g.pivotDigest) bad.push("pivot digest");
        gpu.release(solution._tableau);
        if (bad.length) { fail += 1; console.log("FAIL cycle", f, bad.join("; ")); } else cycleOk += 1;
    }
}
// the editing API under the binding (see editScript above)
let editOk = 0;
for (const f of Object.keys(editBase)) {
    const got = editScript(f);
    for (let i = 0; i < got.length; i++) {
        if (got[i] === editBase[f][i]) editOk += 1;
        else { fail += 1; console.log("FAIL edit", f, "step", i); }
    }
    gpu.bringHome(solver.lastSolvedModel.tableau);
}
// Solve(modelInstance) twice, edited in between (ADVICE r02: the engine of a caller's instance must not be released)
let instanceOk = 0;
for (const f of Object.keys(instanceBase)) {
    const got = instanceScript(f);
    for (let i = 0; i < got.length; i++) {
        if (got[i] === instanceBase[f][i]) instanceOk += 1;
        else { fail += 1; console.log("FAIL model instance re-solve", f, "step", i); }
    }
    gpu.bringHome(solver.lastSolvedModel.tableau);
}
// a tableau whose engine went back to the pool when Solve() returned must refuse further solves (never a stale host copy)
let releasedOk = 0;
if (!filter && dir.indexOf("fixtures") >= 0) {
    solver.Solve(JSON.parse(JSON.stringify(loadGolden(strategyDir, "Monster_Problem.json.gz").model)));
    try {
        solver.lastSolvedModel.tableau.simplex();
        fail += 1; console.log("FAIL released tableau solved again");
    } catch (e) {
        if (/released/.test(String(e.message))) releasedOk += 1; else { fail += 1; console.log("FAIL released:", e.message); }
    }
}
// install(..., { speculate: 16, devices: [0, 0, 0, 0] }): the speculative batches split over a device pool (jslp_pool_*;
// here four engines on device 0 -- "virtual devices" -- each with its own stream and host thread inside the library)
// -- with the compact read-back (round 4: jslp_pool_relax_batch_watched, the default) and with the full one (fullReadBack: true)
let poolOk = 0, poolFullOk = 0;
for (const fullReadBack of [false, true]) if (!filter && dir.indexOf("fixtures") >= 0) {
    uninstall();
    uninstall = gpu.install(Tableau, { SlackVariable, solver, speculate: 16, devices: [0, 0, 0, 0], minCells: 0, fullReadBack });
    for (const f of fs.readdirSync(dir).filter((x) => x.endsWith(".json.gz")).sort()) {
        const g = loadGolden(dir, f);
        if (!g.model || !g.tableau || g.tableau.integerVarIndexes.length === 0) continue;
        const o = g.model.options || {};
        if (o.nodeSelection || o.branching || o.useMIRCuts || o.useIncremental) continue;
        const m = JSON.parse(JSON.stringify(g.model));
        if (m.options) delete m.options.timeout;
        const solution = solver.Solve(m, undefined, true);
        const res = solver.buildSimplifiedResult(solution);
        const got = {};
        for (const k of Object.keys(res)) got[k] = num(res[k]);
        const bad = [];
        if (JSON.stringify(Object.keys(res)) !== JSON.stringify(g.resultKeys)) bad.push("keys");
        if (JSON.stringify(got) !== JSON.stringify(g.result)) bad.push("values");
        if (solution._tableau.branchAndCutIterations !== g.final.branchAndCutIterations) bad.push("B&B iterations");
        const usedPool = !!(solution._tableau.__gpu && solution._tableau.__gpu.pool);
        gpu.release(solution._tableau);
        if (bad.length) { fail += 1; console.log("FAIL pool", fullReadBack ? "(full read-back)" : "(compact)", f, bad.join("; ")); }
        else if (fullReadBack) poolFullOk += usedPool ? 1 : 0;
        else poolOk += usedPool ? 1 : 0;
    }
}
// compact read-back of a batch (relaxBatchWatched) against the full one, node by node
// -- on one engine, and over the device pool (poolRelaxBatchWatched against poolRelaxBatch)
let watchedOk = 0, poolWatchedOk = 0, packedOk = 0;
for (const devices of [null, [0, 0, 0, 0]]) if (!filter && dir.indexOf("fixtures") >= 0) {
    uninstall();
    uninstall = gpu.install(Tableau, devices ? { SlackVariable, solver, speculate: 16, minCells: 0, devices } : { SlackVariable, solver, speculate: 16, minCells: 0 });
    const g = loadGolden(dir, "Monster_II.json.gz");
    const m = JSON.parse(JSON.stringify(g.model));
    if (m.options) delete m.options.timeout;
    const solution = solver.Solve(m, undefined, true);
    const t = solution._tableau;
    const ints = t.model.integerVariables.map((v) => v.index);
    const lists = [[]];
    for (let i = 0; i < 24; i++) lists.push([{ type: i % 2 ? "min" : "max", varIndex: ints[i], value: i % 3 }, { type: "max", varIndex: ints[(i * 7) % ints.length], value: 1 }]);
    const full = gpu.relaxBatch(t, lists), compact = gpu.relaxBatchWatched(t, lists);
    for (let i = 0; i < lists.length; i++) {
        const rowOf = new Map();
        for (let r = 1; r < full[i].rows.length; r++) rowOf.set(full[i].rows[r], r);
        let same = full[i].res.feasible === compact[i].res.feasible && full[i].res.height === compact[i].res.height;
        for (let k = 0; k < ints.length && same; k++) {
            const r = rowOf.has(ints[k]) ? rowOf.get(ints[k]) : -1;
            same = compact[i].rows[k] === r && Object.is(compact[i].values[k], r > 0 ? full[i].rhs[r] : 0);
        }
        if (!same) { fail += 1; console.log("FAIL watched batch, node", i, devices ? "(pool)" : ""); }
        else if (devices) poolWatchedOk += 1;
        else watchedOk += 1;
    }
    if (!devices) {
        // round 5: the batch results come back PACKED (two typed arrays, unpacked in gpu-tableau.js); the addon still builds the array of
        // objects itself when the two arrays are not passed -- both forms must be the same objects, key for key
        const addon = require(path.join(root, "addon", "jslp_napi.node"));
        const offsets = new Int32Array(lists.length + 1);
        let total = 0;
        for (let i = 0; i < lists.length; i++) { total += lists[i].length; offsets[i + 1] = total; }
        const type = new Int8Array(total), varIndex = new Int32Array(total), value = new Float64Array(total);
        for (let i = 0, k = 0; i < lists.length; i++) for (const c of lists[i]) { type[k] = c.type === "min" ? 0 : 1; varIndex[k] = c.varIndex; value[k] = c.value; k++; }
        const legacy = addon.relaxBatchWatched(t.__gpu.h, offsets, type, varIndex, value, t.model.checkForCycles === true,
            new Int32Array(lists.length * ints.length), new Float64Array(lists.length * ints.length));
        for (let i = 0; i < lists.length; i++) {
            if (JSON.stringify(legacy[i]) === JSON.stringify(compact[i].res) && Object.keys(legacy[i]).join() === Object.keys(full[i].res).join()) packedOk += 1;
            else { fail += 1; console.log("FAIL packed batch results, node", i, JSON.stringify(legacy[i]), JSON.stringify(compact[i].res)); }
        }
    }
    gpu.release(t);
}
// the DEFAULT size policy (no minCells given): small tableaus stay on the reference's own path -- also under the injected services
let policyOk = 0;
if (!filter && dir.indexOf("fixtures") >= 0) {
    uninstall();
    // (nothing but the defaults: size policy + 16-node speculative batches -- except the CPU time budget of deferred LPs, which is made
    //  generous here so that the ROUTING is what is checked, whatever the speed of the box; the budget itself is exercised below)
    uninstall = gpu.install(Tableau, { SlackVariable, solver, cpuBudgetMs: 60000 });
    // (round 5: LPs go by their structural non-zeros -- Monster LP, 345 k cells but 3.4 k non-zeros, stays on the reference's own path;
    //  a dense generated LP of 19 k cells / ~15 k non-zeros goes to the engine)
    for (const [f, expectOnEngine] of [["Knapsack_1", false], ["LargeFarmMIP", false], ["Monster_II", true], ["Monster_Problem", false], ["@denseLP", true]]) {
        for (const extra of [{}, { useIncremental: true }]) {
            const g = f === "@denseLP" ? denseLpGolden : loadGolden(dir, f + ".json.gz");
            const m = JSON.parse(JSON.stringify(g.model));
            m.options = Object.assign({}, m.options || {}, extra);
            delete m.options.timeout;
            const solution = solver.Solve(m, undefined, true);
            const onEngine = gpu.pivotTrace(solution._tableau) !== null;
            gpu.release(solution._tableau);
            const sameResult = extra.useIncremental ? solution.feasible === g.final.feasible
                : JSON.stringify(solver.buildSimplifiedResult(solution).result) === JSON.stringify(g.result.result);
            if (onEngine === expectOnEngine && sameResult) policyOk += 1;
            else { fail += 1; console.log("FAIL size policy", f, JSON.stringify(extra), "on engine:", onEngine); }
        }
    }
}
// a deferred LP whose CPU budget runs out (made to: 0.001 ms): abandoned on the reference's path, rebuilt in the engine's pinned buffer and
// solved there from the first pivot -- the reference's result object and the reference's pivot digest; with a generous budget the same LP
// finishes on the CPU and no engine is created
let deferOk = 0;
if (!filter && dir.indexOf("fixtures") >= 0) {
    for (const [budget, expectOnEngine] of [[0.001, true], [60000, false]]) {
        uninstall();
        uninstall = gpu.install(Tableau, { SlackVariable, solver, cpuBudgetMs: budget });
        const g = loadGolden(dir, "Monster_Problem.json.gz");
        const before = gpu.stats.deferredToEngine;
        const solution = solver.Solve(JSON.parse(JSON.stringify(g.model)), undefined, true);
        const trace = gpu.pivotTrace(solution._tableau);
        const onEngine = trace !== null;
        let digestOk = true;
        if (onEngine) {
            let h = 2166136261;
            for (let i = 0; i < trace.length; i++) h = Math.imul(h ^ trace[i], 16777619);
            digestOk = (h >>> 0).toString(16) === g.pivotDigest && trace.length === 2 * g.nPivots;
        }
        gpu.release(solution._tableau);
        const sameResult = JSON.stringify(solver.buildSimplifiedResult(solution).result) === JSON.stringify(g.result.result);
        if (onEngine === expectOnEngine && sameResult && digestOk && (gpu.stats.deferredToEngine - before === (expectOnEngine ? 1 : 0))) deferOk += 1;
        else { fail += 1; console.log("FAIL deferred LP, budget", budget, "on engine:", onEngine, "same result:", sameResult, "digest:", digestOk); }
    }
}
console.log(JSON.stringify({ backend, pass, fail, solved_on_engine: onGpu, defer_ok: deferOk, strategy_variants_ok: strategyOk,
    incremental_ok: incrementalOk, device_checkpoints: checkpointsTaken, mir_ok: mirOk, speculative_ok: speculativeOk, lookahead_ok: lookaheadOk, lookahead_ran: lookaheadNodes > 0,
    size_policy_ok: policyOk, fuzz_ok: fuzzOk, edit_ok: editOk, released_ok: releasedOk, instance_ok: instanceOk, cycle_ok: cycleOk, pool_ok: poolOk, pool_full_ok: poolFullOk, watched_ok: watchedOk, pool_watched_ok: poolWatchedOk, packed_ok: packedOk }));
process.exit(fail === 0 && pass > 0 ? 0 : 1);
