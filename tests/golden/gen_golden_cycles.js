// Golden vectors for DETECTED cycles (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node tests/golden/gen_golden_cycles.js
// simplex.ts:78-93 / 305-320: with model.checkForCycles (the reference's default) every selected (leaving, entering) pair is
// appended to a history and checkForCycles (simplex.ts:415-440) is run on it; a hit ends the phase with feasible = false and
// three messages ("Cycle in phase N", "Start :s", "Length :l").  None of the reference's fixtures ever hits it, so these do:
//   * small LPs found by searching seeds of a degenerate random family for a hit in the REFERENCE itself:
//     - `deg_*`: no unrestricted variables (classical degenerate cycling under the largest-coefficient rule, lengths 6..11);
//     - `unr_*`: with unrestricted variables (an unrestricted variable re-entering with the other sign: length 2);
//   * `embedded_*`: the same LPs embedded in a LARGE tableau -- extra variables with a zero objective coefficient (reduced cost 0:
//     never a candidate) and extra constraints over those variables only (zero in every column that can enter: never in a ratio
//     test) -- so that the identical pivot sequence, the hit and the messages run through the chip-wide kernels (register-resident
//     headline / wide geometries, fused pipeline) with partial pricing on: the candidate columns all sit in pricing batch 0.
// Recorded per case: the model, every pivot, flags, model.messages, the final tableau hash (tests/golden/cycles/*.json.gz).
"use strict";
const path = require("path");
const { run, write } = require("./gen_golden.js");
const solver = require(path.join(__dirname, "..", "..", "oracle", "_ref", "src", "solver.js")).default;

function rng(seed) { let s = seed >>> 0; return () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; }; }
// the family of /tmp searches that found the hits (kept verbatim: the seeds below depend on every draw)
function buildDeg(seed, n, m, degFrac, minFrac, dens, cmax) {
    const r = rng(seed);
    const ri = (lo, hi) => lo + Math.floor(r() * (hi - lo + 1));
    const model = { optimize: "obj", opType: r() < 0.5 ? "max" : "min", constraints: {}, variables: {} };
    for (let i = 0; i < m; i++) {
        if (r() < minFrac) model.constraints["c" + i] = { min: r() < degFrac ? 0 : ri(1, 6) };
        else model.constraints["c" + i] = { max: r() < degFrac ? 0 : ri(1, 6) };
    }
    for (let j = 0; j < n; j++) {
        const v = { obj: ri(-4, 9) };
        for (let i = 0; i < m; i++) if (r() < dens) { const x = ri(-cmax, cmax); if (x !== 0) v["c" + i] = x; }
        model.variables["x" + j] = v;
    }
    return model;
}
function buildUnr(seed, n, m, degFrac, unrFrac) {
    const r = rng(seed);
    const ri = (lo, hi) => lo + Math.floor(r() * (hi - lo + 1));
    const model = { optimize: "obj", opType: "max", constraints: {}, variables: {} };
    for (let i = 0; i < m; i++) model.constraints["c" + i] = { max: r() < degFrac ? 0 : ri(1, 20) };
    for (let j = 0; j < n; j++) {
        const v = { obj: ri(-3, 12) };
        for (let i = 0; i < m; i++) if (r() < 0.7) { const x = ri(-9, 9); if (x !== 0) v["c" + i] = x; }
        model.variables["x" + j] = v;
    }
    model.unrestricted = {};
    for (let j = 0; j < n; j++) if (r() < unrFrac) model.unrestricted["x" + j] = 1;
    return model;
}
const FAM = [[0.9, 0.0, 0.8, 3], [0.8, 0.3, 0.7, 4], [1.0, 0.2, 0.9, 2], [0.7, 0.5, 0.6, 9]];
const degCase = (seed) => buildDeg(seed, 3 + (seed % 10), 3 + ((seed >> 2) % 8), ...FAM[seed % 4]);
const unrCase = (seed) => buildUnr(seed, 4 + (seed % 12), 3 + (seed % 9), 0.3, 0.2);

// the small LP inside a big one: `extraVars` variables f<j> with no objective entry and `extraCons` constraints over them only
function embed(model, extraVars, extraCons, seed) {
    const r = rng(seed);
    const big = JSON.parse(JSON.stringify(model));
    for (let i = 0; i < extraCons; i++) big.constraints["fc" + i] = { max: 100 + Math.floor(r() * 900) };
    for (let j = 0; j < extraVars; j++) {
        const v = {};
        for (let i = 0; i < extraCons; i++) v["fc" + i] = 1 + Math.floor(r() * 20);
        big.variables["f" + j] = v;
    }
    big.options = Object.assign({}, big.options || {}, { presolve: false });  // (it would fix the zero-cost variables)
    return big;
}

function record(name, model, lite) {
    const out = run(model, true, lite);
    out.model = lite ? null : model;
    out.messages = solver.lastSolvedModel.messages.slice();
    if (lite) { out.tableau.rows = out.tableau.cols = out.tableau.vals = null; out.tableau.variableIds = null; out.final.rhs = null; }
    write(path.join(__dirname, "cycles"), name, out);
    console.log(name, out.tableau.height + "x" + out.tableau.width, out.nPivots, out.pivotDigest, out.final.feasible, JSON.stringify(out.messages));
    return out;
}

const degSeeds = [35358, 137788, 178868, 233528, 292715, 347708, 398167];
const unrSeeds = [3, 15, 20, 46, 53, 97];
for (const s of degSeeds) record("deg_" + s, Object.assign(degCase(s), { options: { presolve: false } }), false);
for (const s of unrSeeds) record("unr_" + s, Object.assign(unrCase(s), { options: { presolve: false } }), false);
// large: the embedding is rebuilt at test time from the small model + (extraVars, extraCons, seed) by tests/test_cycle_goldens.py
const big = [
    { from: "deg", seed: 35358, vars: 2000, cons: 2000 },   // 2011 x 2012: the headline register-resident geometry (1024 lanes x 2 columns x 8 rows), partial pricing
    { from: "deg", seed: 35358, vars: 2040, cons: 2030 },   // 2041 x 2052: just past 2048 columns -> 512 lanes x 6 columns
    { from: "deg", seed: 292715, vars: 2040, cons: 2030 },
    { from: "deg", seed: 137788, vars: 2600, cons: 2500 },  // ~2511 x 2612: 512 lanes x 6 columns x 12 rows (phase 2 resident) / fused
    { from: "deg", seed: 178868, vars: 1500, cons: 3000 },  // ~3009 x 1512: the tall geometry / fused, one column tile
    { from: "unr", seed: 15, vars: 2040, cons: 2030 },      // unrestricted variables: the general resident build
    { from: "unr", seed: 46, vars: 1500, cons: 3000 },
];
for (const b of big) {
    const small = Object.assign(b.from === "deg" ? degCase(b.seed) : unrCase(b.seed), { options: { presolve: false } });
    const out = record("embedded_" + b.from + "_" + b.seed + "_" + b.vars + "x" + b.cons, embed(small, b.vars, b.cons, 777), true);
    out.embedding = null;
}
