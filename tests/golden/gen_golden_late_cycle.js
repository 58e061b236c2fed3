// Golden vector for a cycle DETECTED LATE (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node --max-old-space-size=8000 tests/golden/gen_golden_late_cycle.js
// The detected cycles of gen_golden_cycles.js all sit within the first few dozen pivots.  Here a small cycling LP (deg_35358, a
// "min" model) follows an ACTIVE dense block: generateResourceAllocation(seed 4242, 1500 x 1500) rewritten as a minimisation
// (cost = -profit: the same tableau), whose 1500 columns fill pricing batches 0..29 exactly (batch = 50 columns at 1511), so the
// reference solves that block first -- thousands of pivots -- and only then reaches batch 30, where the small LP cycles: the hit
// lands far beyond the part of the history the register-resident kernel keeps in LDS (simplex.ts:305-320, 415-440).
// Recorded like the embedded cases (model rebuilt at test time: tests/test_cycle_goldens.py).
"use strict";
const path = require("path");
const { run, write, gen } = require("./gen_golden.js");
const solver = require(path.join(__dirname, "..", "..", "oracle", "_ref", "src", "solver.js")).default;
const zlib = require("zlib"), fs = require("fs");

function lateCycle(small, n, seed) {
    const ra = gen.generateResourceAllocation({ seed, numVariables: n, numConstraints: n, density: 1.0 });
    const big = { optimize: "obj", opType: "min", constraints: {}, variables: {} };
    for (const k of Object.keys(ra.constraints)) big.constraints[k] = ra.constraints[k];
    for (const k of Object.keys(ra.variables)) {
        const v = Object.assign({}, ra.variables[k]);
        v.obj = -v[ra.optimize];
        delete v[ra.optimize];
        big.variables[k] = v;
    }
    for (const k of Object.keys(small.constraints)) big.constraints["z_" + k] = small.constraints[k];
    for (const k of Object.keys(small.variables)) {
        const v = {};
        for (const a of Object.keys(small.variables[k])) v[a === small.optimize ? "obj" : "z_" + a] = small.variables[k][a];
        big.variables["z_" + k] = v;
    }
    big.options = { presolve: false };
    return big;
}

const small = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, "cycles", "deg_35358.json.gz"))).toString()).model;
const model = lateCycle(small, 1500, 4242);
const out = run(model, true, true);
out.model = null;
out.messages = solver.lastSolvedModel.messages.slice();
out.tableau.rows = out.tableau.cols = out.tableau.vals = null; out.tableau.variableIds = null; out.final.rhs = null;
out.meta = { kind: "late_cycle", small: "deg_35358", n: 1500, seed: 4242 };
write(path.join(__dirname, "cycles"), "late_deg_35358_after_RA_1500", out);
console.log(out.tableau.height + "x" + out.tableau.width, out.nPivots, out.pivotDigest, out.final.feasible, JSON.stringify(out.messages), out.refWallMs);
