// TEST/BENCH INFRASTRUCTURE: CPU side of bench.py's LP-relaxations/s figure (SURVEY.md 8d).  Times the reference itself
// (oracle/_ref) on config 4: solver.Solve(Monster_II) repeated for about <seconds> of wall time after one warm-up;
// one relaxation = one applyCuts = one Tableau.simplex() call (restore + addCutConstraints + simplex,
// branch-and-cut.ts:33-37), counted inside the branch-and-bound only (model parsing and presolve are outside the timed
// region: the clock runs from Tableau.solve() entry to its return).
//   node oracle/ref_relax_rate.js <golden fixture .json.gz> <seconds>
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const root = path.join(__dirname, "_ref", "src");
const solver = require(path.join(root, "solver.js")).default;
const Tableau = require(path.join(root, "tableau", "tableau.js")).default;

const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(process.argv[2])).toString());
const seconds = Number(process.argv[3] || 3);
let relaxations = 0, inSolve = 0n;
const P = Tableau.prototype;
const simplex = P.simplex, solve = P.solve;
P.simplex = function () { relaxations += 1; return simplex.call(this); };
P.solve = function () { const t0 = process.hrtime.bigint(); try { return solve.call(this); } finally { inSolve += process.hrtime.bigint() - t0; } };
solver.Solve(JSON.parse(JSON.stringify(g.model))); // warm-up (JIT)
relaxations = 0; inSolve = 0n;
const start = process.hrtime.bigint();
let solves = 0;
while (Number(process.hrtime.bigint() - start) / 1e9 < seconds) {
    solver.Solve(JSON.parse(JSON.stringify(g.model)));
    solves += 1;
}
const sec = Number(inSolve) / 1e9;
console.log(JSON.stringify({ solves, relaxations, seconds: sec, relaxations_per_sec: relaxations / sec, node: process.version }));
