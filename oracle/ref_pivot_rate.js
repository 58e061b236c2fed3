// TEST/BENCH INFRASTRUCTURE: cpu_baseline leg of bench.py.  Times the reference itself (oracle/_ref) on the
// benchmark instance: generateResourceAllocation({seed:12345, numVariables:n, numConstraints:n, density:1}),
// options.exitOnCycles=false, single thread.  Stops after `sample` pivots (a bounded sample of the same
// workload) and reports pivots/sec over [simplex() entry, last sampled pivot].
//   node oracle/ref_pivot_rate.js <n> <sample>
"use strict";
const path = require("path");
const root = path.join(__dirname, "_ref", "src");
const solver = require(path.join(root, "solver.js")).default;
const Tableau = require(path.join(root, "tableau", "tableau.js")).default;
const gen = require(path.join(root, "test-utils", "problem-generator.js"));

const n = Number(process.argv[2] || 2000);
const sample = Number(process.argv[3] || 1000);
const model = gen.generateResourceAllocation({ seed: 12345, numVariables: n, numConstraints: n, density: 1.0 });
model.options = { exitOnCycles: false };

let t0 = 0n, t1 = 0n, count = 0;
const P = Tableau.prototype;
const pivot = P.pivot, simplex = P.simplex;
class Stop extends Error {}
P.simplex = function () { t0 = process.hrtime.bigint(); return simplex.call(this); };
P.pivot = function (r, c) {
    pivot.call(this, r, c);
    count += 1;
    t1 = process.hrtime.bigint();
    if (count >= sample) throw new Stop();
};
try {
    solver.Solve(model);
} catch (e) {
    if (!(e instanceof Stop)) throw e;
}
const sec = Number(t1 - t0) / 1e9;
console.log(JSON.stringify({ n, pivots: count, seconds: sec, pivots_per_sec: count / sec, node: process.version }));
