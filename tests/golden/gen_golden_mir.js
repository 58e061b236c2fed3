// Golden vectors for MIR cuts (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node tests/golden/gen_golden_mir.js
//
// Runs oracle/_ref (the type-erased reference itself) with options.useMIRCuts = true (src/model.ts:354-356 ->
// applyMIRCuts in the three branch-and-bound services, src/tableau/cutting-strategies.ts:74-212) on the integer
// fixtures stored under tests/golden/{fixtures,synthetic}, with the default and the incremental service.  Every case
// runs in its own child process under a wall-clock limit (the MIR loop of the default service has no iteration
// bound, branch-and-cut.ts:38-52); cases that exceed it are left out and listed.  Recorded per case: pivot digest,
// relaxation count, per simplex() call [pivots phase 1, pivots phase 2, feasible, evaluation, height], final tableau
// hash, result.  Output: tests/golden/mir.json.gz.
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const crypto = require("crypto");
const child = require("child_process");

function num(x) {
    if (Number.isFinite(x)) return Object.is(x, -0) ? "-0" : x;
    return String(x);
}

function runCase(file, options) {
    const refRoot = path.join(__dirname, "..", "..", "oracle", "_ref", "src");
    const solver = require(path.join(refRoot, "solver.js")).default;
    const Tableau = require(path.join(refRoot, "tableau", "tableau.js")).default;
    const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, file))).toString());
    const rec = { n: 0, h: 2166136261 | 0, calls: [], cur: null, mirCuts: 0 };
    const P = Tableau.prototype;
    const orig = { pivot: P.pivot, phase1: P.phase1, phase2: P.phase2, simplex: P.simplex, applyMIRCuts: P.applyMIRCuts };
    P.pivot = function (r, c) {
        rec.n += 1;
        rec.h = Math.imul(rec.h ^ r, 16777619);
        rec.h = Math.imul(rec.h ^ c, 16777619);
        return orig.pivot.call(this, r, c);
    };
    P.phase1 = function () { const n = orig.phase1.call(this); if (rec.cur) rec.cur.p1 = n; return n; };
    P.phase2 = function () { const n = orig.phase2.call(this); if (rec.cur) rec.cur.p2 = n; return n; };
    P.applyMIRCuts = function () { const h = this.height; orig.applyMIRCuts.call(this); rec.mirCuts += this.height - h; };
    P.simplex = function () {
        const cur = { p1: 0, p2: -1 };
        rec.cur = cur;
        orig.simplex.call(this);
        rec.cur = null;
        rec.calls.push([cur.p1, cur.p2, this.feasible ? 1 : 0, num(this.evaluation), this.height]);
        return this;
    };
    const model = JSON.parse(JSON.stringify(g.model));
    model.options = options;
    const solution = solver.Solve(model, undefined, true);
    const t = solution._tableau;
    const result = solver.buildSimplifiedResult(solution);
    return {
        file, options, presolveFixed: g.presolve ? g.presolve.nFixed : 0,
        nPivots: rec.n, pivotDigest: (rec.h >>> 0).toString(16), calls: rec.calls, mirCuts: rec.mirCuts,
        iterations: t.branchAndCutIterations, isIntegral: !!t.__isIntegral,
        feasible: solution.feasible, bounded: solution.bounded, evaluation: num(solution.evaluation),
        matrixSha: t.width > 0 ? crypto.createHash("sha256").update(Buffer.from(t.matrix.buffer, t.matrix.byteOffset, t.width * t.height * 8)).digest("hex") : null,
        height: t.height,
        result: JSON.parse(JSON.stringify(result, (k, v) => (typeof v === "number" ? num(v) : v))),
        resultKeys: Object.keys(result),
    };
}

if (process.argv[2] === "--case") {
    process.stdout.write(JSON.stringify(runCase(process.argv[3], JSON.parse(process.argv[4]))));
    process.exit(0);
}

const policies = [{}, { useIncremental: true }, { useIncremental: true, nodeSelection: "depth-first", branching: "most-fractional" },
    { nodeSelection: "best-first", branching: "most-fractional" }];
const cases = [], skipped = [];
for (const dir of ["fixtures", "synthetic"]) {
    const d = path.join(__dirname, dir);
    for (const f of fs.readdirSync(d).filter((x) => x.endsWith(".json.gz")).sort()) {
        const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(d, f))).toString());
        if (!g.model || !g.tableau || g.tableau.integerVarIndexes.length === 0) continue;
        for (const pol of policies) {
            const options = Object.assign({}, g.model.options || {}, pol, { useMIRCuts: true });
            delete options.timeout; // wall-clock limits are not replayable
            const r = child.spawnSync(process.execPath, [__filename, "--case", dir + "/" + f, JSON.stringify(options)],
                { timeout: 30000, maxBuffer: 1 << 28, encoding: "utf8" });
            if (r.status !== 0 || !r.stdout) {
                skipped.push({ file: dir + "/" + f, options, why: r.error ? String(r.error.code || r.error) : (r.stderr || "").slice(-200) });
                console.log("SKIP", f, JSON.stringify(pol), r.error ? r.error.code : r.status);
                continue;
            }
            const out = JSON.parse(r.stdout);
            cases.push(out);
            console.log(f, JSON.stringify(pol), out.iterations, out.nPivots, out.mirCuts, out.pivotDigest, out.result.result, out.feasible);
        }
    }
}
fs.writeFileSync(path.join(__dirname, "mir.json.gz"), zlib.gzipSync(Buffer.from(JSON.stringify({ cases, skipped })), { level: 9 }));
console.log(cases.length, "cases,", skipped.length, "skipped");
