// Golden vectors for the incremental branch-and-bound service (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node tests/golden/gen_golden_incremental.js
//
// Runs oracle/_ref (the type-erased reference itself) with options.useIncremental = true
// (src/main.ts:62-72 -> src/tableau/incremental-branch-and-cut.ts) on every integer fixture already stored under
// tests/golden/fixtures and tests/golden/synthetic, for each node-selection / branching policy the service
// has, and records what a re-implementation must reproduce: every pivot (FNV-1a digest), the number of
// relaxations, per relaxation its pivots per phase / flags / evaluation, the final tableau hash and the Solve()
// result.  Output: tests/golden/incremental.json.gz (the models themselves are in the fixture files).
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const crypto = require("crypto");

const refRoot = path.join(__dirname, "..", "..", "oracle", "_ref", "src");
const solver = require(path.join(refRoot, "solver.js")).default;
const Tableau = require(path.join(refRoot, "tableau", "tableau.js")).default;

function num(x) {
    if (Number.isFinite(x)) return Object.is(x, -0) ? "-0" : x;
    return String(x);
}
function sha(buf) {
    return crypto.createHash("sha256").update(buf).digest("hex");
}

let rec = null;
const P = Tableau.prototype;
const orig = { pivot: P.pivot, phase1: P.phase1, phase2: P.phase2, simplex: P.simplex };
P.pivot = function (r, c) {
    if (rec) {
        rec.n += 1;
        rec.h = Math.imul(rec.h ^ r, 16777619);
        rec.h = Math.imul(rec.h ^ c, 16777619);
    }
    return orig.pivot.call(this, r, c);
};
P.phase1 = function () {
    const n = orig.phase1.call(this);
    if (rec && rec.cur) rec.cur.p1 = n;
    return n;
};
P.phase2 = function () {
    const n = orig.phase2.call(this);
    if (rec && rec.cur) rec.cur.p2 = n;
    return n;
};
P.simplex = function () {
    if (!rec) return orig.simplex.call(this);
    const cur = { p1: 0, p2: -1 };
    rec.cur = cur;
    orig.simplex.call(this);
    rec.cur = null;
    rec.calls.push([cur.p1, cur.p2, this.feasible ? 1 : 0, num(this.evaluation), this.height]);
    return this;
};

function run(model) {
    rec = { n: 0, h: 2166136261 | 0, calls: [], cur: null };
    const solution = solver.Solve(JSON.parse(JSON.stringify(model)), undefined, true);
    const r = rec;
    rec = null;
    const t = solution._tableau;
    const result = solver.buildSimplifiedResult(solution);
    return {
        nPivots: r.n,
        pivotDigest: (r.h >>> 0).toString(16),
        calls: r.calls,
        iterations: t.branchAndCutIterations,
        isIntegral: !!t.__isIntegral,
        feasible: solution.feasible,
        bounded: solution.bounded,
        evaluation: num(solution.evaluation),
        matrixSha: t.width > 0 ? sha(Buffer.from(t.matrix.buffer, t.matrix.byteOffset, t.width * t.height * 8)) : null,
        height: t.height,
        result: JSON.parse(JSON.stringify(result, (k, v) => (typeof v === "number" ? num(v) : v))),
        resultKeys: Object.keys(result),
    };
}

// node tests/golden/gen_golden_incremental.js            -> incremental.json.gz (options.useIncremental = true)
// node tests/golden/gen_golden_incremental.js enhanced   -> enhanced.json.gz (options.nodeSelection / branching alone:
//                                                           src/tableau/enhanced-branch-and-cut.ts, src/main.ts:74-80)
const enhanced = process.argv[2] === "enhanced";
const policies = enhanced
    ? [
        { nodeSelection: "hybrid" }, // + the default branching of that service: pseudocost
        { nodeSelection: "depth-first" },
        { nodeSelection: "best-first" },
        { branching: "most-fractional" },
        { branching: "strong" },
        { nodeSelection: "best-first", branching: "strong" },
        { nodeSelection: "depth-first", branching: "most-fractional" },
    ]
    : [
        {}, // the defaults: hybrid + pseudocost
        { nodeSelection: "depth-first" },
        { nodeSelection: "best-first" },
        { branching: "most-fractional" },
        { nodeSelection: "depth-first", branching: "most-fractional" },
    ];

const cases = [];
for (const dir of ["fixtures", "synthetic"]) {
    const d = path.join(__dirname, dir);
    for (const f of fs.readdirSync(d).filter((x) => x.endsWith(".json.gz")).sort()) {
        const g = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(d, f))).toString());
        if (!g.model || !g.tableau || g.tableau.integerVarIndexes.length === 0) continue;
        if (g.tableau.useMIRCuts) continue;
        for (const pol of policies) {
            const model = JSON.parse(JSON.stringify(g.model));
            model.options = Object.assign({}, model.options || {}, pol, enhanced ? {} : { useIncremental: true });
            const out = run(model);
            out.file = dir + "/" + f;
            out.options = model.options;
            // the pre-pass outcome decides whether a host without presolve can replay the case
            out.presolveFixed = g.presolve ? g.presolve.nFixed : 0;
            cases.push(out);
            console.log(f, JSON.stringify(pol), out.iterations, out.nPivots, out.pivotDigest, out.result.result);
        }
    }
}
fs.writeFileSync(path.join(__dirname, enhanced ? "enhanced.json.gz" : "incremental.json.gz"), zlib.gzipSync(Buffer.from(JSON.stringify(cases)), { level: 9 }));
console.log(cases.length, "cases");
