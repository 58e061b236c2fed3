// TEST INFRASTRUCTURE ONLY.  Runs every known-answer fixture of the reference through oracle/_ref (the
// type-erased reference itself) with the reference's own comparison rule
// (src/solver.integration.test.ts:60-100: toFixed(6) normalisation, both-infeasible short-circuit,
// keys feasible/_timeout/isIntegral/bounded ignored).  Node 12 compatible.
//
//   node oracle/ref_fixtures.js <dir with *.json fixtures>
"use strict";
const fs = require("fs");
const path = require("path");
const solver = require("./_ref/src/solver.js").default;

function normalize(value) {
    if (typeof value === "string") {
        const n = Number(value);
        if (!Number.isNaN(n)) return normalize(n);
        return value;
    }
    if (typeof value === "number" && Number.isFinite(value)) return Number(value.toFixed(6));
    return value === undefined || value === null ? 0 : value;
}

function compare(actual, expected) {
    if (!actual.feasible && !expected.feasible) return [];
    const bad = [];
    if (actual.feasible !== expected.feasible) bad.push("feasible");
    for (const key of Object.keys(expected)) {
        if (key === "feasible" || key === "_timeout" || key === "isIntegral" || key === "bounded") continue;
        if (normalize(actual[key]) !== normalize(expected[key])) bad.push(key);
    }
    return bad;
}

const dir = process.argv[2];
let pass = 0;
let fail = 0;
for (const f of fs.readdirSync(dir).filter((x) => x.endsWith(".json")).sort()) {
    const model = JSON.parse(fs.readFileSync(path.join(dir, f), "utf8"));
    const t0 = process.hrtime.bigint();
    const res = solver.Solve(model);
    const ms = Number(process.hrtime.bigint() - t0) / 1e6;
    const bad = compare(res, model.expects);
    if (bad.length === 0) {
        pass += 1;
    } else {
        fail += 1;
        console.log("FAIL", f, bad.join(","), JSON.stringify(res).slice(0, 200));
    }
    if (process.argv.includes("-v")) console.log((bad.length ? "FAIL " : "ok   ") + f + "  " + ms.toFixed(1) + " ms");
}
console.log(JSON.stringify({ pass, fail }));
process.exit(fail === 0 ? 0 : 1);
