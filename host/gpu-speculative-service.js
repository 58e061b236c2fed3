// host/gpu-speculative-service.js -- the DEFAULT branch-and-bound policy with speculative, batched evaluation.
//
// Same contract and the same tree policy as the reference's createBranchAndCutService (src/tableau/branch-and-cut.ts:32-199:
// best-first on the relaxed evaluation with LIFO ties, src/tableau/min-heap.ts:43-49; most-fractional branching,
// src/tableau/mip-utils.ts:100-126; tolerance / timeout / optional-objective tie-break / keep_solutions), injected
// through the same seam (src/main.ts:62-83) when install() is given `speculate > 1`.  What changes is WHEN nodes are
// evaluated: every node is a pure function of the saved root and its cut list (applyCuts always restores the root,
// :33-37), so whenever the loop needs a node that has not been evaluated yet, that node and the next `speculate - 1`
// entries the heap would hand out are evaluated as ONE batch of independent relaxations on the MI355X
// (jslp_engine_relax_batch: one workgroup per node).  The loop itself still pops, prunes and commits in the reference's
// order (speculation with in-order commit), so the incumbent, the iteration count and the result are those of the
// sequential run (host/test/dropin.js checks that against the goldens).
//
// Round 4 -- LOOKAHEAD (install(..., { lookahead: n }); off by default, see below).  A real tree is narrower than a batch: on Monster_II a batch of up to 16 nodes yields ~5 that the walk
// ever uses, and every NEW node (the children of the node committed a moment ago) costs one dependent round trip of ~115 us
// -- 31 of them are 4 of the 5 ms a Solve spends in the addon.  A batch therefore also carries the CHILDREN of nodes whose
// outcome is cached but which the walk has not reached yet: their compact outcome (row and value of every integer variable)
// is all isIntegral / getMostFractionalVar read (mip-utils.ts:43-61, 100-126), so the two cut lists the walk WILL build when it
// pops such a node (branch-and-cut.ts:163-191) can be built now.  Outcomes are cached by cut list (a relaxation is a pure
// function of the saved root and its cuts), so a wrong guess costs a wasted node, never a wrong answer: the walk itself is
// unchanged.
"use strict";

// the cut lists branch-and-cut.ts:163-191 derives from a node's outcome -- from the COMPACT read-back, without touching the tableau;
// null when the walk would not branch on it (infeasible, integral, or pruned by `bound`)
function predictChildren(cuts, outcome, precision, bound) {
    const res = outcome.res;
    if (res.feasible === false || !res.optimal || res.evaluation > bound) return null;
    const rows = outcome.rows, values = outcome.values;
    let biggest = 0, selected = -1, selectedValue = 0, integral = true;
    for (let k = 0; k < rows.length; k++) {
        if (rows[k] === -1) continue;
        const v = values[k];
        const fraction = Math.abs(v - Math.round(v));
        if (fraction > precision) integral = false;
        if (fraction > biggest) {
            biggest = fraction;
            selected = k;
            selectedValue = v;
        }
    }
    if (integral || selected < 0) return null;
    return { k: selected, value: selectedValue };
}

function cutsKey(cuts) {
    let s = "";
    for (let i = 0; i < cuts.length; i++) s += (cuts[i].type === "min" ? "m" : "M") + cuts[i].varIndex + ":" + cuts[i].value + "|";
    return s;
}

function Heap() {
    this.items = [];
    this.stamp = 0;
}
Heap.prototype.before = function (a, b) {
    return a.key !== b.key ? a.key < b.key : a.stamp > b.stamp;
};
Heap.prototype.push = function (key, cuts) {
    const it = { key, stamp: this.stamp++, cuts };
    const h = this.items;
    let i = h.length;
    h.push(it);
    while (i > 0) {
        const p = (i - 1) >> 1;
        if (!this.before(it, h[p])) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = it;
};
Heap.prototype.pop = function () {
    const h = this.items;
    const top = h[0];
    const last = h.pop();
    const n = h.length;
    if (n > 0) {
        let i = 0;
        for (;;) {
            let c = 2 * i + 1;
            if (c >= n) break;
            if (c + 1 < n && this.before(h[c + 1], h[c])) c += 1;
            if (!this.before(h[c], last)) break;
            h[i] = h[c];
            i = c;
        }
        h[i] = last;
    }
    return top;
};

function createGpuSpeculativeService(gpu, options) {
    const width = Math.max(1, (options && options.speculate) || 16);

    function applyCuts(t, cuts) { // :33-37 (MIR models never get this service)
        t.restore();
        t.addCutConstraints(cuts);
        t.simplex();
    }

    const fallback = options && options.fallback;

    function branchAndCut(t) {
        if (fallback && !gpu.isOnEngine(t)) return fallback.branchAndCut(t); // kept off the engine by the host's size policy
        const model = t.model;
        const heap = new Heap();
        const cache = new Map(); // cut list (cutsKey) -> outcome of that node
        // extra nodes per batch: children of cached nodes.  OFF by default: measured on the MI355X box (profiles/r04_speculative_lookahead.txt)
        // Monster_II goes from 31 to 28 batches per Solve and 8.4-8.7 to 7.3-8.6 ms (inside the run-to-run spread), LargeFarmMIP from 135
        // to 97 batches but 35 -> 38 ms (26 % more nodes evaluated and keyed): best-first with LIFO ties DIVES -- the node the walk
        // needs next is a child of the node it committed a moment ago, whose outcome nothing could have known a batch earlier
        const lookahead = options && options.lookahead !== undefined ? options.lookahead : 0;
        const stats = { batches: 0, evaluated: 0, committedFromCache: 0, lookaheadNodes: 0 };
        const keyOf = (it) => (it.ck !== undefined ? it.ck : (it.ck = cutsKey(it.cuts)));
        const childLists = (cuts, varIndex, value) => {  // branch-and-cut.ts:163-191
            const high = [], low = [];
            for (let k = 0; k < cuts.length; k++) {
                const cut = cuts[k];
                if (cut.varIndex !== varIndex) {
                    high.push(cut);
                    low.push(cut);
                } else if (cut.type === "min") low.push(cut);
                else high.push(cut);
            }
            high.push({ type: "min", varIndex, value: Math.ceil(value) });
            low.push({ type: "max", varIndex, value: Math.floor(value) });
            return [high, low];
        };
        let iterations = 0;
        const tolerance = model && model.tolerance ? model.tolerance : 0;
        let withinTolerance = true;
        const deadline = model && model.timeout ? Date.now() + model.timeout : 1e99;
        let bestEvaluation = Infinity;
        let bestCuts = null;
        let lastCuts = null;
        let saved = false;
        const nOpt = t.optionalObjectives.length;
        const bestOptional = new Array(nOpt).fill(Infinity);
        // the tie-break below reads the live optional-objective cells: evaluate such models in order; and a batch runs one
        // workgroup per node, which only pays while a node's tableau is small next to the chip (a single child of a
        // 20 MB tableau is faster through the chip-wide kernels, one node at a time)
        const speculate = nOpt > 0 || t.width * t.height > 1536 * 1024 ? 1 : width;
        // keep_solutions reads every node's whole column (round 4: the device pool -- install(..., {devices}) -- splits the compact
        // read-back too: jslp_pool_relax_batch_watched)
        const compact = !(options && options.fullReadBack) && !(model && model.keep_solutions);

        heap.push(-Infinity, []);
        while (heap.items.length > 0 && withinTolerance && Date.now() < deadline) {
            const acceptable = model && model.isMinimization
                ? t.bestPossibleEval * (1 + tolerance)
                : t.bestPossibleEval * (1 - tolerance);
            if (tolerance > 0 && bestEvaluation < acceptable) withinTolerance = false;

            const node = heap.pop();
            if (node.key > bestEvaluation) continue;
            const cuts = node.cuts;
            if (speculate > 1 && saved) {
                if (!cache.has(keyOf(node))) {
                    // this node + what the heap would hand out next (best first, LIFO ties) that is not pruned already
                    const ahead = heap.items.slice().sort((a, b) => (a.key !== b.key ? a.key - b.key : b.stamp - a.stamp));
                    const batch = [node.cuts];
                    const keys = [keyOf(node)];
                    const inBatch = new Set(keys);
                    for (let i = 0; i < ahead.length && batch.length < speculate; i++) {
                        const k = keyOf(ahead[i]);
                        if (!cache.has(k) && !inBatch.has(k) && ahead[i].key <= bestEvaluation) { batch.push(ahead[i].cuts); keys.push(k); inBatch.add(k); }
                    }
                    // lookahead: the children of nodes the walk has not reached yet but whose outcome is known
                    if (compact && lookahead > 0) {
                        const ints = model.integerVariables;
                        const limit = batch.length + lookahead;
                        for (let i = 0; i < ahead.length && batch.length < limit; i++) {
                            if (ahead[i].key > bestEvaluation) break;
                            const oc = cache.get(keyOf(ahead[i]));
                            if (oc === undefined) continue;
                            const pick = predictChildren(ahead[i].cuts, oc, t.precision, bestEvaluation);
                            if (pick === null) continue;
                            const lists = childLists(ahead[i].cuts, ints[pick.k].index, pick.value);
                            for (let c = 0; c < 2 && batch.length < limit; c++) {
                                const k = cutsKey(lists[c]);
                                if (!cache.has(k) && !inBatch.has(k)) { batch.push(lists[c]); keys.push(k); inBatch.add(k); stats.lookaheadNodes += 1; }
                            }
                        }
                    }
                    // compact read-back: per node the row / value of the integer variables -- all the tree reads between relaxations
                    const outcomes = compact ? gpu.relaxBatchWatched(t, batch) : gpu.relaxBatch(t, batch);
                    for (let i = 0; i < batch.length; i++) cache.set(keys[i], outcomes[i]);
                    stats.batches += 1;
                    stats.evaluated += batch.length;
                } else stats.committedFromCache += 1;
                const outcome = cache.get(keyOf(node));
                cache.delete(keyOf(node));
                // restore() + addCutConstraints(cuts) bookkeeping + the cached simplex()
                if (compact) gpu.commitWatched(t, cuts, outcome);
                else gpu.commitOutcome(t, cuts, outcome);
            } else {
                applyCuts(t, cuts);
            }
            lastCuts = cuts;
            iterations += 1;
            if (t.feasible === false) continue;
            const evaluation = t.evaluation;
            if (evaluation > bestEvaluation) continue;
            if (evaluation === bestEvaluation) {
                let worse = true; // :107-127
                for (let k = 0; k < nOpt; k++) {
                    const cell = t.optionalObjectives[k].reducedCosts[0];
                    if (cell > bestOptional[k]) break;
                    if (cell < bestOptional[k]) {
                        worse = false;
                        break;
                    }
                }
                if (worse) continue;
            }
            if (t.isIntegral() === true) {
                t.__isIntegral = true;
                if (iterations === 1) {
                    t.branchAndCutIterations = iterations;
                    return;
                }
                bestCuts = cuts;
                bestEvaluation = evaluation;
                for (let k = 0; k < nOpt; k++) bestOptional[k] = t.optionalObjectives[k].reducedCosts[0];
                if (model && model.keep_solutions) {
                    const now = model.tableau.getSolution();
                    const store = now.generateSolutionSet();
                    store.result = now.evaluation;
                    if (!model.solutions) model.solutions = [];
                    model.solutions.push(store);
                }
            } else {
                if (iterations === 1) {
                    t.save();
                    saved = true;
                }
                const variable = t.getMostFractionalVar();
                const lists = childLists(cuts, variable.index, variable.value);
                heap.push(evaluation, lists[0]);
                heap.push(evaluation, lists[1]);
            }
        }
        if (bestCuts !== null) applyCuts(t, bestCuts); // :194-196
        else if (speculate > 1 && saved && lastCuts !== null) applyCuts(t, lastCuts); // no incumbent: end where the sequential run ends
        t.branchAndCutIterations = iterations;
        t.__gpuSpeculativeStats = stats; // (diagnostics: host/test/dropin.js, bench.py's dropin_js leg)
    }

    return { applyCuts, branchAndCut, predictChildren, __gpuSpeculative: true };
}

module.exports = { createGpuSpeculativeService };
