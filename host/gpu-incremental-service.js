// host/gpu-incremental-service.js -- the incremental branch-and-bound service over DEVICE-RESIDENT checkpoints.
//
// Same contract as the reference's createIncrementalBranchAndCutService (src/tableau/incremental-branch-and-cut.ts:130-499,
// selected by options.useIncremental in src/main.ts:62-72): an object {applyCuts, branchAndCut} injected as the
// Tableau's BranchAndCutService (src/tableau/branch-and-cut.ts:19-22).  The reference keeps a parent's state as a host
// copy of tableau.matrix (:55-70) and puts it back with matrix.set (:72-107); with the tableau living on the MI355X the
// copy is a device buffer instead (jslp_engine_checkpoint_create) and a child = restoreCheckpoint + ONE cut + simplex is
// one addon call (jslp_engine_relax_from).  The tree policy -- depth-first stack, best-first heap, the hybrid switch after
// the first incumbent, pseudocost scores and their order-dependent update -- decides which incumbent wins, so it is
// restated here line for line in behaviour (not in code): pivots, relaxation count and results equal the reference's
// (host/test/dropin.js compares them for every integer fixture and policy).
"use strict";

// min-heap on relaxedEvaluation; equal keys leave in LIFO order (src/tableau/min-heap.ts:43-49)
function Heap() {
    this.items = [];
    this.stamp = 0;
}
Heap.prototype.before = function (a, b) {
    return a.key !== b.key ? a.key < b.key : a.stamp > b.stamp;
};
Heap.prototype.push = function (branch) {
    const it = { key: branch.relaxedEvaluation, stamp: this.stamp++, branch };
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
    return top.branch;
};

function createGpuIncrementalService(gpu, options) {
    const o = options || {};
    const nodeSelection = o.nodeSelection || "hybrid";
    const branching = o.branching || "pseudocost";
    const maxCheckpoints = o.maxCheckpoints === undefined ? 50 : o.maxCheckpoints;

    // pseudocosts per variable index: [upSum, upCount, downSum, downCount] (:138-174)
    const pseudo = new Map();
    const pc = (v) => {
        let d = pseudo.get(v);
        if (!d) pseudo.set(v, (d = [0, 0, 0, 0]));
        return d;
    };
    const score = (v, fraction) => {
        const d = pc(v);
        const up = d[1] > 0 ? d[0] / d[1] : 1;
        const down = d[3] > 0 ? d[2] / d[3] : 1;
        return Math.max(up * (1 - fraction), 1e-6) * Math.max(down * fraction, 1e-6);
    };

    // :176-221
    function pickVariable(t) {
        const found = [];
        const ints = t.model.integerVariables;
        for (let i = 0; i < ints.length; i++) {
            const row = t.rowByVarIndex[ints[i].index];
            if (row === -1) continue;
            const value = t.matrix[row * t.width + t.rhsColumn];
            const fraction = Math.abs(value - Math.round(value));
            if (fraction > t.precision) found.push({ index: ints[i].index, value, fraction });
        }
        if (found.length === 0) return null;
        let best = found[0];
        if (branching === "most-fractional") {
            for (let i = 1; i < found.length; i++) if (found[i].fraction > best.fraction) best = found[i];
            return best;
        }
        let bestScore = -Infinity;
        for (let i = 0; i < found.length; i++) {
            const s = score(found[i].index, found[i].fraction);
            if (s > bestScore) {
                bestScore = s;
                best = found[i];
            }
        }
        return best;
    }

    // the MIR loop both evaluation paths end with (:228-243, :261-276): feasible tableaus only, at most 3 rounds, stop when
    // the fractional volume shrinks by less than 10 %
    function mirLoop(t) {
        if (!(t.model && t.model.useMIRCuts) || !t.feasible) return;
        for (let round = 0; round < 3; round++) {
            const before = t.computeFractionalVolume(true);
            t.applyMIRCuts();
            t.simplex();
            if (t.computeFractionalVolume(true) >= 0.9 * before) break;
        }
    }

    // the default path: from the saved root with the whole cut list (:224-245)
    function applyCuts(t, cuts) {
        t.restore();
        t.addCutConstraints(cuts);
        t.simplex();
        mirLoop(t);
    }

    function evaluate(t, branch) {
        if (branch.checkpoint && branch.newCut) {
            gpu.relaxFromCheckpoint(t, branch.checkpoint, [branch.newCut]); // :248-253
            mirLoop(t);
        } else {
            applyCuts(t, branch.cuts); // :254-258
        }
    }

    function branchAndCut(t) {
        if (o.fallback && !gpu.isOnEngine(t)) return o.fallback.branchAndCut(t); // kept off the engine by the host's size policy
        const model = t.model;
        const heap = new Heap();
        const stack = [];
        const taken = [];
        let iterations = 0;
        let checkpointCount = 0;
        const tolerance = model && model.tolerance ? model.tolerance : 0;
        let withinTolerance = true;
        const deadline = model && model.timeout ? Date.now() + model.timeout : 1e99;
        let bestEvaluation = Infinity;
        let bestBranch = null;
        const nOpt = t.optionalObjectives.length;
        const bestOptional = new Array(nOpt).fill(Infinity);
        let solutionsFound = 0;
        let depthFirst = nodeSelection === "depth-first" || nodeSelection === "hybrid";

        const root = { relaxedEvaluation: -Infinity, cuts: [], depth: 0, checkpoint: undefined, newCut: undefined };
        if (depthFirst) stack.push(root);
        else heap.push(root);

        try {
            while ((depthFirst ? stack.length > 0 : heap.items.length > 0) && withinTolerance && Date.now() < deadline) {
                const acceptable = model && model.isMinimization
                    ? t.bestPossibleEval * (1 + tolerance)
                    : t.bestPossibleEval * (1 - tolerance);
                if (tolerance > 0 && bestEvaluation < acceptable) withinTolerance = false;

                let branch;
                if (depthFirst && stack.length > 0) branch = stack.pop();
                else if (heap.items.length > 0) branch = heap.pop();
                else break;
                if (branch.relaxedEvaluation > bestEvaluation) continue;

                const parentEval = t.evaluation;
                evaluate(t, branch);
                iterations += 1;
                if (!t.feasible) continue;
                const evaluation = t.evaluation;
                if (evaluation > bestEvaluation) continue;

                if (branch.newCut && parentEval !== 0) {
                    // :354-365 (fraction is the constant 0.5 there)
                    const d = pc(branch.newCut.varIndex);
                    const gain = Math.abs(evaluation - parentEval) / 0.5;
                    if (branch.newCut.type === "min") {
                        d[0] += gain;
                        d[1] += 1;
                    } else {
                        d[2] += gain;
                        d[3] += 1;
                    }
                }

                if (evaluation === bestEvaluation) {
                    let worse = true; // :367-388
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

                if (t.isIntegral()) {
                    t.__isIntegral = true;
                    solutionsFound += 1;
                    if (iterations === 1) {
                        t.branchAndCutIterations = iterations;
                        return;
                    }
                    bestBranch = branch;
                    bestEvaluation = evaluation;
                    for (let k = 0; k < nOpt; k++) bestOptional[k] = t.optionalObjectives[k].reducedCosts[0];
                    if (model && model.keep_solutions) {
                        const now = model.tableau.getSolution();
                        const store = now.generateSolutionSet();
                        store.result = now.evaluation;
                        if (!model.solutions) model.solutions = [];
                        model.solutions.push(store);
                    }
                    if (nodeSelection === "hybrid" && solutionsFound >= 1) {
                        depthFirst = false; // :418-428
                        while (stack.length > 0) heap.push(stack.pop());
                    }
                    continue;
                }

                if (iterations === 1) t.save();
                const picked = pickVariable(t);
                if (!picked) continue;

                let checkpoint;
                if (depthFirst && checkpointCount < maxCheckpoints) {
                    checkpoint = gpu.createCheckpoint(t); // :440-445, in HBM
                    taken.push(checkpoint);
                    checkpointCount += 1;
                }
                const high = [];
                const low = [];
                for (let k = 0; k < branch.cuts.length; k++) {
                    const cut = branch.cuts[k];
                    if (cut.varIndex !== picked.index) {
                        high.push(cut);
                        low.push(cut);
                    } else if (cut.type === "min") low.push(cut);
                    else high.push(cut);
                }
                const cutHigh = { type: "min", varIndex: picked.index, value: Math.ceil(picked.value) };
                const cutLow = { type: "max", varIndex: picked.index, value: Math.floor(picked.value) };
                high.push(cutHigh);
                low.push(cutLow);
                const depth = branch.depth + 1;
                if (depthFirst) {
                    stack.push({ relaxedEvaluation: evaluation, cuts: low, depth, checkpoint, newCut: cutLow });
                    stack.push({ relaxedEvaluation: evaluation, cuts: high, depth, checkpoint, newCut: cutHigh });
                } else {
                    heap.push({ relaxedEvaluation: evaluation, cuts: high, depth, checkpoint: undefined, newCut: undefined });
                    heap.push({ relaxedEvaluation: evaluation, cuts: low, depth, checkpoint: undefined, newCut: undefined });
                }
            }
            if (bestBranch !== null) applyCuts(t, bestBranch.cuts); // :491-493
            t.branchAndCutIterations = iterations;
        } finally {
            for (let k = 0; k < taken.length; k++) gpu.releaseCheckpoint(t, taken[k]);
            t.__gpuCheckpoints = checkpointCount;
        }
    }

    return { applyCuts, branchAndCut, __gpuIncremental: true };
}

module.exports = { createGpuIncrementalService };
