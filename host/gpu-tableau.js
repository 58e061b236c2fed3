// host/gpu-tableau.js -- the reference-side binding of the MI355X engine (node >= 12, plain CommonJS).
//
// jsLPSolver keeps everything it does today on the CPU -- JSON model parsing, validation, presolve, the
// branch-and-bound tree (src/tableau/branch-and-cut.ts) and result assembly -- and only the hot path moves:
// install() overrides, on the reference's own Tableau class, exactly the methods SURVEY.md 8b names as the seam
//   Tableau.simplex()            src/tableau/tableau.ts:103-111  -> jslp_engine_simplex / jslp_engine_relax
//   Tableau.save() / restore()   src/tableau/tableau.ts:223-229  -> jslp_engine_save / (folded into relax)
//   Tableau.addCutConstraints()  src/tableau/tableau.ts:145-147  -> (folded into relax)
// so `solver.Solve(model)` keeps its API and result shape.  One B&B node (restore + addCutConstraints +
// simplex, branch-and-cut.ts:33-37) becomes ONE addon call; after it only what the host tree reads comes
// back: flags, evaluation, the RHS column and the row -> variable map (mip-utils.ts:43-61,100-126).
//
// Optional objectives (soft constraints, simplex.ts:221-263,394-412) travel with the tableau: their reducedCosts rows are
// uploaded next to the matrix and read back after every simplex().  The incremental B&B service (options.useIncremental)
// runs over device-resident checkpoints (host/gpu-incremental-service.js) when install() is given the solver; MIR cuts
// (options.useMIRCuts) are built on the device by Tableau.applyMIRCuts()'s override.
"use strict";
const path = require("path");

const DEFAULT_ADDON = path.join(__dirname, "..", "addon", "jslp_napi.node");
const DEFAULT_LIBRARY = path.join(__dirname, "..", "jslpsolver_amd", "csrc", "libjslp_hip.so");

let addon = null;
let backend = null;
let bypass = 0; // > 0 while a Solve() that must stay on the reference's own path is running
// extra cut rows for the incremental service: it stacks one row per tree level and never merges cuts on one variable
// (incremental-branch-and-cut.ts:248-253); the reference reallocates, device memory is sized once
const INCREMENTAL_EXTRA_ROWS = 256;
// MIR cuts: every round appends up to 10 rows (cutting-strategies.ts:199-212) and the default service's loop is bounded
// only by a 10 % volume gain per round (branch-and-cut.ts:38-52): 32 rounds of headroom, loud failure beyond
const MIR_EXTRA_ROWS = 320;

function loadEngine(options) {
    const o = options || {};
    addon = require(o.addon || DEFAULT_ADDON);
    backend = addon.load(o.library || DEFAULT_LIBRARY); // throws when the library is missing: no CPU fallback
    return backend;
}

// Size policy: which tableaus go to the engine.  Measured on the MI355X box (tools/mincells_sweep.js, tools/shim_profile.js;
// profiles/r02_mincells_sweep.md, profiles/r05_policy_sweep.md): what the engine charges per simplex() is FIXED -- create + pinned build
// buffer + upload + one launch + read-back, ~0.3-0.5 ms -- plus ~6-7 us per pivot whatever the pivot touches; what the reference's CPU
// loop charges is proportional to the cells its zero gates let through (simplex.ts:370-383), i.e. to the NON-ZEROS of the tableau, not
// to its cells.  An LP therefore goes to the engine by its structural non-zero count (the model's term count, known before the matrix
// is built): a dense 120 x 90 LP (8.8 k non-zeros) pays, BASELINE's config 2 -- 'Monster LP', 625 x 553 = 345 k cells but 3.4 k non-zeros,
// 60 pivots, 0.66 ms inside Model.solve on the CPU -- does not and stays on the reference's own path (three rounds it lost 2x on the
// engine).  A branch-and-bound tree walked node by node costs ~50-80 us per relaxation on the engine whatever the size, which the
// reference's CPU path undercuts until a relaxation touches a few hundred thousand cells (LargeFarmMIP, 36 x 101: 0.017 ms per
// relaxation on the CPU); batched speculative evaluation (install(..., {speculate})) amortises that latency over the batch.
// opts.minCells overrides every default (0 = everything runs on the engine: what the parity runs use); opts.minNnz overrides the LP rule.
const DEFAULT_MIN_CELLS_LP = 8192;
const DEFAULT_MIN_NNZ_LP = 8192;
const DEFAULT_MIN_CELLS_INTEGER = 262144;
const DEFAULT_MIN_CELLS_INTEGER_BATCHED = 32768;
function minCellsFor(t, opts) {
    if (opts.minCells !== undefined) return opts.minCells;
    const nInts = t.model ? t.model.getNumberOfIntegerVariables() : 0;
    if (nInts === 0) return DEFAULT_MIN_CELLS_LP;
    return opts.speculate > 1 ? DEFAULT_MIN_CELLS_INTEGER_BATCHED : DEFAULT_MIN_CELLS_INTEGER;
}
// structural non-zeros of the tableau _resetMatrix is about to build (tableau.ts:319-380): one cell per term, one per non-zero
// right-hand side and cost; -1 when the tableau has no model to count on.  O(constraints + variables), no pass over the matrix.
function structuralNnz(t) {
    const m = t.model;
    if (!m || !m.constraints || !m.variables) return -1;
    // (counted once per built tableau: initialize() and the first simplex() both ask)
    const c = t.__gpuNnz;
    if (c !== undefined && c.model === m && c.nc === m.constraints.length && c.nv === m.variables.length) return c.n;
    let n = 0;
    const cs = m.constraints;
    for (let i = 0; i < cs.length; i++) n += cs[i].terms.length + (cs[i].rhs !== 0 ? 1 : 0);
    const vs = m.variables;
    for (let i = 0; i < vs.length; i++) if (vs[i].cost !== 0) n += 1;
    t.__gpuNnz = { model: m, nc: cs.length, nv: vs.length, n };
    return n;
}
// An LP below the non-zero threshold is not necessarily cheap: a sparse RANDOM 300 x 225 LP with 3.9 k non-zeros fills in and takes the
// reference 11.9 s (thousands of pivots) where the engine takes 24 ms, while the structured Monster LP with 3.4 k non-zeros is done in 60
// pivots (profiles/r05_policy_sweep.md).  The count cannot tell them apart, so such an LP STARTS on the reference's own path with a time
// budget (ski rental: cpuBudgetMs, default 3 ms = several times the engine's fixed cost per Solve, and 4x what Monster LP needs); when the budget runs out the solve
// is abandoned, the tableau is built again -- this time in the engine's pinned buffer -- and solved on the engine FROM THE START, so the
// answer and the pivot sequence are the reference's whichever side finishes.  (Continuing the half-done tableau on the engine would
// re-enter phase 1, which the reference never does in mid-phase 2.)  LPs with optional objectives are not deferred (an aborted run
// leaves their reducedCosts rows modified, and setModel does not rebuild the slack columns' entries).
const DEFAULT_CPU_BUDGET_MS = 3;
const BAIL = { bail: true };  // thrown by the pivot() wrapper when a deferred LP's budget is spent
function lpBelowNnz(t, opts) {
    if (!(opts.minCells === undefined || opts.minNnz !== undefined)) return false;
    const minNnz = opts.minNnz !== undefined ? opts.minNnz : DEFAULT_MIN_NNZ_LP;
    const nInts = t.model ? t.model.getNumberOfIntegerVariables() : 0;
    if (!(minNnz > 0 && nInts === 0)) return false;
    const nnz = structuralNnz(t);
    return nnz >= 0 && nnz < minNnz;
}
function eligible(t, opts) {
    if (bypass !== 0) return false;
    const min = minCellsFor(t, opts);
    if (min > 0 && t.width * t.height < min) return false;
    if (t.__gpuForce === true) return true;  // (a deferred LP whose CPU budget ran out: to the engine, whatever it counts)
    // LPs: by work, not by area (MILPs keep the cell rule: their trees re-solve the same tableau hundreds of times)
    if (lpBelowNnz(t, opts)) return false;
    return true;
}
// may this tableau, kept off the engine by the non-zero rule, still move there when its CPU solve runs long?
function deferrable(t, opts) {
    if (bypass !== 0 || t.__gpuForce === true) return false;
    const budget = opts.cpuBudgetMs !== undefined ? opts.cpuBudgetMs : DEFAULT_CPU_BUDGET_MS;
    if (!(budget > 0) || !t.model) return false;
    const min = minCellsFor(t, opts);
    if (min > 0 && t.width * t.height < min) return false;
    return lpBelowNnz(t, opts) && t.optionalObjectives.length === 0;
}

// the engine of a tableau whose dimensions are known (Tableau.initialize ran); the upload follows in activate()
function createEngine(t, opts) {
    const nInts = t.model ? t.model.getNumberOfIntegerVariables() : 0;
    const incremental = !!(t.branchAndCutService && t.branchAndCutService.__gpuIncremental);
    // <= one "min" and one "max" cut per integer variable for the services that start every node from the root
    const useMir = !!(t.model && t.model.useMIRCuts);
    const rowCapacity = t.height + 2 * nInts + 8 + (incremental ? INCREMENTAL_EXTRA_ROWS : 0) + (useMir ? MIR_EXTRA_ROWS : 0);
    const h = addon.create(t.height, t.width, rowCapacity, t.precision, opts.device || 0);
    return { h, rowCapacity, height: t.height, width: t.width, nInts, useMir, pinnedMatrix: null };
}

function activate(t, opts) {
    if (t.__gpuReleased) throw new Error("[gpu-tableau] this tableau's engine was released when Solve() returned; " +
        "use Solve(model, precision, true) (or model.solve()) to keep working with the tableau");
    if (!eligible(t, opts)) {
        t.__gpu = { active: false, deferred: deferrable(t, opts) };
        return t.__gpu;
    }
    // built in the engine's pinned buffer by the initialize() override?  (same dimensions, nothing appended since)
    let pre = t.__gpuEngine || null;
    t.__gpuEngine = null;
    if (pre && (pre.height !== t.height || pre.width !== t.width)) {
        // the tableau grew or shrank between initialize() and its first simplex(): the matrix moves to ordinary memory BEFORE the
        // engine (and with it the pinned buffer `t.matrix` may still view) goes back to the library's resource pool
        if (pre.pinnedMatrix) {
            if (t.matrix === pre.pinnedMatrix) t.matrix = new Float64Array(pre.pinnedMatrix);
            addon.detach(pre.pinnedMatrix);
            pre.pinnedMatrix = null;
        }
        addon.destroy(pre.h);
        pre = null;
    }
    const made = pre || createEngine(t, opts);
    const h = made.h, rowCapacity = made.rowCapacity, nInts = made.nInts, useMir = made.useMir;
    const rows = Int32Array.from(t.varIndexByRow);
    const cols = Int32Array.from(t.varIndexByCol);
    rows[0] = -1;
    cols[0] = -1;
    const unrestricted = Int32Array.from(
        Object.keys(t.unrestrictedVars).filter((k) => t.unrestrictedVars[k] === true).map(Number)
    );
    addon.upload(h, t.matrix.subarray(0, t.height * t.width), rows, cols, unrestricted);
    if (useMir) addon.setIntegerVariables(h, Int32Array.from(t.model.integerVariables.map((v) => v.index)));
    const nOpt = t.optionalObjectives.length; // already sorted by priority (tableau.ts:286)
    let optional = null;
    if (nOpt > 0) {
        optional = new Float64Array(nOpt * t.width);
        for (let o = 0; o < nOpt; o++) {
            const rc = t.optionalObjectives[o].reducedCosts;
            for (let c = 0; c < t.width; c++) optional[o * t.width + c] = rc[c] || 0;
        }
        addon.setOptionalObjectives(h, nOpt, optional);
    }
    // which variable indexes the branch-and-bound tree reads between relaxations (integer variables only:
    // isIntegral / getMostFractionalVar / computeFractionalVolume / the services' branching rules)
    let isInt = null;
    if (nInts > 0) {
        isInt = new Uint8Array(t.width + 2 * rowCapacity + 2);
        for (const v of t.model.integerVariables) isInt[v.index] = 1;
    }
    t.__gpu = {
        nOpt,
        optional,
        isInt,
        stale: false, // true while only the integer variables' rows of the host copy are up to date
        lastHeight: 0,
        active: true,
        h,
        rowCapacity,
        rhs: new Float64Array(rowCapacity),
        rows: new Int32Array(rowCapacity),
        pendingRestore: false,
        pendingCuts: null,
        saved: null,
        pinnedMatrix: made.pinnedMatrix,
        pool: null,
    };
    return t.__gpu;
}

// drop the device side of a tableau: pools first (their members hold copies of this engine's root), the pinned view of
// the build buffer is detached (the memory belongs to the engine), then the engine itself
function dropEngine(t, st, keepMatrix) {
    if (st.pool) {
        addon.poolDestroy(st.pool);
        st.pool = null;
    }
    if (st.pinnedMatrix) {
        // (a finished tableau -- release() -- is not worth a copy of its stale host matrix: detaching leaves it zero-length)
        if (t.matrix === st.pinnedMatrix && keepMatrix) t.matrix = new Float64Array(st.pinnedMatrix);
        addon.detach(st.pinnedMatrix);
        st.pinnedMatrix = null;
    }
    addon.destroy(st.h);
    st.active = false;
}

const stats = { deferredToEngine: 0 };  // deferred LPs whose CPU budget ran out (tests / tools read it)
let installedOpts = {};
function state(t, opts) {
    return t.__gpu || activate(t, opts || installedOpts);
}
// does this tableau live on the engine?  (decides it on first use: size policy, bypass)
function isOnEngine(t) {
    return state(t, installedOpts).active === true;
}

function packCuts(cuts) {
    const n = cuts.length;
    const type = new Int8Array(n);
    const varIndex = new Int32Array(n);
    const value = new Float64Array(n);
    for (let i = 0; i < n; i++) {
        type[i] = cuts[i].type === "min" ? 0 : 1; // JSLP_CUT_MIN / JSLP_CUT_MAX
        varIndex[i] = cuts[i].varIndex;
        value[i] = cuts[i].value;
    }
    return { type, varIndex, value };
}

// fold the engine's outcome into the Tableau exactly as simplex.ts / tableau.ts do on the CPU
function absorb(t, st, res) {
    t.feasible = res.feasible;
    t.bounded = res.bounded;
    if (res.optimal) {
        // setEvaluation + simplexIters += 1 (simplex.ts:265-269, tableau.ts:420-430)
        t.evaluation = res.evaluation;
        if (t.simplexIters === 0) t.bestPossibleEval = res.evaluation;
        t.simplexIters += 1;
    } else if (!res.bounded) {
        t.evaluation = -Infinity; // simplex.ts:298-303
        t.unboundedVarIndex = res.unboundedVarIndex;
    }
    if (res.cyclePhase !== 0 && t.model) {
        t.model.messages.push("Cycle in phase " + res.cyclePhase); // simplex.ts:86-88 / 313-315
        t.model.messages.push("Start :" + res.cycleStart);
        t.model.messages.push("Length :" + res.cycleLength);
    }
    if (st.nOpt > 0) {
        // the host reads optionalObjectives[o].reducedCosts[0] for its tie-break (branch-and-cut.ts:107-127)
        addon.getOptionalObjectives(st.h, st.optional);
        for (let o = 0; o < st.nOpt; o++) {
            const rc = t.optionalObjectives[o].reducedCosts;
            for (let c = 0; c < t.width; c++) rc[c] = st.optional[o * t.width + c];
        }
    }
    // read-back: RHS column + row map (and the inverse map the host tree indexes with).  While a tree is being walked
    // only the integer variables' rows are read on the host, so only those cells of the (large, strided) host matrix
    // are written per relaxation; flush() completes the picture before anything reads the other rows.
    const H = res.height;
    st.lastHeight = H;
    const width = t.width;
    const rhsColumn = t.rhsColumn;
    const matrix = t.matrix;
    const rowByVarIndex = t.rowByVarIndex;
    const varIndexByRow = t.varIndexByRow;
    const isInt = st.isInt;
    if (isInt === null) {
        for (let v = 0; v < rowByVarIndex.length; v++) rowByVarIndex[v] = -1;
        for (let r = 0; r < H; r++) {
            matrix[r * width + rhsColumn] = st.rhs[r];
            const v = st.rows[r];
            varIndexByRow[r] = v;
            if (v >= 0) rowByVarIndex[v] = r;
        }
        st.stale = false;
        return;
    }
    const ints = t.model.integerVariables;
    for (let i = 0; i < ints.length; i++) rowByVarIndex[ints[i].index] = -1;
    for (let r = 0; r < H; r++) {
        const v = st.rows[r];
        varIndexByRow[r] = v;
        if (v >= 0 && isInt[v] === 1) {
            rowByVarIndex[v] = r;
            matrix[r * width + rhsColumn] = st.rhs[r];
        }
    }
    st.stale = true;
}

// complete the host copy (every row's RHS, the whole rowByVarIndex) from the last read-back
function flush(t) {
    const st = t.__gpu;
    if (!st || !st.active) return;
    if (st.watchedCuts) {
        // the live node was committed from a COMPACT outcome (commitWatched: only the integer variables' rows are known to the
        // host): evaluate it once more with the full read-back -- a relaxation is a pure function of the saved root and its cuts
        const c = packCuts(st.watchedCuts);
        st.watchedCuts = null;
        const check = t.model ? t.model.checkForCycles === true : false;
        const res = addon.relax(st.h, c.type, c.varIndex, c.value, check, st.rhs, st.rows);
        st.lastHeight = res.height;
        st.stale = true;
    }
    if (!st.stale) return;
    const H = st.lastHeight;
    const width = t.width;
    const rhsColumn = t.rhsColumn;
    const matrix = t.matrix;
    const rowByVarIndex = t.rowByVarIndex;
    const varIndexByRow = t.varIndexByRow;
    for (let v = 0; v < rowByVarIndex.length; v++) rowByVarIndex[v] = -1;
    for (let r = 0; r < H; r++) {
        matrix[r * width + rhsColumn] = st.rhs[r];
        const v = st.rows[r];
        // (a compact commit -- commitWatched -- patched only the integer variables' rows of this map: copy(),
        //  computeFractionalVolume and whoever else reads it after a flush must see the node's whole row map)
        varIndexByRow[r] = v;
        if (v >= 0) rowByVarIndex[v] = r;
    }
    st.stale = false;
}

// speculative batches are the default whenever the solver instance is handed over (they need its service seam): same results and
// relaxation counts as the sequential walk by construction, Monster_II 17.0 -> 9.7 ms (profiles/r02_z_config_wall_times.md);
// install(..., { speculate: 0 }) keeps the reference's one-node-at-a-time services
const DEFAULT_SPECULATE = 16;
function install(Tableau, options) {
    const opts = Object.assign({}, options || {});
    if (opts.solver && opts.speculate === undefined) opts.speculate = DEFAULT_SPECULATE;
    installedOpts = opts;
    if (!addon) loadEngine(opts);
    const P = Tableau.prototype;
    const orig = { simplex: P.simplex, save: P.save, restore: P.restore, addCutConstraints: P.addCutConstraints,
        applyMIRCuts: P.applyMIRCuts, updateVariableValues: P.updateVariableValues, getSolution: P.getSolution };

    // SURVEY.md 8f.4: Tableau.initialize (tableau.ts:292-317) allocates `matrix`, _resetMatrix (:319-380) fills it cell by
    // cell.  For tableaus that go to the engine the array is a view of the engine's PINNED build buffer, so the upload that
    // follows is one DMA straight from where the host built the tableau (no pageable staging copy).
    const origInitialize = P.initialize;
    P.initialize = function (width, height, variables, unrestrictedVars) {
        origInitialize.call(this, width, height, variables, unrestrictedVars);
        if (this.__gpu) { // a tableau that is set up again (setModel twice): start over
            if (this.__gpu.active) dropEngine(this, this.__gpu, false);
            this.__gpu = undefined;
        }
        // (ADVICE r05: the deferred-LP budget -- see P.simplex -- is armed only for a tableau that setModel has JUST built: this flag is
        //  cleared by the first simplex() and by every editing call, so a tableau whose first solve was infeasible and which was then
        //  edited through the dynamic-modification API is never rebuilt from its model)
        this.__gpuFresh = true;
        if (opts.pinnedBuild === false || !eligible(this, opts)) return;
        const made = createEngine(this, opts);
        made.pinnedMatrix = addon.hostMatrix(made.h); // zero-filled like the Float64Array it replaces (tableau.ts:304)
        this.matrix = made.pinnedMatrix;
        this.__gpuEngine = made;
    };

    // The post-solve editing API (dynamic-modification.ts: updateRightHandSide, updateConstraintCoefficient, updateCost,
    // addConstraint, removeConstraint, addVariable, removeVariable, putInBase / takeOutOfBase -> pivot) and copy() work on the
    // HOST matrix and maps, which are stale while the tableau lives on the device (only the RHS column is mirrored).  They
    // first bring the tableau home -- full read-back, engine dropped -- and the next simplex() uploads the edited tableau.
    const editing = ["updateRightHandSide", "updateConstraintCoefficient", "updateCost", "addConstraint", "removeConstraint",
        "addVariable", "removeVariable", "putInBase", "takeOutOfBase", "pivot"];
    const origEditing = {};
    for (const name of editing) {
        if (typeof P[name] !== "function") continue;
        origEditing[name] = P[name];
        P[name] = function () {
            this.__gpuFresh = false;
            bringHome(this);
            return origEditing[name].apply(this, arguments);
        };
    }
    // pivot() is also what the reference's own phase loops call (simplex.ts:95, 322): a deferred LP's time budget is checked behind
    // every eighth pivot, i.e. in a consistent state between two iterations
    if (origEditing.pivot) {
        const pivotThenHome = P.pivot;
        P.pivot = function (r, c) {
            const b = this.__gpuBudget;
            if (b === null || b === undefined) return pivotThenHome.call(this, r, c);
            origEditing.pivot.call(this, r, c);
            b.pivots += 1;
            if ((b.pivots & 7) === 0 && process.hrtime.bigint() - b.t0 > b.ns) throw BAIL;
        };
    }
    const origCopy = P.copy;
    P.copy = function () {
        const st = this.__gpu;
        if (st && st.active) { flush(this); sync(this); }
        return origCopy.call(this);
    };

    // the two readers of EVERY row's value (tableau.ts:256-257; keep_solutions inside the services): complete the host copy first
    P.updateVariableValues = function () {
        flush(this);
        return orig.updateVariableValues.call(this);
    };
    P.getSolution = function () {
        flush(this);
        return orig.getSolution.call(this);
    };

    // slack bookkeeping of one appended row (cutting-strategies.ts:64-71 / :104-109); the row itself is built on the device
    function newSlackRow(t, row) {
        const slack = t.getNewElementIndex();
        t.varIndexByRow[row] = slack;
        t.rowByVarIndex[slack] = row;
        t.colByVarIndex[slack] = -1;
        // (slack indexes repeat from node to node -- restore() rewinds lastElementIndex --: one object per index is enough)
        const slackObjects = t.__gpu.slackObjects || (t.__gpu.slackObjects = []);  // per tableau: updateVariableValues writes their values
        let sv = slackObjects[slack];
        if (sv === undefined) {
            sv = opts.SlackVariable
                ? new opts.SlackVariable("s" + slack, slack)
                : { id: "s" + slack, cost: 0, index: slack, value: 0, priority: 0, isSlack: true };
            slackObjects[slack] = sv;
        }
        t.variablesPerIndex[slack] = sv;
    }

    // Tableau.applyMIRCuts (cutting-strategies.ts:199-212): the scan and the <= 10 new rows happen on the device
    P.applyMIRCuts = function () {
        this.__gpuFresh = false;
        const st = state(this, opts);
        if (!st.active) return orig.applyMIRCuts.call(this);
        if (st.pendingRestore || st.pendingCuts) throw new Error("[gpu-tableau] applyMIRCuts before the pending simplex()");
        const n = addon.applyMirCuts(st.h);
        if (this.matrix.length < (this.height + n) * this.width) this.matrix = new Float64Array(st.rowCapacity * this.width);
        for (let k = 0; k < n; k++) {
            const row = this.height;
            this.height += 1; // :101-102
            this.nVars += 1;
            newSlackRow(this, row);
        }
    };

    P.simplex = function () {
        const st = state(this, opts);
        const fresh = this.__gpuFresh === true;  // built by setModel and untouched since (not: "no optimum reached yet")
        this.__gpuFresh = false;
        if (!st.active && st.deferred === true && fresh) {
            // a small-count LP: the reference's own loop with a time budget (see `deferrable`); the pivot() wrapper below throws BAIL
            const budgetMs = opts.cpuBudgetMs !== undefined ? opts.cpuBudgetMs : DEFAULT_CPU_BUDGET_MS;
            this.__gpuBudget = { t0: process.hrtime.bigint(), ns: BigInt(Math.round(budgetMs * 1e6)), pivots: 0 };
            try {
                return orig.simplex.call(this);
            } catch (e) {
                if (e !== BAIL) throw e;
            } finally {
                this.__gpuBudget = null;
            }
            // budget spent: the same model once more, built in the engine's pinned buffer, solved there from the first pivot
            stats.deferredToEngine += 1;
            this.__gpuForce = true;
            this.__gpu = undefined;
            this.setModel(this.model);  // (initialize override: engine + pinned matrix; _resetMatrix: the reference's own builder)
            return this.simplex();
        }
        if (!st.active) return orig.simplex.call(this);
        const check = this.model ? this.model.checkForCycles === true : false;
        let res;
        if (st.pendingRestore) {
            // restore + addCutConstraints + simplex = one LP relaxation (branch-and-cut.ts:33-37)
            const c = packCuts(st.pendingCuts || []);
            res = addon.relax(st.h, c.type, c.varIndex, c.value, check, st.rhs, st.rows);
        } else {
            if (st.pendingCuts) {
                const c = packCuts(st.pendingCuts);
                addon.addCuts(st.h, c.type, c.varIndex, c.value);
            }
            res = addon.simplex(st.h, check);
            addon.readRhs(st.h, st.rhs, st.rows);
        }
        st.pendingRestore = false;
        st.pendingCuts = null;
        st.watchedCuts = null;  // a full read-back: the host copy is whole again
        absorb(this, st, res);
        return this;
    };

    P.save = function () {
        const st = state(this, opts);
        if (!st.active) return orig.save.call(this);
        addon.save(st.h); // device-resident snapshot (backup.ts:13-51)
        st.saved = { height: this.height, nVars: this.nVars, lastElementIndex: this.lastElementIndex };
    };

    P.restore = function () {
        this.__gpuFresh = false;
        const st = state(this, opts);
        if (!st.active) return orig.restore.call(this);
        if (st.saved === null) return; // backup.ts:54-56
        this.height = st.saved.height; // backup.ts:58-68 (the scalars); matrix + maps live on the device
        this.nVars = st.saved.nVars;
        this.lastElementIndex = st.saved.lastElementIndex;
        this.varIndexByRow.length = this.height;
        st.pendingRestore = true;
        st.pendingCuts = null;
    };

    P.addCutConstraints = function (cuts) {
        this.__gpuFresh = false;
        const st = state(this, opts);
        if (!st.active) return orig.addCutConstraints.call(this, cuts);
        // host-side bookkeeping of cutting-strategies.ts:16-34,64-71; the rows themselves are built on the device
        const n = cuts.length;
        const height = this.height;
        const heightWithCuts = height + n;
        if (heightWithCuts > st.rowCapacity) throw new Error("[gpu-tableau] cut rows exceed the engine's row capacity");
        // The reference grows the host matrix by reallocating and copying it (cutting-strategies.ts:24-30).  With the
        // tableau on the device only the RHS column of the host copy is live, and absorb() rewrites it for every row
        // after each simplex(): size the host array once for the row capacity and skip the copy (22 MB per new tree
        // depth on Vendor Selection).
        if (this.matrix.length < heightWithCuts * this.width) this.matrix = new Float64Array(st.rowCapacity * this.width);
        this.height = heightWithCuts;
        this.nVars = this.width + this.height - 2;
        for (let h = 0; h < n; h++) {
            newSlackRow(this, height + h);
            this.nVars += 1;
        }
        st.pendingCuts = (st.pendingCuts || []).concat(cuts);
    };

    // The reference's incremental B&B service (options.useIncremental, src/tableau/incremental-branch-and-cut.ts:55-107)
    // checkpoints a parent by copying tableau.matrix on the host, which no Tableau method can intercept.  Given the
    // solver, install() therefore answers its service selection (src/main.ts:62-83) with the same policy over device
    // checkpoints; without the solver such a Solve must be kept on the reference's own path (guardIncremental).
    let hadOwnSelect = false;
    let origSelect = null;
    let origSolve = null;
    if (opts.solver) {
        const solver = opts.solver;
        // a simplified-result Solve() is done with its tableau when it returns: hand the engine back right away (its
        // stream / arenas go to the library's resource pool for the next Solve) instead of waiting for the garbage
        // collector.  Solve(model, precision, full = true) keeps it for sync() / the post-solve API.
        origSolve = solver.Solve;
        solver.Solve = function (model, precision, full) {
            const result = origSolve.apply(this, arguments);
            // a Model INSTANCE handed in by the caller stays the caller's (main.ts:127-134: `modelInstance = model`; it may be
            // edited and solved again): its tableau keeps its engine.  Only the Model that Solve() itself built from a JSON
            // definition is finished when Solve() returns.
            const callersInstance = this.lastSolvedModel === model;
            if (!full && !callersInstance && this.lastSolvedModel && this.lastSolvedModel.tableau) release(this.lastSolvedModel.tableau);
            return result;
        };
        const service = require("./gpu-incremental-service.js");
        hadOwnSelect = Object.prototype.hasOwnProperty.call(solver, "selectBranchAndCutService");
        origSelect = solver.selectBranchAndCutService;
        solver.selectBranchAndCutService = function (model) {
            const o = model && model.options;
            // (each injected service falls back to the reference's own choice for tableaus the size policy keeps off the engine)
            if (o && o.useIncremental === true) {
                return service.createGpuIncrementalService(api, { nodeSelection: o.nodeSelection, branching: o.branching,
                    fallback: origSelect.call(this, model) });
            }
            // install(..., { speculate: n > 1 }): the default policy with n-node speculative batches (in-order commit);
            // models that ask for another policy or for MIR cuts keep the reference's own services
            if (opts.speculate > 1 && !(o && (o.nodeSelection || o.branching || o.useMIRCuts))) {
                return require("./gpu-speculative-service.js").createGpuSpeculativeService(api, { speculate: opts.speculate,
                    fullReadBack: opts.fullReadBack === true, lookahead: opts.lookahead, fallback: origSelect.call(this, model) });
            }
            return origSelect.call(this, model);
        };
    }
    // without the solver the only safe thing for useIncremental is to keep the engine out: callers that cannot hand over
    // the solver can still wrap Solve themselves with guardIncremental()
    return function uninstall() {
        P.simplex = orig.simplex;
        P.save = orig.save;
        P.restore = orig.restore;
        P.addCutConstraints = orig.addCutConstraints;
        P.applyMIRCuts = orig.applyMIRCuts;
        P.updateVariableValues = orig.updateVariableValues;
        P.getSolution = orig.getSolution;
        P.initialize = origInitialize;
        P.copy = origCopy;
        for (const name of Object.keys(origEditing)) P[name] = origEditing[name];
        if (opts.solver) {
            if (hadOwnSelect) opts.solver.selectBranchAndCutService = origSelect;
            else delete opts.solver.selectBranchAndCutService;
            opts.solver.Solve = origSolve;
        }
    };
}

// keep a useIncremental Solve entirely on the reference's CPU path (for hosts that install() without the solver)
function guardIncremental(solver) {
    const solve = solver.Solve;
    solver.Solve = function (model) {
        const keepOut = !!(model && model.options && model.options.useIncremental === true);
        if (!keepOut) return solve.apply(this, arguments);
        bypass += 1;
        try {
            return solve.apply(this, arguments);
        } finally {
            bypass -= 1;
        }
    };
    return function unguard() {
        solver.Solve = solve;
    };
}

// ---- device-resident checkpoints (incremental-branch-and-cut.ts:31-107) -----------------------------------------------
// createCheckpoint: the matrix and the maps stay in HBM; the host scalars of a StateCheckpoint ride along
function createCheckpoint(t) {
    const st = t.__gpu;
    if (!st || !st.active) throw new Error("[gpu-tableau] createCheckpoint: tableau is not on the engine");
    return {
        id: addon.checkpointCreate(st.h),
        height: t.height,
        nVars: t.nVars,
        lastElementIndex: t.lastElementIndex,
        availableIndexes: t.availableIndexes.slice(),
        evaluation: t.evaluation,
        feasible: t.feasible,
    };
}

// applyIncrementalCuts' fast path (:248-253): restoreCheckpoint + addCutConstraints(cuts) + simplex as ONE addon call
function relaxFromCheckpoint(t, cp, cuts) {
    const st = t.__gpu;
    t.height = cp.height; // restoreCheckpoint's host side (:82-106); matrix + maps are restored on the device
    t.nVars = cp.nVars;
    t.lastElementIndex = cp.lastElementIndex;
    t.availableIndexes = cp.availableIndexes.slice();
    t.evaluation = cp.evaluation;
    t.feasible = cp.feasible;
    t.varIndexByRow.length = t.height;
    st.pendingRestore = false;
    st.pendingCuts = null;
    t.addCutConstraints(cuts); // the override: slack bookkeeping, queues the cuts
    const c = packCuts(st.pendingCuts || []);
    st.pendingCuts = null;
    const check = t.model ? t.model.checkForCycles === true : false;
    const res = addon.relaxFrom(st.h, cp.id, c.type, c.varIndex, c.value, check, st.rhs, st.rows);
    absorb(t, st, res);
    return t;
}

// ---- batches of independent nodes (host/gpu-speculative-service.js) ----------------------------------------------------
// The per-node results of a batch come back PACKED (addon/jslp_napi.c batch_results: 10 int32 + 2 doubles per node) and become the
// objects the rest of this file reads here -- same fields, same key order as the addon's own result objects.  (Built through N-API
// they cost 12 property stores per node: tens of microseconds for a 16-node batch, comparable to the batch's kernel.)
const RES_I32 = 10, RES_F64 = 2;
function unpackResults(I, F, n) {
    const out = new Array(n);
    for (let i = 0, a = 0, b = 0; i < n; i++, a += RES_I32, b += RES_F64) {
        out[i] = {
            feasible: I[a] !== 0, bounded: I[a + 1] !== 0, optimal: I[a + 2] !== 0, unboundedVarIndex: I[a + 3],
            pivotsPhase1: I[a + 4], pivotsPhase2: I[a + 5], cyclePhase: I[a + 6], cycleStart: I[a + 7], cycleLength: I[a + 8],
            height: I[a + 9], objCell: F[b], evaluation: F[b + 1],
        };
    }
    return out;
}
// every node = restore() + addCutConstraints(cuts) + simplex() from the saved root, all nodes in ONE engine call
function relaxBatch(t, cutLists) {
    const st = t.__gpu;
    if (!st || !st.active) throw new Error("[gpu-tableau] relaxBatch: tableau is not on the engine");
    const n = cutLists.length;
    const offsets = new Int32Array(n + 1);
    let total = 0;
    for (let i = 0; i < n; i++) {
        total += cutLists[i].length;
        offsets[i + 1] = total;
    }
    const type = new Int8Array(total);
    const varIndex = new Int32Array(total);
    const value = new Float64Array(total);
    for (let i = 0, k = 0; i < n; i++) {
        const cuts = cutLists[i];
        for (let j = 0; j < cuts.length; j++, k++) {
            type[k] = cuts[j].type === "min" ? 0 : 1;
            varIndex[k] = cuts[j].varIndex;
            value[k] = cuts[j].value;
        }
    }
    const stride = st.rowCapacity;
    if (!st.batchRhs || st.batchRhs.length < n * stride) {
        st.batchRhs = new Float64Array(n * stride);
        st.batchRows = new Int32Array(n * stride);
    }
    const check = t.model ? t.model.checkForCycles === true : false;
    // install(..., { devices: [0, 1, ...] }): the batch is split over one engine per listed device (jslp_pool_*: the saved
    // root is fanned out over xGMI once per save(), every member has its own host thread and stream inside the library)
    const devices = installedOpts.devices;
    if (devices && devices.length > 1 && !st.pool) st.pool = addon.poolCreate(st.h, Int32Array.from(devices));
    const resI = new Int32Array(n * RES_I32), resF = new Float64Array(n * RES_F64);
    if (st.pool) addon.poolRelaxBatch(st.pool, offsets, type, varIndex, value, check, st.batchRhs, st.batchRows, stride, resI, resF);
    else addon.relaxBatch(st.h, offsets, type, varIndex, value, check, st.batchRhs, st.batchRows, stride, resI, resF);
    const results = unpackResults(resI, resF, n);
    const out = new Array(n);
    for (let i = 0; i < n; i++) {
        const H = results[i].height;
        // copies: outcomes are cached across batches, the staging arrays are not
        out[i] = { res: results[i], rhs: st.batchRhs.slice(i * stride, i * stride + H), rows: st.batchRows.slice(i * stride, i * stride + H) };
    }
    return out;
}

// relaxBatch with the compact read-back (jslp_engine_relax_batch_watched): per node only rowByVarIndex / the RHS cell of
// `varIndexes` (default: the model's integer variables) -- what isIntegral / getMostFractionalVar read between relaxations
// (mip-utils.ts:43-61, 100-126); ~10x fewer bytes over PCIe than the full RHS columns + row maps of relaxBatch
function relaxBatchWatched(t, cutLists, varIndexes) {
    const st = t.__gpu;
    if (!st || !st.active) throw new Error("[gpu-tableau] relaxBatchWatched: tableau is not on the engine");
    // install(..., { devices: [0, 1, ...] }): the compact read-back over the device pool too (jslp_pool_relax_batch_watched: every member's
    // outcomes in one pinned [nodes x watched] buffer; round 3's pool split full read-backs only -- 11.2 KB per Monster_II node per member)
    const devices = installedOpts.devices;
    if (devices && devices.length > 1 && !st.pool) st.pool = addon.poolCreate(st.h, Int32Array.from(devices));
    const setWatched = (w) => (st.pool ? addon.poolSetWatchedVariables(st.pool, w) : addon.setWatchedVariables(st.h, w));
    let watched;
    if (varIndexes) {
        watched = Int32Array.from(varIndexes);
        const key = Array.prototype.join.call(watched, ",");
        if (st.watchedKey !== key) {
            setWatched(watched);
            st.watchedKey = key;
        }
    } else {
        // the model's integer variables: the list is fixed once the tableau is built (registered once per engine, not re-derived and
        // re-compared for every batch of a tree: ~15 us per call, 31 calls in a Solve(Monster_II))
        watched = st.watchedDefault;
        if (!watched) watched = st.watchedDefault = Int32Array.from(t.model.integerVariables.map((v) => v.index));
        if (st.watchedKey !== "\u0000model") {  // (no list of indexes joins to this)
            setWatched(watched);
            st.watchedKey = "\u0000model";
        }
    }
    const n = cutLists.length, w = watched.length;
    const offsets = new Int32Array(n + 1);
    let total = 0;
    for (let i = 0; i < n; i++) {
        total += cutLists[i].length;
        offsets[i + 1] = total;
    }
    const type = new Int8Array(total);
    const varIndex = new Int32Array(total);
    const value = new Float64Array(total);
    for (let i = 0, k = 0; i < n; i++) {
        const cuts = cutLists[i];
        for (let j = 0; j < cuts.length; j++, k++) {
            type[k] = cuts[j].type === "min" ? 0 : 1;
            varIndex[k] = cuts[j].varIndex;
            value[k] = cuts[j].value;
        }
    }
    const rows = new Int32Array(n * w), values = new Float64Array(n * w);
    const check = t.model ? t.model.checkForCycles === true : false;
    const resI = new Int32Array(n * RES_I32), resF = new Float64Array(n * RES_F64);
    if (st.pool) addon.poolRelaxBatchWatched(st.pool, offsets, type, varIndex, value, check, rows, values, resI, resF);
    else addon.relaxBatchWatched(st.h, offsets, type, varIndex, value, check, rows, values, resI, resF);
    const results = unpackResults(resI, resF, n);
    const out = new Array(n);
    for (let i = 0; i < n; i++) out[i] = { res: results[i], rows: rows.subarray(i * w, (i + 1) * w), values: values.subarray(i * w, (i + 1) * w) };
    return out;
}

// make `t` the tableau of a node evaluated earlier by relaxBatch: the host bookkeeping of restore() + addCutConstraints(cuts)
// and then the cached outcome in place of simplex()
function commitOutcome(t, cuts, outcome) {
    const st = t.__gpu;
    t.restore();
    t.addCutConstraints(cuts);
    st.pendingRestore = false;
    st.pendingCuts = null;
    st.rhs.set(outcome.rhs);
    st.rows.set(outcome.rows);
    st.watchedCuts = null;
    absorb(t, st, outcome.res);
    return t;
}

// commitOutcome for a node evaluated by relaxBatchWatched: the host learns what the tree reads between relaxations -- flags,
// evaluation, and for every integer variable its row and value (mip-utils.ts:43-61, 100-126) -- and nothing else; whoever needs
// the whole column (getSolution, updateVariableValues, copy, the editing API) goes through flush(), which re-evaluates the node
function commitWatched(t, cuts, outcome) {
    const st = t.__gpu;
    t.restore();
    t.addCutConstraints(cuts);
    st.pendingRestore = false;
    st.pendingCuts = null;
    const res = outcome.res;
    t.feasible = res.feasible;
    t.bounded = res.bounded;
    if (res.optimal) {
        t.evaluation = res.evaluation;
        if (t.simplexIters === 0) t.bestPossibleEval = res.evaluation;
        t.simplexIters += 1;
    } else if (!res.bounded) {
        t.evaluation = -Infinity;
        t.unboundedVarIndex = res.unboundedVarIndex;
    }
    if (res.cyclePhase !== 0 && t.model) {
        t.model.messages.push("Cycle in phase " + res.cyclePhase);
        t.model.messages.push("Start :" + res.cycleStart);
        t.model.messages.push("Length :" + res.cycleLength);
    }
    const ints = t.model.integerVariables;
    const rows = outcome.rows, values = outcome.values;
    const width = t.width, rhsColumn = t.rhsColumn, matrix = t.matrix;
    const rowByVarIndex = t.rowByVarIndex, varIndexByRow = t.varIndexByRow;
    for (let k = 0; k < ints.length; k++) {
        const v = ints[k].index, r = rows[k];
        rowByVarIndex[v] = r;
        if (r > 0) {
            varIndexByRow[r] = v;
            matrix[r * width + rhsColumn] = values[k];
        }
    }
    st.lastHeight = res.height;
    st.stale = false;
    st.watchedCuts = cuts;  // (flush() completes the picture on demand)
    return t;
}

function releaseCheckpoint(t, cp) {
    const st = t.__gpu;
    if (st && st.active && cp && cp.id >= 0) {
        addon.checkpointRelease(st.h, cp.id);
        cp.id = -1;
    }
}

// Full read-back for `Solve(model, precision, full=true)` consumers and the post-solve editing API.
function sync(t) {
    const st = t.__gpu;
    if (!st || !st.active) return t;
    const d = addon.dims(st.h);
    const m = new Float64Array(d.height * d.width);
    const rows = new Int32Array(d.height);
    const cols = new Int32Array(d.width);
    const rbv = new Int32Array(d.nVarIndexes);
    const cbv = new Int32Array(d.nVarIndexes);
    addon.download(st.h, m, rows, cols, rbv, cbv);
    if (t.matrix.length < m.length || t.matrix === st.pinnedMatrix) t.matrix = new Float64Array(Math.max(m.length, st.rowCapacity * d.width));
    t.matrix.set(m);
    for (let r = 0; r < d.height; r++) t.varIndexByRow[r] = rows[r];
    for (let c = 0; c < d.width; c++) t.varIndexByCol[c] = cols[c];
    for (let v = 0; v < t.rowByVarIndex.length && v < d.nVarIndexes; v++) {
        t.rowByVarIndex[v] = rbv[v];
        t.colByVarIndex[v] = cbv[v];
    }
    st.stale = false;
    return t;
}

function pivotTrace(t) {
    const st = t.__gpu;
    return st && st.active ? addon.pivotTrace(st.h) : null;
}

// Solve() returned a simplified result: the engine goes back to the library's resource pool.  The tableau is finished --
// its host matrix was never kept in step with the device -- so any later simplex() on it throws instead of silently
// solving a stale copy; Solve(model, precision, true) / model.solve() keep the engine for the post-solve API.
function release(t) {
    const st = t.__gpu;
    if (t.__gpuEngine) { // built in pinned memory but never solved (e.g. presolve decided the model)
        const made = t.__gpuEngine;
        t.__gpuEngine = null;
        addon.detach(made.pinnedMatrix);
        addon.destroy(made.h);
        t.__gpuReleased = true;
    }
    if (st && st.active) {
        dropEngine(t, st, false);
        t.__gpuReleased = true;
        t.__gpu = undefined;
    }
}

// full read-back, then the tableau is an ordinary host tableau again (the next simplex() uploads it afresh)
function bringHome(t) {
    if (t.__gpuEngine) { // still being built in the pinned buffer: keep the contents, give the buffer back
        const made = t.__gpuEngine;
        t.__gpuEngine = null;
        t.matrix = new Float64Array(made.pinnedMatrix);
        addon.detach(made.pinnedMatrix);
        addon.destroy(made.h);
    }
    const st = t.__gpu;
    if (!st || !st.active) return;
    if (st.pendingRestore || st.pendingCuts) throw new Error("[gpu-tableau] tableau edited between restore()/addCutConstraints() and simplex()");
    flush(t);
    sync(t);
    dropEngine(t, st, true);
    t.__gpu = undefined;
}

function usesPool() {
    return !!(installedOpts.devices && installedOpts.devices.length > 1);
}

const api = {
    loadEngine, install, sync, pivotTrace, release, guardIncremental, createCheckpoint, relaxFromCheckpoint, releaseCheckpoint,
    relaxBatch, relaxBatchWatched, commitOutcome, commitWatched, usesPool, isOnEngine, bringHome, structuralNnz, stats,
    backend: () => backend,
};
module.exports = api;
