"""Host-side branch-and-bound tree (CPU), calling the engine for every LP relaxation.

Mirrors the reference's DEFAULT BranchAndCutService (src/tableau/branch-and-cut.ts:54-199) -- best-first
with LIFO ties (src/tableau/min-heap.ts:43-49), most-fractional branching (src/tableau/mip-utils.ts:100-126)
-- because its visiting order decides which incumbent wins.  The three bulk operations of a node
(restore + addCutConstraints + simplex, branch-and-cut.ts:33-37) are ONE engine call (Tableau.applyCuts).
"""
import math
import os
import time

import numpy as np


def js_round(x):
    """Math.round: nearest integer, ties toward +Infinity"""
    if math.isinf(x) or math.isnan(x):
        return x
    f = math.floor(x)
    return f + 1.0 if x - f >= 0.5 else float(f)


class BranchMinHeap:
    """src/tableau/min-heap.ts: key = relaxedEvaluation ascending, ties -> most recently pushed first."""

    def __init__(self):
        self.heap = []
        self.seq = 0

    def __len__(self):
        return len(self.heap)

    @staticmethod
    def _before(a, b):
        if a[0] != b[0]:
            return a[0] < b[0]
        return a[1] > b[1]

    def push(self, relaxed_evaluation, cuts):
        entry = (relaxed_evaluation, self.seq, cuts)
        self.seq += 1
        h = self.heap
        h.append(entry)
        i = len(h) - 1
        while i > 0:
            p = (i - 1) >> 1
            if not self._before(entry, h[p]):
                break
            h[i] = h[p]
            i = p
        h[i] = entry

    def pop(self):
        h = self.heap
        top = h[0]
        last = h.pop()
        n = len(h)
        if n == 0:
            return top
        i = 0
        half = n >> 1
        while i < half:
            c = 2 * i + 1
            if c + 1 < n and self._before(h[c + 1], h[c]):
                c += 1
            if not self._before(h[c], last):
                break
            h[i] = h[c]
            i = c
        h[i] = last
        return top


class _RowMap:
    """rowByVarIndex as the host tree reads it: variable index -> row of the final tableau (-1 = not basic)"""
    __slots__ = ("row_of",)

    def __init__(self, vibr):
        v = np.asarray(vibr[1:], dtype=np.int64)
        self.row_of = np.full(int(v.max()) + 2 if v.size else 1, -1, dtype=np.int64)
        self.row_of[v] = np.arange(1, v.size + 1)

    def get(self, index, default=-1):
        if 0 <= index < self.row_of.shape[0]:
            r = int(self.row_of[index])
            return r if r != -1 else default
        return default

    def rows(self, indexes):
        """rows of several variable indexes at once (-1 = not basic)"""
        out = np.full(indexes.shape[0], -1, dtype=np.int64)
        ok = indexes < self.row_of.shape[0]
        out[ok] = self.row_of[indexes[ok]]
        return out


def _rows_by_var(vibr):
    return _RowMap(vibr)


def _js_round_vec(x):
    f = np.floor(x)
    return np.where(x - f >= 0.5, f + 1.0, f)


class _WatchedRows:
    """the compact outcome of a node (jslp_engine_relax_batch_watched*): rowByVarIndex and the RHS cell of the model's integer
    variables, in model.integerVariables order -- everything isIntegral / getMostFractionalVar read (mip-utils.ts:43-61, 100-126)"""
    __slots__ = ("wrows", "wvals")

    def __init__(self, wrows, wvals):
        self.wrows, self.wvals = wrows, wvals


def _integer_values(model, rhs, rows):
    """(variable indexes, values) of the integer variables that are basic, in model.integerVariables order"""
    idx = model.integer_index_array
    if isinstance(rows, _WatchedRows):
        basic = rows.wrows != -1
        return idx[basic], np.asarray(rows.wvals, dtype=np.float64)[basic]
    r = rows.rows(idx)
    basic = r != -1
    return idx[basic], np.asarray(rhs, dtype=np.float64)[r[basic]]


def is_integral(model, rhs, rows, precision):
    """mip-utils.ts:43-61"""
    _idx, values = _integer_values(model, rhs, rows)
    return not bool((np.abs(values - _js_round_vec(values)) > precision).any())


def most_fractional_var(model, rhs, rows):
    """mip-utils.ts:100-126: first variable with the strictly biggest |v - round(v)|"""
    idx, values = _integer_values(model, rhs, rows)
    if values.size == 0:
        return None, 0.0
    fraction = np.abs(values - _js_round_vec(values))
    fraction = np.where(np.isnan(fraction), -1.0, fraction)  # `fraction > biggest` is false for NaN
    k = int(np.argmax(fraction))  # first index of the maximum
    if not fraction[k] > 0.0:
        return None, 0.0
    return int(idx[k]), float(values[k])


def fractional_volume(model, rhs, vibr, precision):
    """computeFractionalVolume(ignoreIntegerValues = true) (mip-utils.ts:67-92): the product, in row order, of |value|
    over the basic integer variables that are not at an integer value"""
    integer = model.integer_index_set
    volume = -1.0
    for r in range(1, len(vibr)):
        if int(vibr[r]) in integer:
            distance = abs(float(rhs[r]))
            if min(distance - math.floor(distance), math.floor(distance + 1)) < precision:
                continue
            volume = distance if volume == -1.0 else volume * distance
    return 0.0 if volume == -1.0 else volume


def mir_loop(tableau, model, rhs, vibr, check, max_rounds=None, need_feasible=False):
    """The MIR loop that follows the simplex() of a node: branch-and-cut.ts:38-52 (unbounded, no feasibility test) or
    incremental-branch-and-cut.ts:228-243, 261-276 (at most 3 rounds, feasible tableaus only).  Every round is one
    engine call (applyMIRCuts + simplex + read-back); the stopping rule needs only the RHS column."""
    if not model.useMIRCuts or (need_feasible and not tableau.feasible):
        return rhs, vibr
    rounds = 0
    while max_rounds is None or rounds < max_rounds:
        before = fractional_volume(model, rhs, vibr, tableau.precision)
        _n, _res, rhs, vibr = tableau.mirRound(check_cycles=check)
        after = fractional_volume(model, rhs, vibr, tableau.precision)
        rounds += 1
        if after >= 0.9 * before:
            break
    return rhs, vibr


class _NodeEval:
    """outcome of one LP relaxation as the host tree consumes it"""
    __slots__ = ("res", "rhs", "vibr")

    def __init__(self, res, rhs, vibr):
        self.res, self.rhs, self.vibr = res, rhs, vibr


class _NodeEvalWatched:
    """the same from a COMPACT outcome: the integer variables' rows and values only (the full column of the one leaf the tree commits
    to is re-evaluated at the end, as host/gpu-tableau.js's flush() does)"""
    __slots__ = ("res", "wrows", "wvals")

    def __init__(self, res, wrows, wvals):
        self.res, self.wrows, self.wvals = res, wrows, wvals


EVAL_STATS = {"seconds": 0.0, "batches": 0, "nodes": 0}  # speculative batches since the caller last zeroed it (bench.py)


def branch_and_cut(tableau, model, speculate=1, evaluate_batch=None):
    """branch-and-cut.ts:54-199.  Leaves `tableau` holding the incumbent; returns the iteration count and
    whether an integral node was accepted (tableau.__isIntegral).

    speculate > 1 turns on SPECULATIVE EVALUATION WITH IN-ORDER COMMIT: whenever the sequential
    loop needs a node that has not been evaluated yet, that node and the next best `speculate - 1` heap entries
    are evaluated as ONE batch of independent relaxations (every node is a pure function of the saved root and
    its cut list); the loop itself still pops, prunes and commits in the reference's order, so the incumbent,
    the iteration count and the result are those of the sequential run.  `evaluate_batch(cut_lists)` may shard
    the batch over several GPUs (jslpsolver_amd.sharding); default: the tableau's own applyCutsBatch.
    """
    branches = BranchMinHeap()
    iterations = 0
    tolerance = model.tolerance or 0
    tolerance_flag = True
    terminal_time = 1e99
    if model.timeout:
        terminal_time = time.time() * 1000.0 + model.timeout
    best_evaluation = math.inf
    best_cuts = None
    found_integral = False
    check = model.checkForCycles
    precision = tableau.precision
    n_opt = getattr(tableau, "n_optional", 0)
    best_optional = [math.inf] * n_opt  # bestOptionalObjectivesEvaluations (:66-69)
    if n_opt > 0 or model.useMIRCuts:
        speculate = 1  # the tie-break below reads the live optional-objective cells / the MIR loop is driven per node
    if evaluate_batch is None and tableau.width * tableau.height0 > 1536 * 1024:
        speculate = 1  # one workgroup per node only pays for small tableaus; big ones go node by node through the chip-wide kernels
    cache = {}          # heap sequence number -> _NodeEval
    # (JSLP_TREE_COMPACT=0: whole columns per speculated node, as before round 5)
    # (the engine takes at most row_capacity watched variables -- a model with more integers than rows + cut capacity keeps the
    #  whole-column read-back instead of failing on its first speculative batch)
    compact_ok = (evaluate_batch is None and os.environ.get("JSLP_TREE_COMPACT", "1") != "0" and hasattr(tableau, "applyCutsBatchWatched")
                  and 0 < len(model.integer_index_array) <= getattr(tableau, "row_capacity", 0))
    watched_set = []
    watched_before = list(getattr(tableau, "watched", []) or [])  # the caller's own registration, put back when the tree is done
    saved = False
    last_cuts = None    # cuts of the node the sequential run evaluated last
    speculated = 0

    def run_batch(cut_lists):
        # (wall time of the evaluation share of the tree -- the engine calls, and for a sharded tree the exchange step --, so that
        #  a strong-scaling run can be read: bench.py's `tree.eval_ms` / `tree.host_ms`)
        t0 = time.perf_counter()
        try:
            if evaluate_batch is not None:
                return evaluate_batch(cut_lists)
            if compact_ok:
                # the compact read-back (jslp_engine_relax_batch_watched): per node the integer variables' rows and values -- what the loop
                # below reads (mip-utils.ts:43-61, 100-126) -- instead of whole RHS columns + row maps; the leaf the tree commits to is
                # evaluated once more with the full read-back at the end (tableau.applyCuts(best_cuts))
                if not watched_set:
                    tableau.set_watched_variables(model.integer_index_array)
                    watched_set.append(True)
                results, wrows, wvals = tableau.applyCutsBatchWatched(cut_lists, check_cycles=check)
                return [_NodeEvalWatched(results[i], wrows[i].copy(), wvals[i].copy()) for i in range(len(cut_lists))]
            results, rhs, vibr = tableau.applyCutsBatch(cut_lists, check_cycles=check)
            return [_NodeEval(results[i], rhs[i, :results[i].height].copy(), vibr[i, :results[i].height].copy())
                    for i in range(len(cut_lists))]
        finally:
            EVAL_STATS["seconds"] += time.perf_counter() - t0
            EVAL_STATS["batches"] += 1
            EVAL_STATS["nodes"] += len(cut_lists)

    branches.push(-math.inf, [])
    while len(branches) > 0 and tolerance_flag and time.time() * 1000.0 < terminal_time:
        if model.isMinimization:
            acceptable = tableau.bestPossibleEval * (1 + tolerance)
        else:
            acceptable = tableau.bestPossibleEval * (1 - tolerance)
        if tolerance > 0 and best_evaluation < acceptable:
            tolerance_flag = False
        relaxed, seq, cuts = branches.pop()
        if relaxed > best_evaluation:
            continue
        if speculate > 1 and saved:
            if seq not in cache:
                # this node + the entries the heap would hand out next (best first, LIFO ties); nodes already
                # prunable by the current incumbent are not worth speculating on
                ahead = sorted(branches.heap, key=lambda e: (e[0], -e[1]))
                batch = [(seq, cuts)] + [(e[1], e[2]) for e in ahead if e[1] not in cache and e[0] <= best_evaluation][:speculate - 1]
                for (sq, _), ev in zip(batch, run_batch([c for _, c in batch])):
                    cache[sq] = ev
                speculated += len(batch)
            ev = cache.pop(seq)
            tableau._absorb(ev.res)
            if isinstance(ev, _NodeEvalWatched):
                rhs, vibr = None, _WatchedRows(ev.wrows, ev.wvals)
            else:
                rhs, vibr = ev.rhs, ev.vibr
        else:
            _res, rhs, vibr = tableau.applyCuts(cuts, check_cycles=check)
            rhs, vibr = mir_loop(tableau, model, rhs, vibr, check)  # :38-52
        last_cuts = cuts
        iterations += 1
        if not tableau.feasible:
            continue
        evaluation = tableau.evaluation
        if evaluation > best_evaluation:
            continue
        optional_cells = None
        if evaluation == best_evaluation:  # :107-127: ties are broken on the optional objectives, in priority order
            worse = True
            if n_opt > 0:
                optional_cells = tableau.optional_objectives()[:, 0]
                for o in range(n_opt):
                    if optional_cells[o] > best_optional[o]:
                        break
                    if optional_cells[o] < best_optional[o]:
                        worse = False
                        break
            if worse:
                continue
        rows = vibr if isinstance(vibr, _WatchedRows) else _rows_by_var(vibr)
        if is_integral(model, rhs, rows, precision):
            found_integral = True
            if iterations == 1:
                return iterations, True
            best_cuts = cuts
            best_evaluation = evaluation
            if n_opt > 0:
                if optional_cells is None:
                    optional_cells = tableau.optional_objectives()[:, 0]
                best_optional = [float(x) for x in optional_cells]
        else:
            if iterations == 1:
                tableau.save()
                saved = True
            var_index, var_value = most_fractional_var(model, rhs, rows)
            cuts_high, cuts_low = [], []
            for cut in cuts:
                if cut["varIndex"] == var_index:
                    if cut["type"] == "min":
                        cuts_low.append(cut)
                    else:
                        cuts_high.append(cut)
                else:
                    cuts_high.append(cut)
                    cuts_low.append(cut)
            cuts_high.append({"type": "min", "varIndex": var_index, "value": float(math.ceil(var_value))})
            cuts_low.append({"type": "max", "varIndex": var_index, "value": float(math.floor(var_value))})
            branches.push(evaluation, cuts_high)
            branches.push(evaluation, cuts_low)
    if best_cuts is not None:
        _res, rhs, vibr = tableau.applyCuts(best_cuts, check_cycles=check)
        mir_loop(tableau, model, rhs, vibr, check)
    elif speculate > 1 and saved and last_cuts is not None:
        # no incumbent: the reference's tableau is left on the node it evaluated last; put ours there too
        tableau.applyCuts(last_cuts, check_cycles=check)
    if watched_set and watched_before != list(model.integer_index_array):
        tableau.set_watched_variables(watched_before)  # (the caller's registration was borrowed for the compact read-back)
    tableau.speculated_nodes = speculated
    return iterations, found_integral
