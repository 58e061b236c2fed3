"""Incremental branch-and-bound: children start from their PARENT's solved tableau instead of the root.

Mirrors the reference's `createIncrementalBranchAndCutService` (src/tableau/incremental-branch-and-cut.ts:130-499),
the service `Solve` picks for `options.useIncremental === true` (src/main.ts:62-72).  What it changes against the
default service is where a node starts: while the tree is walked depth-first every branched node leaves a
checkpoint (:440-445) and its two children are evaluated as restoreCheckpoint + ONE new cut + simplex (:248-253)
instead of restore(root) + the whole cut list (:254-258).  The reference keeps each checkpoint as a host copy of the
matrix (:55-70); here it is a device buffer (jslp_engine_checkpoint_*): a child costs one HBM-to-HBM copy, one
cut row and a handful of repair pivots, all in one engine call, and nothing crosses PCIe but the RHS column.

Host-side policy restated here because it decides which incumbent wins: depth-first stack / best-first heap and
the hybrid switch after the first incumbent (:295-299, 418-428), pseudocost branching with its order-dependent
update from `tableau.evaluation` (:339, 354-365, 172-221), the checkpoint budget (:132, 442).
"""
import math
import time

from .branch_and_cut import BranchMinHeap, _rows_by_var, is_integral, js_round, mir_loop


class _Branch:
    """IncrementalBranch (:49-53)"""
    __slots__ = ("relaxed", "cuts", "depth", "checkpoint", "new_cut")

    def __init__(self, relaxed, cuts, depth, checkpoint=None, new_cut=None):
        self.relaxed, self.cuts, self.depth, self.checkpoint, self.new_cut = relaxed, cuts, depth, checkpoint, new_cut


class _PseudoCosts:
    """:138-174"""

    def __init__(self):
        self.data = {}

    def _get(self, var_index):
        d = self.data.get(var_index)
        if d is None:
            d = self.data[var_index] = [0.0, 0, 0.0, 0]  # upSum, upCount, downSum, downCount
        return d

    def update(self, var_index, up, improvement, fraction):
        d = self._get(var_index)
        normalized = improvement / ((1 - fraction) if up else fraction)
        if up:
            d[0] += normalized
            d[1] += 1
        else:
            d[2] += normalized
            d[3] += 1

    def score(self, var_index, fraction):
        d = self._get(var_index)
        up = d[0] / d[1] if d[1] > 0 else 1
        down = d[2] / d[3] if d[3] > 0 else 1
        return max(up * (1 - fraction), 1e-6) * max(down * fraction, 1e-6)


def _select_branching_variable(model, rhs, rows, precision, branching, pseudo, strong_candidates=5):
    """selectBranchingVariable (incremental-branch-and-cut.ts:176-221; enhanced-branch-and-cut.ts:112-196, which adds the
    "strong" rule): (varIndex, value) or None"""
    candidates = []
    for var in model.integerVariables:
        r = rows.get(var["index"], -1)
        if r != -1:
            value = float(rhs[r])
            fraction = abs(value - js_round(value))
            if fraction > precision:
                candidates.append((var["index"], value, fraction))
    if not candidates:
        return None
    if branching == "most-fractional":  # stable sort by descending fraction, first element (:203-206)
        best = candidates[0]
        for c in candidates[1:]:
            if c[2] > best[2]:
                best = c
        return best[0], best[1]
    if branching == "strong":  # enhanced-branch-and-cut.ts:161-193: the 5 most fractional, scored by pseudocosts once both
        # directions of a variable have been seen twice, by fraction * (1 - fraction) until then
        ranked = sorted(candidates, key=lambda c: -c[2])[:strong_candidates]  # stable, like Array.prototype.sort
        best_score, best = -math.inf, ranked[0]
        for c in ranked:
            d = pseudo._get(c[0])
            score = pseudo.score(c[0], c[2]) if d[1] >= 2 and d[3] >= 2 else c[2] * (1 - c[2])
            if score > best_score:
                best_score, best = score, c
        return best[0], best[1]
    best_score, best = -math.inf, candidates[0]
    for c in candidates:
        score = pseudo.score(c[0], c[2])
        if score > best_score:
            best_score, best = score, c
    return best[0], best[1]


def enhanced_branch_and_cut(tableau, model, node_selection="hybrid", branching="pseudocost"):
    """createEnhancedBranchAndCutService (src/tableau/enhanced-branch-and-cut.ts:58-437), the service `Solve` picks for
    `options.nodeSelection` / `options.branching` (src/main.ts:74-80): the same tree walk as the incremental service
    with every node evaluated from the saved root (no checkpoints) and the "strong" branching rule."""
    return incremental_branch_and_cut(tableau, model, node_selection=node_selection, branching=branching, max_checkpoints=0,
                                      incremental_rules=False)


def incremental_branch_and_cut(tableau, model, node_selection="hybrid", branching="pseudocost", max_checkpoints=50,
                               incremental_rules=True):
    """branchAndCut (:283-495).  Leaves `tableau` holding the incumbent; returns (iterations, found_integral).
    `tableau.checkpoints_used` / `tableau.incremental_nodes` report how many checkpoints were taken and how many
    nodes started from one.  incremental_rules=False: the enhanced service's variant of the two places where the
    services differ (which branching rules exist; both update pseudocosts from the node's last cut)."""
    heap = BranchMinHeap()     # entries carry the _Branch in the `cuts` position
    stack = []
    iterations = 0
    checkpoint_count = 0
    incremental_nodes = 0
    tolerance = model.tolerance or 0
    tolerance_flag = True
    terminal_time = 1e99
    if model.timeout:
        terminal_time = time.time() * 1000.0 + model.timeout
    best_evaluation = math.inf
    best_branch = None
    found_integral = False
    check = model.checkForCycles
    precision = tableau.precision
    n_opt = getattr(tableau, "n_optional", 0)
    best_optional = [math.inf] * n_opt
    pseudo = _PseudoCosts()
    solutions_found = 0
    use_depth_first = node_selection in ("depth-first", "hybrid")
    checkpoints = []

    root = _Branch(-math.inf, [], 0)
    if use_depth_first:
        stack.append(root)
    else:
        heap.push(root.relaxed, root)

    while (len(stack) > 0 if use_depth_first else len(heap) > 0) and tolerance_flag and time.time() * 1000.0 < terminal_time:
        if model.isMinimization:
            acceptable = tableau.bestPossibleEval * (1 + tolerance)
        else:
            acceptable = tableau.bestPossibleEval * (1 - tolerance)
        if tolerance > 0 and best_evaluation < acceptable:
            tolerance_flag = False
        if use_depth_first and stack:
            branch = stack.pop()
        elif len(heap) > 0:
            branch = heap.pop()[2]
        else:
            break
        if branch.relaxed > best_evaluation:
            continue
        parent_eval = tableau.evaluation
        # applyIncrementalCuts (:246-259)
        if branch.checkpoint is not None and branch.new_cut is not None:
            (res, rhs, vibr), = tableau.applyCutsFrom(branch.checkpoint, [[branch.new_cut]], check_cycles=check)
            tableau.absorb_from(branch.checkpoint, res)
            incremental_nodes += 1
        else:
            _res, rhs, vibr = tableau.applyCuts(branch.cuts, check_cycles=check)
        rhs, vibr = mir_loop(tableau, model, rhs, vibr, check, max_rounds=3, need_feasible=True)  # :261-276
        iterations += 1
        if not tableau.feasible:
            continue
        evaluation = tableau.evaluation
        if evaluation > best_evaluation:
            continue
        # :354-365; the enhanced service takes the node's LAST cut instead (enhanced-branch-and-cut.ts:299-311), which also
        # exists for nodes that came through the best-first heap
        observed = branch.new_cut if incremental_rules else (branch.cuts[-1] if branch.cuts else None)
        if observed is not None and parent_eval != 0:
            pseudo.update(observed["varIndex"], observed["type"] == "min", abs(evaluation - parent_eval), 0.5)
        optional_cells = None
        if evaluation == best_evaluation:  # :367-388
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
        rows = _rows_by_var(vibr)
        if is_integral(model, rhs, rows, precision):
            found_integral = True
            solutions_found += 1
            if iterations == 1:
                tableau.checkpoints_used, tableau.incremental_nodes = 0, 0
                return iterations, True
            best_branch = branch
            best_evaluation = evaluation
            if n_opt > 0:
                if optional_cells is None:
                    optional_cells = tableau.optional_objectives()[:, 0]
                best_optional = [float(x) for x in optional_cells]
            if node_selection == "hybrid" and solutions_found >= 1:  # :418-428
                use_depth_first = False
                while stack:
                    b = stack.pop()
                    heap.push(b.relaxed, b)
        else:
            if iterations == 1:
                tableau.save()
            rule = branching if (not incremental_rules or branching == "most-fractional") else "pseudocost"  # (:203-221)
            sel = _select_branching_variable(model, rhs, rows, precision, rule, pseudo)
            if sel is None:
                continue
            var_index, var_value = sel
            checkpoint = None
            if use_depth_first and checkpoint_count < max_checkpoints:  # :440-445
                checkpoint = tableau.createCheckpoint()
                checkpoints.append(checkpoint)
                checkpoint_count += 1
            cuts_high, cuts_low = [], []
            for cut in branch.cuts:
                if cut["varIndex"] == var_index:
                    if cut["type"] == "min":
                        cuts_low.append(cut)
                    else:
                        cuts_high.append(cut)
                else:
                    cuts_high.append(cut)
                    cuts_low.append(cut)
            cut_high = {"type": "min", "varIndex": var_index, "value": float(math.ceil(var_value))}
            cut_low = {"type": "max", "varIndex": var_index, "value": float(math.floor(var_value))}
            cuts_high.append(cut_high)
            cuts_low.append(cut_low)
            depth = branch.depth + 1
            if use_depth_first:
                stack.append(_Branch(evaluation, cuts_low, depth, checkpoint, cut_low))
                stack.append(_Branch(evaluation, cuts_high, depth, checkpoint, cut_high))
            else:  # best-first nodes start from the root (:485-488)
                b_high, b_low = _Branch(evaluation, cuts_high, depth), _Branch(evaluation, cuts_low, depth)
                heap.push(b_high.relaxed, b_high)
                heap.push(b_low.relaxed, b_low)
    if best_branch is not None:
        _res, rhs, vibr = tableau.applyCuts(best_branch.cuts, check_cycles=check)  # :491-493, always from the root
        mir_loop(tableau, model, rhs, vibr, check, max_rounds=3, need_feasible=True)  # :228-243
    for c in checkpoints:
        tableau.releaseCheckpoint(c)
    tableau.checkpoints_used, tableau.incremental_nodes = checkpoint_count, incremental_nodes
    return iterations, found_integral
