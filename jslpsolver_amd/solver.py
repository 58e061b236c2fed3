"""`Solve(model)` with the reference's JSON model API and result shape (src/main.ts:94-193), the simplex
hot path running on the MI355X engine.  Model parsing and the branch-and-bound tree stay on the CPU.
"""
import math
import warnings

from . import _capi
from .branch_and_cut import branch_and_cut, js_round
from .engine import Tableau
from .model import Model, js_keys

EPSILON = 2.220446049250313e-16
_presolve_warned = False


def _round_value(v, rounding_coeff):
    """Math.round((Number.EPSILON + v) * rc) / rc (solution.ts:55-56)"""
    return js_round((EPSILON + v) * rounding_coeff) / rounding_coeff


def Solve(model, precision=None, full=False, validate=False, lib=None, device=0, row_capacity_extra=None,
          speculate=1, group=None):
    """Drop-in for `solver.Solve(model, precision, full, validate)`.

    speculate > 1: evaluate up to that many branch-and-bound nodes per engine call (speculation with in-order
    commit: identical results, see branch_and_cut); `group`: a torch.distributed process group over which each
    batch is sharded, one GPU per rank (every rank calls Solve with the same model and gets the same result).

    `lib` selects the engine library (default: the HIP product library; tests pass the CPU oracle to exercise
    the host logic without a GPU).  Returns the simplified result dict, or -- with full=True -- a dict that
    also carries the final tableau read-back.

    Differences from the reference host, by design (this Python host exists for pytest and bench.py; the drop-in host
    is the reference's own code under host/gpu-tableau.js): the presolve pre-pass (src/tableau/presolve.ts, on by
    default for models with integer variables) is NOT run -- it can fix variables or declare infeasibility before any
    simplex, so results may differ on models it touches (a UserWarning says so once); `options.keep_solutions` is
    rejected (UnsupportedModel).
    """
    global _presolve_warned
    if model is None:
        raise ValueError("Solver requires a model to operate on")  # main.ts:110-112
    m = Model(model, precision)
    if m.keep_solutions:
        from .model import UnsupportedModel
        raise UnsupportedModel("options.keep_solutions is not collected by the Python host (use the reference host + binding)")
    if m.usePresolve and len(m.integerVariables) > 0 and not _presolve_warned:
        _presolve_warned = True
        warnings.warn("jslpsolver_amd.Solve does not run the reference's presolve pre-pass (src/tableau/presolve.ts): on models "
                      "where presolve fixes variables or proves infeasibility the result can differ from solver.Solve(); "
                      "pass options.presolve = false to compare like with like, or use the reference host + binding", stacklevel=2)
    matrix, vibr, vibc = m.build_tableau()
    n_int = len(m.integerVariables)
    # cut rows: at most one "min" and one "max" cut per integer variable (branch-and-cut.ts:166-179)
    extra = 2 * n_int if row_capacity_extra is None else row_capacity_extra
    incremental = n_int > 0 and (m.options.get("useIncremental") is True)
    if incremental and row_capacity_extra is None:
        # the incremental service stacks ONE row per tree level on the parent's tableau and never merges cuts on the
        # same variable (incremental-branch-and-cut.ts:248-253), so its height is bounded by the depth of the tree, not
        # by the number of integer variables; the reference reallocates (cutting-strategies.ts:24-30), device memory
        # is sized once -- a deeper tree fails loudly with JSLP_ERR_CAPACITY (pass row_capacity_extra)
        extra = 2 * n_int + 256
    if n_int > 0 and m.useMIRCuts and row_capacity_extra is None:
        # every MIR round appends up to 10 rows (cutting-strategies.ts:199-212); the default service's loop has no bound
        # but a 10 % volume gain per round (branch-and-cut.ts:38-52): 32 rounds of headroom, loud failure beyond
        extra += 320
    _priorities, optional_rows = m.optional_objectives()
    t = Tableau(matrix, vibr, vibc, m.unrestricted, precision=m.precision, row_capacity=matrix.shape[0] + extra,
                device=device, lib=lib, optional_objectives=optional_rows,
                integer_variables=[v["index"] for v in m.integerVariables] if m.useMIRCuts else None)
    try:
        return _solve_on(t, m, n_int, incremental, speculate, group, full)
    finally:
        t.close()  # also on errors (JSLP_ERR_CAPACITY ...): the engine goes back to the library's resource pool


def _solve_on(t, m, n_int, incremental, speculate, group, full):
    iterations = 0
    integral = False
    if n_int > 0:  # tableau.ts:250-258
        evaluate = None
        if group is not None:
            from .sharding import make_sharded_evaluator
            evaluate = make_sharded_evaluator(t, m.checkForCycles, group, watched=m.integer_index_array)
        if incremental:  # selectBranchAndCutService (main.ts:62-72)
            from .incremental_branch_and_cut import incremental_branch_and_cut
            iterations, integral = incremental_branch_and_cut(
                t, m, node_selection=m.options.get("nodeSelection") or "hybrid",
                branching=m.options.get("branching") or "pseudocost")
        elif m.options.get("nodeSelection") or m.options.get("branching"):  # main.ts:74-80
            from .incremental_branch_and_cut import enhanced_branch_and_cut
            iterations, integral = enhanced_branch_and_cut(
                t, m, node_selection=m.options.get("nodeSelection") or "hybrid",
                branching=m.options.get("branching") or "pseudocost")
        else:
            iterations, integral = branch_and_cut(t, m, speculate=speculate, evaluate_batch=evaluate)
    else:
        t.simplex(check_cycles=m.checkForCycles)
    rhs, rows = t.read_rhs()
    evaluation = t.evaluation if m.isMinimization else -t.evaluation  # tableau.ts:261
    result = {"feasible": t.feasible, "result": evaluation, "bounded": t.bounded}
    if integral:
        result["isIntegral"] = True
    # generateSolutionSet (solution.ts:35-60) + buildSimplifiedResult (main.ts:173-193)
    by_index = {v["index"]: v for v in m.variables}
    rc = js_round(1 / m.precision)
    solution_set = {}
    for r in range(1, len(rows)):
        var = by_index.get(int(rows[r]))
        if var is None:
            continue
        solution_set[var["id"]] = _round_value(float(rhs[r]), rc)
    for vid in js_keys(solution_set):
        if solution_set[vid] != 0:
            result[vid] = solution_set[vid]
    result = {k: result[k] for k in js_keys(result)}  # JS property order: integer-like ids first
    if full:
        fm, fvibr, fvibc, _, _ = t.download()
        result = {"result": result, "solutionSet": solution_set, "matrix": fm, "varIndexByRow": fvibr,
                  "varIndexByCol": fvibc, "iter": iterations, "pivots": t.pivot_trace(), "model": m,
                  "checkpoints": getattr(t, "checkpoints_used", 0), "incrementalNodes": getattr(t, "incremental_nodes", 0)}
    return result
