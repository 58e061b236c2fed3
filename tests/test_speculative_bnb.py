"""Speculative batched branch-and-bound (in-order commit) must reproduce the sequential reference run."""
import pytest

import golden_util as G
from jslpsolver_amd import Model, Solve, UnsupportedModel
from test_host_solve import check_fixture


def _mips():
    out = []
    for p in G.fixture_paths() + G.synthetic_paths():
        g = G.load(p)
        if g.get("model") and g["tableau"] and g["tableau"]["integerVarIndexes"] and not g["tableau"]["optionalObjectives"]:
            out.append(p)
    return out


@pytest.mark.parametrize("path", _mips(), ids=G.ident)
@pytest.mark.parametrize("spec", [2, 8])
def test_speculation_gives_the_sequential_result(oracle_lib, path, spec):
    g = G.load(path)
    model = g["model"]
    if g["presolve"] and g["presolve"]["nFixed"] > 0:
        pytest.skip("reference presolve fixed variables")
    if model.get("timeout") or (model.get("options") or {}).get("timeout"):
        if "LargeFarm" in path:
            pytest.skip("wall-clock bounded run")
    try:
        Model(model)
    except UnsupportedModel as e:
        pytest.skip(str(e))
    seq = Solve(model, full=True, lib=oracle_lib)
    par = Solve(model, full=True, lib=oracle_lib, speculate=spec)
    assert par["result"] == seq["result"] and list(par["result"]) == list(seq["result"])
    assert par["iter"] == seq["iter"]
    assert par["matrix"].tobytes() == seq["matrix"].tobytes()
    ref = {k: (G.num(v) if not isinstance(v, bool) else v) for k, v in g["result"].items()}
    assert par["result"] == ref


def test_pinned_batch_view_equals_copying_batch(oracle_lib):
    """jslp_engine_relax_batch_pinned (views of the engine's read-back buffer) == jslp_engine_relax_batch"""
    import os
    import numpy as np
    from jslpsolver_amd.engine import Tableau
    g = G.load(os.path.join(G.GOLDEN, "fixtures", "Knapsack_1.json.gz"))
    tab = g["tableau"]
    m, vibr, vibc = G.dense_tableau(tab)
    calls = g["simplexCalls"]
    t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"],
                row_capacity=tab["height"] + max(len(c["cuts"] or []) for c in calls), lib=oracle_lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    nodes = [c["cuts"] or [] for c in calls[1:40]]
    res_a, rhs_a, rows_a = t.applyCutsBatch(nodes)
    packed = t.pack_cut_lists(nodes)
    res_b, rhs_b, rows_b = t.applyCutsBatch(None, packed=packed, copy=False)
    for i in range(len(nodes)):
        h = res_a[i].height
        assert res_a[i].as_dict() == res_b[i].as_dict()
        assert np.array_equal(rhs_a[i, :h].view(np.uint64), rhs_b[i, :h].view(np.uint64))
        assert np.array_equal(rows_a[i, :h], rows_b[i, :h])
    t.close()


def test_more_integer_variables_than_rows_keeps_the_whole_column_read_back(oracle_lib):
    """the compact read-back takes at most row_capacity watched variables (jslp_engine_set_watched_variables); a model with more
    integer variables than rows + cut capacity must still run its speculative batches (whole columns), give the sequential result -- and a
    watched list the caller registered on the tableau survives a tree that borrows the registration"""
    prof = [19, 11, 27, 17, 6, 21, 26, 27, 28, 14, 30, 12, 4, 16]
    wt = [3, 5, 7, 10, 11, 7, 4, 7, 6, 13, 10, 3, 6, 12]
    model = {"optimize": "profit", "opType": "max", "constraints": {"weight": {"max": 65}},
             "variables": {"x%d" % i: {"profit": prof[i], "weight": wt[i]} for i in range(14)},
             "ints": {"x%d" % i: 1 for i in range(14)}, "options": {"presolve": False}}
    seq = Solve(model, full=True, lib=oracle_lib)
    par = Solve(model, full=True, lib=oracle_lib, speculate=4, row_capacity_extra=11)  # 2 rows + 11 cut rows = 13 < 14 integer variables
    assert seq["iter"] == 11 and par["result"] == seq["result"] and par["iter"] == seq["iter"]
    # the engine-level bound itself, and the borrowed registration
    import numpy as np
    from jslpsolver_amd import _capi
    from jslpsolver_amd.branch_and_cut import branch_and_cut
    from jslpsolver_amd.engine import Tableau
    m = Model(model)
    matrix, vibr, vibc = m.build_tableau()
    n_int = len(m.integer_index_array)
    t = Tableau(matrix, vibr, vibc, m.unrestricted, precision=m.precision, row_capacity=matrix.shape[0] + 2 * n_int, lib=oracle_lib)
    mine = [int(m.integer_index_array[0])]
    t.set_watched_variables(mine)
    it, _ = branch_and_cut(t, m, speculate=4)
    assert it == seq["iter"] and t.watched == mine and t.watched_count() == 1
    t.close()
    small = Tableau(matrix, vibr, vibc, m.unrestricted, precision=m.precision, row_capacity=n_int - 1, lib=oracle_lib)
    small.applyCuts([], check_cycles=True)
    small.save()
    small.set_watched_variables(list(m.integer_index_array))
    with pytest.raises(_capi.EngineError):  # (what the tree would have run into before the guard)
        small.applyCutsBatchWatched([[]], check_cycles=True)
    small.close()
