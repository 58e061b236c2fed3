"""Pins the C restatement (oracle/jslp_oracle.c) against the reference's own behaviour.

Golden vectors come from the reference itself (type-erased, oracle/_ref) -- see tests/golden/gen_golden.js.
For every fixture that reaches the hot path (soft-constraint models with their optional objectives included)
the oracle must reproduce, bit for bit: every pivot (row, col) in order, every simplex
call's flags / evaluation / RHS column, and the final tableau.
"""
import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd.engine import Tableau, pivot_digest


def replay(lib, g):
    """Drive the engine through the exact call sequence the reference made and compare each step."""
    tab = g["tableau"]
    m, vibr, vibc = G.dense_tableau(tab)
    assert G.sha_matrix(m) == tab["matrixSha"]
    calls = g["simplexCalls"]
    max_cuts = max([len(c["cuts"] or []) for c in calls] + [0])
    oo = None
    if tab["optionalObjectives"]:
        oo = np.array([[G.num(x) for x in o["reducedCosts"]] + [0.0] * (tab["width"] - len(o["reducedCosts"]))
                       for o in tab["optionalObjectives"]], dtype=np.float64)
    t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"],
                row_capacity=tab["height"] + max_cuts, lib=lib, optional_objectives=oo)
    check = tab["checkForCycles"]
    is_mip = len(tab["integerVarIndexes"]) > 0
    for i, call in enumerate(calls):
        if is_mip:
            res, rhs, rows = t.applyCuts(call["cuts"] or [], check_cycles=check)
        else:
            res = t.simplex(check_cycles=check)
            rhs, rows = t.read_rhs()
        where = "simplex call %d" % i
        assert bool(res.feasible) == call["feasible"], where
        assert bool(res.bounded) == call["bounded"], where
        assert res.pivots_phase1 == call["p1"], where
        assert res.pivots_phase2 == call["p2"], where
        assert res.height == call["height"], where
        assert G.sha_rhs(rhs, rows) == call["rhsSha"], where
        ev = G.num(call["evaluation"])
        assert t.evaluation == ev or (np.isnan(ev) and np.isnan(t.evaluation)), where
        if is_mip and i == g["savedAfterCall"]:
            t.save()
    trace = t.pivot_trace()
    assert len(trace) == g["nPivots"]
    assert pivot_digest(trace) == g["pivotDigest"]
    assert trace.reshape(-1).tolist() == g["pivots"][:2 * len(trace)]
    fm, fvibr, _, _, _ = t.download()
    fin = g["final"]
    assert fm.shape == (fin["height"], fin["width"])
    assert fvibr.tolist() == [-1 if v is None else v for v in fin["varIndexByRow"]]
    assert G.sha_matrix(fm) == fin["matrixSha"]
    t.close()


def usable(g):
    return g["tableau"] is not None and not g["tableau"]["useMIRCuts"]


@pytest.mark.parametrize("path", G.fixture_paths(), ids=G.ident)
def test_oracle_reproduces_reference_fixture(oracle_lib, path):
    g = G.load(path)
    if not usable(g):
        pytest.skip("presolve-infeasible model: the reference never reached the hot path")
    replay(oracle_lib, g)


@pytest.mark.parametrize("path", [p for p in G.synthetic_paths() if "_1000x" not in p and "_2000x" not in p], ids=G.ident)
def test_oracle_reproduces_reference_synthetic(oracle_lib, path):
    g = G.load(path)
    if g["tableau"]["rows"] is None:
        pytest.skip("dense instance rebuilt by the generator test")
    if not usable(g):
        pytest.skip("out of scope")
    replay(oracle_lib, g)
