"""Round 4: the XCD-LOCAL register-resident geometry (`k_simplex_resident<512, 2, 32, .., XL>`: tableaus up to 1024 x 1024 on the
<= 32 workgroups of ONE XCD, hand-offs through that XCD's L2 -- jslp_resident.hip.h / jslp_resident_pipe.hip.h, `XL`) takes
mid-size LPs without unrestricted variables / optional objectives (BASELINE configs 2 and 4-root: tableau.ts:250-258,
simplex.ts:14-23) when JSLP_XL=1 (opt-in: measured correct but not faster than the chip-wide kernel in round 4).  Against the reference's own goldens (every pivot, flags, final tableau), the C restatement on two-phase
instances, with the health counters asserted: a launch whose workgroups did not land on one XCD is an ABORT, never a silent slow path."""
import os
import sys

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Model, generators
from jslpsolver_amd.engine import Tableau, pivot_digest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from resident_stress import int_instance  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture
def xl_lib(hip_hooks_lib):
    """Round 5: the XCD-local instances live in the TEST library only (libjslp_hip_chaos.so: -DJSLP_CHAOS_BUILD implies -DJSLP_WITH_XL) --
    the default policy never picked the geometry, so the shipped library no longer carries it"""
    return hip_hooks_lib


def _solve(lib, m, vibr, vibc, check, precision=1e-8):
    t = Tableau(m, vibr, vibc, [], precision=precision, lib=lib)
    res = t.simplex(check_cycles=check)
    out = dict(path=t.last_path(), res=res, trace=t.pivot_trace(), final=t.download()[0], cnt=t.get_counters(), evaluation=t.evaluation)
    t.close()
    return out


@pytest.mark.parametrize("check", [False, True])
@pytest.mark.parametrize("kind,n", [("ra", 500), ("lp", 500), ("ra", 1000), ("lp", 1000)])
def test_xcd_local_mid_size_dense_is_the_reference(xl_lib, kind, n, check, monkeypatch):
    """501 x 501 and 1001 x 1001 (all phase 2 / all phase 1, infeasible) with JSLP_XL=1: the XCD-local geometry; every
    pivot and every double of the final tableau are the reference's; one launch, no abort"""
    monkeypatch.setenv("JSLP_XL", "1")
    name = ("generateResourceAllocation" if kind == "ra" else "generateRandomLP") + "_%dx%d_seed12345" % (n, n)
    g = G.load(os.path.join(G.GOLDEN, "synthetic", name + ".json.gz"))
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
    o = _solve(xl_lib, m, vibr, vibc, check)
    assert o["path"] == "resident-xl", o["path"]
    assert (o["cnt"]["resident_launches"], o["cnt"]["resident_aborts"], o["cnt"]["resident_handovers"]) == (1, 0, 0), o["cnt"]
    assert len(o["trace"]) == g["nPivots"] and pivot_digest(o["trace"]) == g["pivotDigest"]
    assert bool(o["res"].feasible) == g["final"]["feasible"] and bool(o["res"].bounded) == g["final"]["bounded"]
    assert G.sha_matrix(o["final"]) == g["final"]["matrixSha"]
    assert o["evaluation"] == G.num(g["final"]["tableauEvaluation"])


@pytest.mark.parametrize("name", ["Monster_Problem", "Monster_II"])
def test_monster_root_lps_xcd_local(xl_lib, name, monkeypatch):
    """JSLP_XL=1 on BASELINE config 2 (Monster LP, 625 x 553, 1 % dense) and config 4's root relaxation (Monster_II, 945 x 925): the first
    simplex() of the reference's own run -- pivots, RHS column + row map, evaluation"""
    monkeypatch.setenv("JSLP_XL", "1")
    g = G.load(os.path.join(G.GOLDEN, "fixtures", name + ".json.gz"))
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    assert G.sha_matrix(m) == g["tableau"]["matrixSha"]
    cap = m.shape[0] + 2 * len(model.integerVariables)
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=cap, lib=xl_lib)
    res = t.simplex(check_cycles=g["tableau"]["checkForCycles"])
    call = g["simplexCalls"][0]
    rhs, rows = t.read_rhs()
    path, cnt, trace = t.last_path(), t.get_counters(), t.pivot_trace()
    t.close()
    assert path == "resident-xl", path
    assert (cnt["resident_launches"], cnt["resident_aborts"]) == (1, 0)
    assert (res.pivots_phase1, res.pivots_phase2) == (call["p1"], call["p2"]) and res.evaluation == call["evaluation"]
    first = np.asarray(g["pivots"], dtype=np.int64).reshape(-1, 2)[:len(trace)]
    assert len(trace) == call["p1"] + call["p2"] and np.array_equal(np.asarray(trace, dtype=np.int64), first)
    assert G.sha_rhs(rhs[:call["height"]], rows[:call["height"]]) == call["rhsSha"]


@pytest.mark.parametrize("shape,kind", [((1000, 1000), "int2p"), ((1000, 1000), "int"), ((1020, 300), "int2p"), ((300, 1020), "int2p"), ((90, 70), "int2p")])
def test_xcd_local_two_phase_against_the_oracle(xl_lib, oracle_lib, shape, kind, monkeypatch):
    """phase 1 then phase 2 inside the one launch (resident_phase1_pipe hands over to resident_phase2_pipe), ragged shapes: rows
    that do not fill the last workgroup, 32 / 10 / 3 rows per workgroup, fewer than 32 workgroups"""
    monkeypatch.setenv("JSLP_FORCE_PATH", "xl")
    m, n = shape
    A, vibr, vibc = int_instance(m, n, 777, kind == "int2p")
    out = []
    for lib in (oracle_lib, xl_lib):
        t = Tableau(A, vibr, vibc, lib=lib)
        res = t.simplex(check_cycles=True)
        out.append((res.pivots_phase1, res.pivots_phase2, bool(res.feasible), bool(res.optimal), pivot_digest(t.pivot_trace()),
                    G.sha_matrix(t.download()[0])))
        if lib is xl_lib:
            assert t.last_path() == "resident-xl" and t.get_counters()["resident_aborts"] == 0
        t.close()
    assert out[0] == out[1]
    if kind == "int2p":
        assert out[0][0] > 0


@pytest.mark.parametrize("abort_at", [0, 9])
def test_xcd_local_abort_rolls_back_and_resolves(hip_hooks_lib, abort_at, monkeypatch):
    hip_lib = hip_hooks_lib  # (the abort hook lives in the test build of the library: tests/conftest.py)
    monkeypatch.setenv("JSLP_XL", "1")
    monkeypatch.setenv("JSLP_TEST_RESIDENT_ABORT", str(abort_at))
    g = G.load(os.path.join(G.GOLDEN, "synthetic", "generateResourceAllocation_500x500_seed12345.json.gz"))
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 500, 500)
    o = _solve(hip_lib, m, vibr, vibc, False)
    assert o["path"] in ("fused", "select+update") and o["cnt"]["resident_aborts"] == 1
    assert pivot_digest(o["trace"]) == g["pivotDigest"] and G.sha_matrix(o["final"]) == g["final"]["matrixSha"]


def test_xcd_local_is_opt_in(xl_lib, monkeypatch):
    monkeypatch.delenv("JSLP_XL", raising=False)
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 500, 500)
    o = _solve(xl_lib, m, vibr, vibc, False)
    assert o["path"] == "resident" and pivot_digest(o["trace"]) == "1cda2607"


def test_shipped_library_has_no_xcd_local_kernels(hip_lib, monkeypatch):
    """the shipped library ignores JSLP_XL (the chip-wide geometry answers, same trace) and refuses JSLP_FORCE_PATH=xl loudly"""
    monkeypatch.setenv("JSLP_XL", "1")
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 500, 500)
    o = _solve(hip_lib, m, vibr, vibc, False)
    assert o["path"] == "resident" and pivot_digest(o["trace"]) == "1cda2607"
    monkeypatch.setenv("JSLP_FORCE_PATH", "xl")
    with pytest.raises(Exception, match="test library only"):
        Tableau(m, vibr, vibc, [], lib=hip_lib)
