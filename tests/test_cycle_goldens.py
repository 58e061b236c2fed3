"""DETECTED cycles (simplex.ts:78-93, 305-320, 415-440) against the reference itself: tests/golden/cycles/*.json.gz hold runs
of the reference that end in "Cycle in phase 2" -- small degenerate LPs, small LPs with unrestricted variables, and the same
LPs embedded in tableaus large enough for the chip-wide kernels (gen_golden_cycles.js explains the embedding).  Pinned: every
pivot up to the stop, the flags (feasible = false), the phase of the hit and the reference's "Start :" / "Length :" messages
(cycle_start / cycle_length of the C ABI), the final tableau.
CPU: the test-only oracle library; -m gpu: the HIP library through every launch shape the instance fits."""
import glob
import os

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Model
from jslpsolver_amd.engine import Tableau, pivot_digest

CYCLES = os.path.join(G.GOLDEN, "cycles")
SMALL = sorted(p for p in glob.glob(os.path.join(CYCLES, "*.json.gz")) if "embedded" not in p and "late_" not in p)
LATE = os.path.join(CYCLES, "late_deg_35358_after_RA_1500.json.gz")  # gen_golden_late_cycle.js: the hit comes after 6581 pivots
EMBEDDED = sorted(glob.glob(os.path.join(CYCLES, "embedded_*.json.gz")))


def _lcg(seed):
    s = [seed & 0xffffffff]

    def r():
        s[0] = (s[0] * 1664525 + 1013904223) & 0xffffffff  # == (Math.imul(s, 1664525) + 1013904223) >>> 0
        return s[0] / 4294967296.0
    return r


def _embed(model, extra_vars, extra_cons, seed=777):
    """gen_golden_cycles.js `embed`, draw for draw"""
    r = _lcg(seed)
    big = {"optimize": model["optimize"], "opType": model["opType"], "constraints": dict(model["constraints"]),
           "variables": {k: dict(v) for k, v in model["variables"].items()}}
    if "unrestricted" in model:
        big["unrestricted"] = dict(model["unrestricted"])
    for i in range(extra_cons):
        big["constraints"]["fc%d" % i] = {"max": 100 + int(r() * 900)}
    for j in range(extra_vars):
        big["variables"]["f%d" % j] = {"fc%d" % i: 1 + int(r() * 20) for i in range(extra_cons)}
    big["options"] = {"presolve": False}
    return big


def _late_model(small, n, seed):
    """gen_golden_late_cycle.js `lateCycle`: generateResourceAllocation(seed, n x n) as a minimisation, then the small cycling LP"""
    from jslpsolver_amd import generators
    ra = generators.resource_allocation_model(seed, num_variables=n, num_constraints=n, density=1.0)
    big = {"optimize": "obj", "opType": "min", "constraints": dict(ra["constraints"]), "variables": {}}
    for k, v in ra["variables"].items():
        v = dict(v)
        v["obj"] = -v.pop(ra["optimize"])
        big["variables"][k] = v
    for k, c in small["constraints"].items():
        big["constraints"]["z_" + k] = c
    for k, v in small["variables"].items():
        big["variables"]["z_" + k] = {("obj" if a == small["optimize"] else "z_" + a): x for a, x in v.items()}
    big["options"] = {"presolve": False}
    return big


def _instance(path):
    g = G.load(path)
    name = G.ident(path)
    if name.startswith("late_"):
        meta = g["meta"]
        small = G.load(os.path.join(CYCLES, "%s.json.gz" % meta["small"]))["model"]
        m, vibr, vibc = Model(_late_model(small, meta["n"], meta["seed"])).build_tableau()
    elif g["model"] is not None:
        m, vibr, vibc = G.dense_tableau(g["tableau"])
    else:  # embedded_<family>_<seed>_<vars>x<cons>: rebuilt from the small golden's model
        _, fam, seed, dims = name.split("_")
        nv, nc = (int(x) for x in dims.split("x"))
        small = G.load(os.path.join(CYCLES, "%s_%s.json.gz" % (fam, seed)))["model"]
        m, vibr, vibc = Model(_embed(small, nv, nc)).build_tableau()
    assert m.shape == (g["tableau"]["height"], g["tableau"]["width"])
    assert G.sha_matrix(m) == g["tableau"]["matrixSha"], "the rebuilt tableau is not the reference's"
    return g, m, vibr, vibc


def _messages(res):
    return ["Cycle in phase %d" % res.cycle_phase, "Start :%d" % res.cycle_start, "Length :%d" % res.cycle_length] if res.cycle_phase else []


def _check(lib, path, expect_path=None):
    g, m, vibr, vibc = _instance(path)
    assert g["messages"] and g["messages"][0].startswith("Cycle in phase"), "golden without a detected cycle"
    t = Tableau(m, vibr, vibc, g["tableau"]["unrestricted"], precision=g["tableau"]["precision"], lib=lib)
    res = t.simplex(check_cycles=True)
    call = g["simplexCalls"][0]
    assert _messages(res) == g["messages"]                       # the phase, and the reference's [start, length] of the repeated block
    assert bool(res.feasible) == call["feasible"] is False       # simplex.ts:90 / 317
    assert (res.pivots_phase1, res.pivots_phase2) == (call["p1"], call["p2"])
    trace = t.pivot_trace()
    assert len(trace) == g["nPivots"] and pivot_digest(trace) == g["pivotDigest"]
    assert G.sha_matrix(t.download()[0]) == g["final"]["matrixSha"]
    rhs, rows = t.read_rhs()
    assert G.sha_rhs(rhs, rows) == call["rhsSha"]
    if expect_path is not None:
        assert t.last_path() == expect_path
    t.close()


def test_goldens_are_there():
    assert len(SMALL) == 13 and len(EMBEDDED) == 7 and os.path.exists(LATE)


def test_oracle_finds_the_late_cycle(oracle_lib):
    """6602 pivots, the repeated block starts at pivot 6581: the reference's own run (3 minutes under node)"""
    _check(oracle_lib, LATE)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "fused", "resident_general"])
def test_hip_finds_the_late_cycle(hip_lib, mode, monkeypatch):
    """default policy: the lean register-resident kernel, whose LDS holds the first 4096 pairs of the history only -- the pair
    filter says "seen before", the suffix test reads the workgroup's global copy of the history; the streaming kernels; the general
    resident build (LDS history of 10240 pairs)"""
    if mode == "fused":
        monkeypatch.setenv("JSLP_FORCE_PATH", "fused")
    if mode == "resident_general":
        monkeypatch.setenv("JSLP_RES_LEAN", "0")
        monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    _check(hip_lib, LATE, "fused" if mode == "fused" else "resident")


@pytest.mark.parametrize("path", SMALL + EMBEDDED, ids=G.ident)
def test_oracle_stops_where_the_reference_stops(oracle_lib, path):
    _check(oracle_lib, path)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "wg", "sp", "fused", "resident", "xl"])
@pytest.mark.parametrize("path", SMALL, ids=G.ident)
def test_hip_small_every_launch_shape(hip_lib, hip_hooks_lib, path, mode, monkeypatch):
    if mode != "auto":
        monkeypatch.setenv("JSLP_FORCE_PATH", mode)
    _check(hip_hooks_lib if mode == "xl" else hip_lib, path)  # (round 5: the XCD-local kernels live in the test library only)


@pytest.mark.gpu
@pytest.mark.parametrize("path", EMBEDDED, ids=G.ident)
def test_hip_embedded_default_policy_runs_register_resident(hip_lib, path):
    """the default policy on the large embeddings: the lean register-resident kernel (every workgroup runs the suffix test on its
    own LDS history), its tall / wide geometries, and -- with unrestricted variables -- the general build"""
    name = G.ident(path)
    _check(hip_lib, path, None if ("unr" in name and "3000" in name) else "resident")  # (4001-row unrestricted: fused by default)


@pytest.mark.gpu
@pytest.mark.parametrize("path", EMBEDDED, ids=G.ident)
def test_hip_embedded_through_the_streaming_kernels(hip_lib, path, monkeypatch):
    """workgroup 0's check in k_pivot_fused / k_select"""
    monkeypatch.setenv("JSLP_FORCE_PATH", "fused")
    _check(hip_lib, path, "fused")


@pytest.mark.gpu
@pytest.mark.parametrize("path", [p for p in EMBEDDED if "2040x2030" in p or "2000x2000" in p], ids=G.ident)
def test_hip_embedded_general_resident_build(hip_lib, path, monkeypatch):
    """JSLP_RES_LEAN=0, forced resident: the GENERAL build's leaderless protocol with its per-workgroup LDS history on the headline
    geometry (2011 x 2012), with and without unrestricted variables.  Round 4: the tall / wide geometries have no general build any
    more (the lean one takes unrestricted variables; the general one spilled and lost to the streaming kernels): the embeddings wider
    than 2048 columns then run the fused pipeline -- same answer"""
    monkeypatch.setenv("JSLP_RES_LEAN", "0")
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    # (assert the kernel each family runs.  The general build exists for the headline geometry only: <= 2048 rows AND <= 2048
    #  columns -- 2011 x 2012 and the unrestricted 2040 x 2048 embedding; the other two "2040x2030" embeddings are 2036 x 2049 and
    #  2041 x 2052 tableaus, wider than the headline geometry, whose geometries have a lean build only -> the fused pipeline)
    g = G.load(path)
    fits_headline = g["tableau"]["height"] <= 2048 and g["tableau"]["width"] <= 2048
    _check(hip_lib, path, "resident" if fits_headline else "fused")
