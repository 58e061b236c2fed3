"""Incremental branch-and-bound (options.useIncremental) with device-resident checkpoints, against the reference's
own incremental service (src/tableau/incremental-branch-and-cut.ts) run by tests/golden/gen_golden_incremental.js:
same pivots in the same order, same number of relaxations, same final tableau, same result object.

CPU: the oracle library behind the ABI.  `-m gpu`: libjslp_hip.so.
"""
import copy
import gzip
import json
import os

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Model, Solve, Tableau, UnsupportedModel, pivot_digest

with gzip.open(os.path.join(G.GOLDEN, "incremental.json.gz"), "rt") as fh:
    CASES = json.load(fh)


def case_id(c):
    pol = "-".join("%s" % v for k, v in sorted(c["options"].items()) if k in ("nodeSelection", "branching"))
    return os.path.basename(c["file"])[:-len(".json.gz")] + ("[" + pol + "]" if pol else "")


def interesting(c):
    """the CPU suite replays every case that actually branches plus a sample of the single-relaxation ones"""
    return c["iterations"] > 1 or "depth-first" in json.dumps(c["options"])


def run_case(lib, c):
    g = G.load(os.path.join(G.GOLDEN, c["file"]))
    model = copy.deepcopy(g["model"])
    model["options"] = {k: v for k, v in c["options"].items() if k != "timeout"}  # wall-clock limits are not replayable;
    # no golden run was ended by one (each case also passes with the limit in place when the host is fast enough)
    try:
        Model(model)
    except UnsupportedModel as e:
        pytest.skip(str(e))
    if c["presolveFixed"] > 0:
        pytest.skip("the reference's presolve pre-pass fixed variables: host pre-pass out of scope")
    out = Solve(model, full=True, lib=lib)
    res = out["result"]
    assert len(out["pivots"]) == c["nPivots"]
    assert pivot_digest(out["pivots"]) == c["pivotDigest"]
    assert out["iter"] == c["iterations"]
    assert list(res.keys()) == c["resultKeys"]
    for k, v in c["result"].items():
        ref = v if isinstance(v, bool) else G.num(v)
        assert res[k] == ref or (isinstance(ref, float) and np.isnan(ref) and np.isnan(res[k])), k
    if c["matrixSha"]:
        assert G.sha_matrix(out["matrix"]) == c["matrixSha"]
    return out


@pytest.mark.parametrize("case", [c for c in CASES if interesting(c)], ids=case_id)
def test_incremental_service_through_oracle_engine(oracle_lib, case):
    run_case(oracle_lib, case)


def test_some_case_uses_checkpoints(oracle_lib):
    """the goldens would be vacuous if no node ever started from a checkpoint"""
    used = 0
    for c in CASES:
        if c["iterations"] > 3 and c["presolveFixed"] == 0 and c["options"].get("nodeSelection") == "depth-first":
            try:
                out = run_case(oracle_lib, c)
            except pytest.skip.Exception:
                continue
            used += out["incrementalNodes"]
            if used > 20:
                break
    assert used > 20


def _small_tableau(lib):
    rng = np.random.default_rng(7)
    H, W = 9, 7
    m = np.zeros((H, W))
    m[0, 1:] = rng.integers(1, 9, W - 1)
    m[1:, 1:] = rng.integers(1, 9, (H - 1, W - 1))
    m[1:, 0] = rng.integers(20, 60, H - 1)
    vibr = np.array([-1] + list(range(W - 1, W + H - 2)), dtype=np.int32)
    vibc = np.array([-1] + list(range(W - 1)), dtype=np.int32)
    return Tableau(m, vibr, vibc, precision=1e-8, row_capacity=H + 6, lib=lib)


def check_checkpoint_semantics(lib):
    """createCheckpoint / restoreCheckpoint (incremental-branch-and-cut.ts:55-107) on the engine alone"""
    t = _small_tableau(lib)
    t.simplex()
    root = t.download()
    t.save()
    ck0 = t.createCheckpoint()
    basic = int(root[1][1])
    cut = {"type": "max", "varIndex": basic, "value": float(np.floor(root[0][1, 0] - 0.5))}
    (r1, rhs1, rows1), = t.applyCutsFrom(ck0, [[cut]])
    t.absorb_from(ck0, r1)
    d1 = t.download()
    assert d1[0].shape[0] == root[0].shape[0] + 1
    ck1 = t.createCheckpoint()
    # a second, deeper cut from the child's checkpoint, then back to the FIRST checkpoint: bit-identical to the root
    basic2 = int(d1[1][2])
    cut2 = {"type": "min", "varIndex": basic2, "value": float(np.ceil(d1[0][2, 0] + 0.5))}
    t.applyCutsFrom(ck1, [[cut2]])
    t.restoreCheckpoint(ck0)
    back = t.download()
    for a, b in zip(back, root):
        assert np.array_equal(a, b)
    assert t.height == root[0].shape[0]
    # the same child again, from the root snapshot this time: the default service's path gives the same tableau
    r1b, rhs1b, rows1b = t.applyCuts([cut])
    assert (r1b.pivots_phase1, r1b.pivots_phase2, r1b.feasible) == (r1.pivots_phase1, r1.pivots_phase2, r1.feasible)
    assert np.array_equal(rhs1b, rhs1) and np.array_equal(rows1b, rows1)
    # two children of one checkpoint in one call == the two single calls
    both = t.applyCutsFrom(ck1, [[cut2], [dict(cut2, type="max", value=cut2["value"] - 1)]])
    (a,) = t.applyCutsFrom(ck1, [[cut2]])
    (b,) = t.applyCutsFrom(ck1, [[dict(cut2, type="max", value=cut2["value"] - 1)]])
    for got, ref in zip(both, (a, b)):
        assert (got[0].feasible, got[0].pivots_phase1, got[0].pivots_phase2) == (ref[0].feasible, ref[0].pivots_phase1, ref[0].pivots_phase2)
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    # ids are recycled after release; a released id is refused
    t.releaseCheckpoint(ck0)
    with pytest.raises(Exception):
        t.restoreCheckpoint(ck0)
    ck2 = t.createCheckpoint()
    assert ck2["id"] == ck0["id"]
    t.close()


def test_checkpoint_semantics_oracle(oracle_lib):
    check_checkpoint_semantics(oracle_lib)


@pytest.mark.gpu
def test_checkpoint_semantics_hip(hip_lib):
    check_checkpoint_semantics(hip_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["iterations"] > 1], ids=case_id)
def test_incremental_service_on_gpu(hip_lib, case):
    run_case(hip_lib, case)
