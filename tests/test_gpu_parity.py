"""GPU parity tests proper: the HIP engine, called through the C ABI, against
  (a) the golden vectors produced by the reference itself (every pivot, every relaxation, final tableau sha),
  (b) the CPU oracle on seeded random inputs (bit-exact final tableau),
for BOTH launch shapes (one workgroup per tableau / select + chip-wide update).
"""
import os

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import generators
from jslpsolver_amd.engine import Tableau, pivot_digest
from test_host_solve import check_fixture
from test_oracle_golden import replay, usable

pytestmark = pytest.mark.gpu

# resident4: 512 lanes x 4 columns (JSLP_RES_CPT=4); wggen: the generic one-workgroup kernels (selection state in global
# memory, JSLP_NO_WGLDS=1) -- "wg" and "auto" use their LDS-resident twins wherever the tableau's vectors fit
# xl: the XCD-local register-resident geometry (round 4; <= 1024 x 1024 without unrestricted variables / optional objectives --
# what does not fit falls through to the chip-wide resident geometries)
PATHS = ["auto", "wg", "wggen", "sp", "fused", "resident", "resident4", "xl"]


def set_path(mode):
    os.environ.pop("JSLP_FORCE_PATH", None)
    os.environ.pop("JSLP_RES_CPT", None)
    os.environ.pop("JSLP_NO_WGLDS", None)
    if mode == "wggen":
        os.environ["JSLP_FORCE_PATH"] = "wg"
        os.environ["JSLP_NO_WGLDS"] = "1"
    elif mode == "resident4":
        os.environ["JSLP_FORCE_PATH"] = "resident"
        os.environ["JSLP_RES_CPT"] = "4"
    elif mode != "auto":
        os.environ["JSLP_FORCE_PATH"] = mode


@pytest.fixture(params=PATHS)
def path_mode(request):
    set_path(request.param)
    yield request.param
    set_path("auto")


def test_backend_is_hip(hip_lib):
    assert hip_lib.backend == "hip-gfx950"
    assert hip_lib.jslp_device_count() >= 1


SMALL = [p for p in G.fixture_paths() if "LargeFarm" not in p and "Vendor" not in p and "Monster" not in p
         and "StockCutting" not in p]


@pytest.mark.parametrize("path", SMALL, ids=G.ident)
def test_fixture_replay(hip_lib, hip_hooks_lib, path_mode, path):
    g = G.load(path)
    if not usable(g):
        pytest.skip("outside the hot-path scope")
    replay(hip_hooks_lib if path_mode == "xl" else hip_lib, g)  # (round 5: the XCD-local kernels live in the test library only)


@pytest.mark.parametrize("name", ["Monster_Problem", "Monster_II", "Vendor_Selection", "StockCuttingProblem", "LargeFarmMIP"])
def test_big_fixture_replay(hip_lib, name):
    replay(hip_lib, G.load(os.path.join(G.GOLDEN, "fixtures", name + ".json.gz")))


@pytest.mark.parametrize("name", ["Monster_Problem", "Monster_II"])
def test_big_fixture_replay_other_paths(hip_lib, hip_hooks_lib, name):
    g = G.load(os.path.join(G.GOLDEN, "fixtures", name + ".json.gz"))
    for mode in ("wg", "wggen", "sp", "fused", "resident", "resident4", "xl"):
        set_path(mode)
        try:
            replay(hip_hooks_lib if mode == "xl" else hip_lib, g)
        finally:
            set_path("auto")


@pytest.mark.parametrize("path", [p for p in G.synthetic_paths() if "_1000x" not in p and "_2000x" not in p], ids=G.ident)
def test_synthetic_replay(hip_lib, hip_hooks_lib, path_mode, path):
    if path_mode == "xl":
        hip_lib = hip_hooks_lib  # (round 5: the XCD-local kernels live in the test library only)
    g = G.load(path)
    if g["tableau"]["rows"] is None or not usable(g):
        pytest.skip("dense instance: see test_dense_synthetic")
    replay(hip_lib, g)


@pytest.mark.parametrize("path", G.fixture_paths(), ids=G.ident)
def test_fixture_through_host_and_hip_engine(hip_lib, path):
    """solver.Solve(json) end to end: model layer + B&B on the host, every relaxation on the GPU"""
    check_fixture(hip_lib, G.load(path))


def _dense_case(lib, kind, n, check_cycles):
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
    t = Tableau(m, vibr, vibc, lib=lib)
    res = t.simplex(check_cycles=check_cycles)
    out = dict(path=t.last_path(), feasible=bool(res.feasible), bounded=bool(res.bounded), p1=res.pivots_phase1, p2=res.pivots_phase2,
               evaluation=t.evaluation, digest=pivot_digest(t.pivot_trace()), n=len(t.pivot_trace()),
               matrix=t.download()[0])
    t.close()
    return out


@pytest.mark.parametrize("kind,n", [("ra", 200), ("lp", 200), ("ra", 500), ("lp", 500), ("ra", 1000), ("lp", 1000)])
def test_dense_synthetic_against_reference_golden(hip_lib, hip_hooks_lib, path_mode, kind, n):
    if path_mode == "xl":
        hip_lib = hip_hooks_lib  # (round 5: the XCD-local kernels live in the test library only)
    if path_mode in ("wg", "wggen") and n > 500:
        pytest.skip("one workgroup on a 1000x1000 dense tableau: correct but slow")
    name = ("generateResourceAllocation" if kind == "ra" else "generateRandomLP") + "_%dx%d_seed12345" % (n, n)
    g = G.load(os.path.join(G.GOLDEN, "synthetic", name + ".json.gz"))
    out = _dense_case(hip_lib, kind, n, check_cycles=g["tableau"]["checkForCycles"])
    assert out["n"] == g["nPivots"]
    assert out["digest"] == g["pivotDigest"]
    assert out["feasible"] == g["final"]["feasible"] and out["bounded"] == g["final"]["bounded"]
    assert G.sha_matrix(out["matrix"]) == g["final"]["matrixSha"]
    assert out["evaluation"] == G.num(g["final"]["tableauEvaluation"])


def test_config3_full_size_against_reference_golden(hip_lib):
    """BASELINE.json config 3 at full size (2001 x 2001, 9726 pivots): identical pivot sequence and an
    identical final tableau (sha256 of all 4M doubles) to the reference."""
    for kind, name in (("ra", "generateResourceAllocation"), ("lp", "generateRandomLP")):
        g = G.load(os.path.join(G.GOLDEN, "synthetic", name + "_2000x2000_seed12345.json.gz"))
        out = _dense_case(hip_lib, kind, 2000, check_cycles=False)
        # 3a is all phase 2, 3b never leaves phase 1: both run as ONE launch of the register-resident kernel
        assert out["path"] == "resident"
        assert out["n"] == g["nPivots"]
        assert out["digest"] == g["pivotDigest"]
        assert G.sha_matrix(out["matrix"]) == g["final"]["matrixSha"]
        assert out["feasible"] == g["final"]["feasible"]


@pytest.mark.parametrize("seed", range(12))
def test_random_lp_hip_equals_oracle(hip_lib, hip_hooks_lib, oracle_lib, path_mode, seed):
    if path_mode == "xl":
        hip_lib = hip_hooks_lib  # (round 5: the XCD-local kernels live in the test library only)
    """seeded random LPs with ragged shapes, unrestricted variables, negative RHS (phase 1) and degenerate
    rows: flags, pivot sequence and every double of the final tableau must match the oracle"""
    rng = np.random.default_rng(1000 + seed)
    n, m = int(rng.integers(1, 60)), int(rng.integers(1, 45))
    A = np.zeros((m + 1, n + 1))
    dens = rng.uniform(0.2, 1.0)
    A[1:, 1:] = np.where(rng.random((m, n)) < dens, rng.integers(-9, 10, (m, n)), 0)
    A[0, 1:] = rng.integers(-5, 20, n)
    A[1:, 0] = rng.integers(-3 if seed % 2 else 0, 30, m)
    if seed % 3 == 0:
        A[1 + (seed % m), 0] = 0.0  # degenerate row
    vibr = np.concatenate(([-1], np.arange(m))).astype(np.int32)
    vibc = np.concatenate(([-1], m + np.arange(n))).astype(np.int32)
    unr = [int(m + j) for j in range(n) if rng.random() < 0.15]
    outs = []
    for lib in (hip_lib, oracle_lib):
        t = Tableau(A, vibr, vibc, unr, lib=lib)
        res = t.simplex(check_cycles=True)
        outs.append((res.as_dict(), t.pivot_trace().tolist(), t.download()[0].tobytes(), t.evaluation))
        t.close()
    d0, d1 = outs[0][0], outs[1][0]
    for k in d0:
        assert d0[k] == d1[k] or (isinstance(d0[k], float) and np.isnan(d0[k]) and np.isnan(d1[k])), k
    assert outs[0][1] == outs[1][1]
    assert outs[0][2] == outs[1][2]


def test_relax_batch_equals_sequential_relax(hip_lib, oracle_lib):
    """Monster_II (config 4): the 151 cut lists the reference visits, evaluated as ONE batch of independent
    nodes, give the reference's per-node outcomes (flags, evaluation, RHS column sha)."""
    g = G.load(os.path.join(G.GOLDEN, "fixtures", "Monster_II.json.gz"))
    tab = g["tableau"]
    m, vibr, vibc = G.dense_tableau(tab)
    calls = g["simplexCalls"]
    max_cuts = max(len(c["cuts"] or []) for c in calls)
    t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"], row_capacity=tab["height"] + max_cuts,
                lib=hip_lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    nodes = [c["cuts"] or [] for c in calls[1:]]
    results, rhs, rows = t.applyCutsBatch(nodes * 3, check_cycles=True)
    # the zero-copy variant (views of the pinned read-back buffer), 5 groups of up to 512 nodes in one call
    packed = t.pack_cut_lists(nodes * 15)
    res_p, rhs_p, rows_p = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    for j in range(len(nodes) * 15):
        call = calls[1 + j % len(nodes)]
        assert bool(res_p[j].feasible) == call["feasible"] and res_p[j].height == call["height"]
        assert G.sha_rhs(rhs_p[j, :res_p[j].height], rows_p[j, :res_p[j].height]) == call["rhsSha"]
    for rep in range(3):
        for i, call in enumerate(calls[1:]):
            j = rep * len(nodes) + i
            r = results[j]
            assert bool(r.feasible) == call["feasible"] and r.height == call["height"]
            assert r.pivots_phase1 == call["p1"] and r.pivots_phase2 == call["p2"]
            assert G.sha_rhs(rhs[j, :r.height], rows[j, :r.height]) == call["rhsSha"]
            if call["feasible"]:
                assert r.evaluation == G.num(call["evaluation"])
    t.close()


@pytest.mark.parametrize("threads,group", [(256, 64), (512, 200), (1024, 1024), (256, 2048)])
def test_batch_launch_shapes(hip_lib, threads, group):
    """every workgroup shape of the per-node kernel and several group sizes give the reference's per-node outcomes"""
    import subprocess
    import sys
    env = dict(os.environ, JSLP_WG_BATCH_THREADS=str(threads), JSLP_GROUP_MAX=str(group))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch_shape_worker.py")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-1500:] + out.stderr[-1500:]


def test_pivot_entry_point(hip_lib, oracle_lib):
    rng = np.random.default_rng(5)
    A = rng.integers(-5, 9, (9, 12)).astype(np.float64)
    vibr = np.concatenate(([-1], np.arange(8))).astype(np.int32)
    vibc = np.concatenate(([-1], 8 + np.arange(11))).astype(np.int32)
    outs = []
    for lib in (hip_lib, oracle_lib):
        t = Tableau(A, vibr, vibc, lib=lib)
        for r, c in ((3, 4), (1, 1), (8, 11), (3, 7)):
            t.pivot(r, c)
        outs.append([x.tobytes() for x in t.download()])
        t.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("name", ["Monster_II", "Knapsack_1", "Sudoku4x4", "Integer_Wood_Shop_Problem", "StockCuttingProblem"])
def test_speculative_batched_bnb_on_gpu(hip_lib, name):
    """host tree with 16-node speculative batches (one k_simplex_wg launch per batch): the reference's result,
    iteration count and final tableau"""
    from jslpsolver_amd import Solve
    g = G.load(os.path.join(G.GOLDEN, "fixtures", name + ".json.gz"))
    out = Solve(g["model"], full=True, lib=hip_lib, speculate=16)
    ref = {k: (G.num(v) if not isinstance(v, bool) else v) for k, v in g["result"].items()}
    assert out["result"] == ref and list(out["result"]) == g["resultKeys"]
    assert out["iter"] == g["final"]["branchAndCutIterations"]
    assert G.sha_matrix(out["matrix"]) == g["final"]["matrixSha"]
