"""Repeat one dense LP through the register-resident kernels and compare EVERY run's pivot count, pivot digest and final tableau
with the known answer of the instance (tools/known_answers.py: the reference's own run, or -- for the integer instances below -- the
C restatement pinned against it; tests/golden/stress_expect.json), not merely with the first run.  Exits 1 on any wrong run, 2 when the
instance has no known answer (JSLP_ALLOW_UNVERIFIED=1: compare with the first run instead, and say so).
  python tools/resident_stress.py rows cols runs [fresh] [--kind int|ra|lp|int2p] [--unr k] [--check]
    int    (default) dense integer LP, all "<=" rows (phase 2 only): rows x cols constraints x variables, SEED (env) = 12345
    int2p  the same with rows/8 ">=" rows: a phase 1 first (through the fused pipeline on the tall / wide geometries)
    ra/lp  the reference's generateResourceAllocation / generateRandomLP(seed 12345) with cols variables and rows constraints
    soft   generateResourceAllocation(12345) with --k soft resources (three optional objective rows: the OPT builds; the known answer is the
           reference's own run, tests/golden/wide/soft_RA_<cols>x<rows>_k<k>: 2000 x 3000 k 30 and 400 x 400 k 30 exist)
    unr    the same with --k unrestricted variables (the UNR builds; tests/golden/wide/unrestricted_RA_*: 2000 x 3950 k 50, 1000 x 950 k 50, 300 x 250 k 20)
    --unr k   the first k variables declared unrestricted (the GENERAL build; its own known answer: they price differently)
    --check   the reference's default cycle check on
  `fresh` = a new engine per run (upload + first launch each time) instead of restore() on one engine.
Also fails when the engine reports a rolled-back resident launch (jslp_work_counters.resident_aborts): a diverged replica ends in a
grid time-out and the streaming kernels then return the RIGHT answer slowly -- which would hide the very bug this tool is after."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import known_answers as KA  # noqa: E402
from jslpsolver_amd import _capi, generators  # noqa: E402
from jslpsolver_amd.engine import Tableau, pivot_digest  # noqa: E402


def int_instance(m, n, seed, two_phase=False):
    """the instances of tests/golden/gen_stress_expect.py (keep the two in step: the expectations are keyed by shape and seed)"""
    rng = np.random.default_rng(seed)
    A = np.zeros((m + 1, n + 1))
    A[1:, 1:] = rng.integers(1, 21, (m, n))
    A[0, 1:] = rng.integers(1, 51, n)
    A[1:, 0] = rng.integers(100, 501, m)
    if two_phase:  # x_j >= b rows (negated: negative RHS), one variable each, which the "<=" rows leave feasible
        k = max(1, min(m // 8, 100))  # (100 variables at <= 20 each x coefficients <= 20 stay below every "<=" row's limit)
        A[1:, 0] = rng.integers(60000, 100001, m)
        ge = rng.choice(np.arange(1, m + 1), k, replace=False)
        A[ge, 0] = -rng.integers(5, 21, k)
        A[ge, 1:] = 0.0
        A[ge, 1 + rng.choice(n, k, replace=False)] = -1.0
    vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
    vibc = np.array([-1] + list(range(n)), dtype=np.int32)
    return A, vibr, vibc


def main(argv):
    skip = {i + 1 for i, a in enumerate(argv) if a in ("--kind", "--unr", "--k")}
    pos = [a for i, a in enumerate(argv) if not a.startswith("--") and i not in skip]
    m, n, runs = int(pos[0]), int(pos[1]), int(pos[2])
    fresh = len(pos) > 3 and pos[3] == "fresh"
    kind = argv[argv.index("--kind") + 1] if "--kind" in argv else "int"
    n_unr = int(argv[argv.index("--unr") + 1]) if "--unr" in argv else 0
    check = "--check" in argv or os.environ.get("CHECK_CYCLES", "0") == "1"
    k_extra = int(argv[argv.index("--k") + 1]) if "--k" in argv else 30
    oo, unr_list = None, None
    seed = int(os.environ.get("SEED", "12345"))
    if kind in ("int", "int2p"):
        A, vibr, vibc = int_instance(m, n, seed, kind == "int2p")
        # (unrestricted variables price differently -- simplex.ts:164-177 -- so the instance with k of them has its own known answer)
        want = KA.expected_stress(kind + ("unr%d" % n_unr if n_unr else ""), m + 1, n + 1, seed)
    elif kind == "ra":
        A, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, m)
        want = KA.expected_dense("ra", n, m)
    elif kind == "soft":
        A, vibr, vibc, oo = generators.soft_resource_allocation_tableau(12345, n, m, k_extra)
        want = KA.expected_wide("soft", n, m, k_extra)
    elif kind == "unr":
        A, vibr, vibc, unr_list = generators.unrestricted_resource_allocation_tableau(12345, n, m, k_extra)
        want = KA.expected_wide("unrestricted", n, m, k_extra)
    else:
        A, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, m)
        want = KA.expected_dense("lp", n, m)
    if want is None and not KA.unverified_allowed():
        print("no known answer for %s %d x %d (seed %d): add it to tests/golden/gen_stress_expect.py; refusing to stress an unverified instance" % (kind, A.shape[0], A.shape[1], seed))
        return 2
    lib = _capi.load_hip()
    unr = unr_list if unr_list is not None else list(range(n_unr))
    kw = {"optional_objectives": oo} if oo is not None else {}
    first, bad, t, pivots_done, aborts, since_upload, retries = None, 0, None, 0, 0, 0, 0
    for i in range(runs):
        # (an engine's pivot trace holds 2^20 pivots since its upload: a fresh engine before it would overflow)
        if fresh or t is None or (first is not None and since_upload + first["pivots"] > 1000000):
            since_upload = 0
            if t is not None:
                aborts += t.get_counters()["resident_aborts"]
                retries += t.get_counters()["resident_fetch_retries"]
                t.close()
            t = Tableau(A, vibr, vibc, unr, lib=lib, **kw)
            t.save()
        else:
            t.restore()
        r = t.simplex(check_cycles=check)
        sig = KA.solve_signature(t, r, pivot_digest)
        sig["path"] = t.last_path()
        pivots_done += sig["pivots"]
        since_upload += sig["pivots"]
        if first is None:
            first = sig
            print("run 0:", sig["pivots"], sig["digest"], sig["final_sha"][:16], sig["path"], flush=True)
        ref = want if want is not None else first
        wrong = [k for k in ("pivots", "digest", "final_sha") if ref.get(k) is not None and sig[k] != ref[k]]
        if wrong or sig["path"] != first["path"]:
            bad += 1
            print("run %d DIFFERS from %s: got %s want %s (path %s)" % (
                i, "the known answer" if want is not None else "run 0", {k: sig[k] for k in wrong}, {k: ref[k] for k in wrong}, sig["path"]), flush=True)
    aborts += t.get_counters()["resident_aborts"]
    retries += t.get_counters()["resident_fetch_retries"]
    t.close()
    print("%d x %d %s%s%s, %d runs (%s), %d pivots, path %s: %d differ from %s; resident aborts %d; repeated row looks %d" % (
        A.shape[0], A.shape[1], kind, " unr=%d" % n_unr if n_unr else "", " check" if check else "", runs, "fresh engines" if fresh else "one engine",
        pivots_done, first["path"], bad, ("the known answer [%s]" % want["source"]) if want is not None else "the first (UNVERIFIED)", aborts, retries))
    return 1 if (bad or aborts) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
