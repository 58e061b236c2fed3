"""What the REFERENCE (or, for instances it has no run of, the C restatement pinned against it) answered on the instances the
profiling / stress tools solve -- read from the committed golden files under tests/golden (data only: nothing under oracle/ is
imported or executed here).  Every tool that reports a number for a solve checks the solve against this first; a tool without
an expectation for its instance refuses to write a row (JSLP_ALLOW_UNVERIFIED=1 overrides, and the row then says so).
  expected_dense(kind, n_vars, n_rows)   generateResourceAllocation ("ra") / generateRandomLP ("lp"), seed 12345, density 1
  expected_stress(kind, rows, cols, seed) the integer instances of tools/resident_stress.py (tests/golden/stress_expect.json,
                                         written by tests/golden/gen_stress_expect.py)"""
import glob
import gzip
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_GEN = {"ra": "generateResourceAllocation", "lp": "generateRandomLP"}


def _load(path):
    with gzip.open(path, "rt") as fh:
        g = json.load(fh)
    return {"pivots": int(g["nPivots"]), "digest": g["pivotDigest"], "final_sha": g["final"]["matrixSha"],
            "feasible": bool(g["final"]["feasible"]), "bounded": bool(g["final"]["bounded"]),
            "source": os.path.relpath(path, ROOT) + " (the reference's own run)"}


def expected_dense(kind, n_vars, n_rows):
    p = os.path.join(GOLDEN, "synthetic", "%s_%dx%d_seed12345.json.gz" % (_GEN[kind], n_vars, n_rows))
    if os.path.exists(p):
        return _load(p)
    if kind == "ra":  # the shapes beyond 2000 x 2000: tests/golden/gen_golden_wide.js (meta = the generator's arguments)
        for q in sorted(glob.glob(os.path.join(GOLDEN, "wide", "*_RA_*.json.gz"))):
            with gzip.open(q, "rt") as fh:
                g = json.load(fh)
            m = g.get("meta") or {}
            if m.get("kind") == "ra" and m.get("n") == n_vars and m.get("m") == n_rows and not g.get("exitOnCycles"):
                return _load(q)
    return None


def expected_wide(prefix, n_vars, n_rows, k):
    """the reference's own runs of tests/golden/gen_golden_wide.js: prefix "soft" (k soft resources: optional objectives, simplex.ts:221-263,
    394-412) or "unrestricted" (k unrestricted variables), generateResourceAllocation(12345) with n_vars variables and n_rows constraints"""
    p = os.path.join(GOLDEN, "wide", "%s_RA_%dx%d_k%d.json.gz" % (prefix, n_vars, n_rows, k))
    return _load(p) if os.path.exists(p) else None


def expected_stress(kind, rows, cols, seed=12345):
    """kind: "int" / "int2p" (tools/resident_stress.py's instances); rows x cols = the TABLEAU's shape"""
    p = os.path.join(GOLDEN, "stress_expect.json")
    if not os.path.exists(p):
        return None
    with open(p) as fh:
        table = json.load(fh)
    e = table.get("%s_%dx%d_seed%d" % (kind, rows, cols, seed))
    if e:
        e = dict(e)
        e["source"] = "tests/golden/stress_expect.json (C restatement of the reference, itself pinned against the reference's goldens)"
    return e


def solve_signature(tableau, result, digest_fn):
    """(pivots, digest, sha256 of the final matrix) of the solve `tableau` just finished"""
    import numpy as np
    piv = result.pivots_phase1 + max(result.pivots_phase2, 0)
    tr = np.asarray(tableau.pivot_trace()[-piv:] if piv else [], dtype=np.int64).reshape(-1, 2)
    final = tableau.download()[0]
    return {"pivots": int(piv), "digest": digest_fn(tr), "final_sha": hashlib.sha256(np.ascontiguousarray(final).tobytes()).hexdigest()}


def check(sig, want, what):
    """raises SystemExit(3) with both answers when a solve differs from the expectation"""
    bad = [k for k in ("pivots", "digest", "final_sha") if k in want and want[k] is not None and sig[k] != want[k]]
    if bad:
        raise SystemExit("WRONG ANSWER on %s: got %s, want %s (%s)" % (what, {k: sig[k] for k in bad}, {k: want[k] for k in bad}, want.get("source")))


def unverified_allowed():
    return os.environ.get("JSLP_ALLOW_UNVERIFIED", "0") == "1"
