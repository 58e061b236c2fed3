#!/usr/bin/env python3
"""Latency of the one-workgroup LDS kernels (k_simplex_lds<1024> / k_node_lds<1024>) -- what a mid-size sparse LP and every dependent
batch of a speculative branch-and-bound tree wait for:
  * Monster LP (625 x 553, 1 % dense, 60 pivots) and Monster_II's root relaxation (935 x 925): wall time of simplex(), us per pivot;
  * Monster_II: one call of jslp_engine_relax_batch_watched_pinned for the first 1 / 8 / 16 nodes of the reference's own tree
    (upload of the cut lists, one launch, compact read-back, one synchronisation): median / min of 40 calls, us per node.
Every outcome is checked against the reference's golden (pivot digest; rhsSha of each relaxation through the full read-back) before a
number is printed.   python tools/node_latency.py [out.md]      (JSLP_HIP_LIBRARY=... to time another build)"""
import gzip
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Model, _capi  # noqa: E402
from jslpsolver_amd.engine import Tableau, pivot_digest  # noqa: E402


def load(name):
    with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz"), "rt") as fh:
        return json.load(fh)


def main(out_path=None):
    lib = _capi.load_hip()
    lines = ["library: %s" % os.environ.get("JSLP_HIP_LIBRARY", "jslpsolver_amd/csrc/libjslp_hip.so"), ""]
    # ---- whole simplex() of the two Monster tableaus in one workgroup ------------------------------------------------------------
    lines += ["| LP | tableau | pivots | path | simplex() ms (best of 7) | us per pivot | pivot digest == the reference's |", "|---|---|---|---|---|---|---|"]
    for name in ("Monster_Problem", "Monster_II"):
        g = load(name)
        model = Model(g["model"])
        m, vibr, vibc = model.build_tableau()
        cap = m.shape[0] + 2 * len(model.integerVariables)
        t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=cap, lib=lib)
        t.save()
        best, res = 1e9, None
        for _ in range(7):
            t.restore()
            t0 = time.perf_counter()
            res = t.simplex(check_cycles=True)
            best = min(best, time.perf_counter() - t0)
        piv = res.pivots_phase1 + max(res.pivots_phase2, 0)
        call = g["simplexCalls"][0]
        want = pivot_digest(np.asarray(g["pivots"][:2 * piv], dtype=np.int64).reshape(-1, 2)) if g.get("pivots") else None  # (flat list of row, col)
        got = pivot_digest(t.pivot_trace()[-piv:])
        ok = (res.pivots_phase1, max(res.pivots_phase2, 0)) == (call["p1"], max(call["p2"], 0)) and (want is None or want == got)
        rhs, rows = t.read_rhs()
        ok = ok and hashlib.sha256(np.ascontiguousarray(rhs).tobytes() + np.ascontiguousarray(rows).tobytes()).hexdigest() == call["rhsSha"]
        if not ok:
            raise SystemExit("WRONG ANSWER on %s" % name)
        lines.append("| %s%s | %dx%d | %d | %s | %.3f | %.2f | yes (%s; rhsSha of the solve) |" % (
            name, " (root relaxation)" if name == "Monster_II" else "", m.shape[0], m.shape[1], piv, t.last_path(), 1e3 * best, 1e6 * best / max(piv, 1), got))
        t.close()
    # ---- small batches of Monster_II nodes -----------------------------------------------------------------------------------
    g = load("Monster_II")
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    cap = m.shape[0] + 2 * len(model.integerVariables)
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=cap, lib=lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    calls = g["simplexCalls"][1:]
    nodes = [c["cuts"] or [] for c in calls]
    ints = [int(v) for v in model.integer_index_array]
    t.set_watched_variables(ints)
    # the outcomes these batches must reproduce: the full read-back of the same nodes, each against the reference's rhsSha
    results, rhs, rows = t.applyCutsBatch(nodes[:32], check_cycles=True)
    for i in range(32):
        h = results[i].height
        sha = hashlib.sha256(np.ascontiguousarray(rhs[i, :h]).tobytes() + np.ascontiguousarray(rows[i, :h]).tobytes()).hexdigest()
        if h != calls[i]["height"] or sha != calls[i]["rhsSha"]:
            raise SystemExit("WRONG ANSWER: node %d differs from the reference's relaxation outcome" % i)
    lines += ["", "| Monster_II batch | pivots in the batch | us per call, median of 40 | min | us per node | outcomes |", "|---|---|---|---|---|---|"]
    for n_small in (1, 8, 16):
        sub = nodes[3:3 + n_small]
        packed = t.pack_cut_lists(sub)
        fn = lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False)
        for _ in range(8):
            fn()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            r_s, rows_s, vals_s = fn()
            ts.append(time.perf_counter() - t0)
        for i in range(n_small):  # compact outcome == what the verified full read-back says about the integer variables
            h = results[3 + i].height
            row_of = {int(v): r for r, v in enumerate(rows[3 + i, :h]) if r > 0}
            want_rows = np.array([row_of.get(v, -1) for v in ints], dtype=np.int32)
            want_vals = np.array([rhs[3 + i, r] if r > 0 else 0.0 for r in want_rows])
            if r_s[i].height != h or not np.array_equal(np.array(rows_s[i]), want_rows) or np.array(vals_s[i]).tobytes() != want_vals.tobytes():
                raise SystemExit("WRONG ANSWER: small batch of %d, node %d" % (n_small, i))
        ts.sort()
        piv = sum(r_s[i].pivots_phase1 + max(r_s[i].pivots_phase2, 0) for i in range(n_small))
        lines.append("| %d node(s) (nodes 3..%d of the reference's tree) | %d | %.1f | %.1f | %.1f | == the verified full read-back |" % (
            n_small, 3 + n_small - 1, piv, 1e6 * ts[len(ts) // 2], 1e6 * ts[0], 1e6 * ts[len(ts) // 2] / n_small))
    t.close()
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:2])
