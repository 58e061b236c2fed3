#!/usr/bin/env python3
"""Batch-path self-check: the 2416-node Monster_II batch under the current environment knobs -> sha256 of every node's outcome
(flags, pivot counts, RHS column, row map over its own height) + the work counters.  Run under different JSLP_* knobs: the
digests and the per-node counters (pivots, gated rows / cells, cut rows) must not depend on the launch shape."""
import gzip, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jslpsolver_amd import Model, _capi
from jslpsolver_amd.engine import Tableau
lib = _capi.load_hip()
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
model = Model(g["model"])
m, vibr, vibc = model.build_tableau()
base = [c["cuts"] or [] for c in g["simplexCalls"][1:]]
t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
t.applyCuts([], check_cycles=True)
t.save()
nodes = base * int(os.environ.get("REPS", "16"))
packed = t.pack_cut_lists(nodes)
for rep in range(3):
    if rep == 2:
        t.set_counting(True)
    out, rhs, rows = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=True)
    h = hashlib.sha256()
    for i, o in enumerate(out):
        H = m.shape[0] + len(nodes[i])
        h.update(np.array([o.feasible, o.bounded, o.pivots_phase1, o.pivots_phase2], dtype=np.int64).tobytes())
        h.update(np.float64(o.evaluation).tobytes())
        h.update(rhs[i, :H].tobytes())
        h.update(rows[i, :H].tobytes())
    print("rep %d outcome sha %s" % (rep, h.hexdigest()[:16]), flush=True)
print(t.get_counters(), flush=True)
if os.environ.get("DIFF"):
    t2 = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
    t2.applyCuts([], check_cycles=True)
    t2.save()
    o0, r0, w0 = t2.applyCutsBatch(None, check_cycles=True, packed=packed, copy=True)
    o1, r1, w1 = t2.applyCutsBatch(None, check_cycles=True, packed=packed, copy=True)
    nd = 0
    for i in range(len(nodes)):
        H = m.shape[0] + len(nodes[i])
        a = (o0[i].feasible, o0[i].bounded, o0[i].pivots_phase1, o0[i].pivots_phase2, o0[i].evaluation)
        b = (o1[i].feasible, o1[i].bounded, o1[i].pivots_phase1, o1[i].pivots_phase2, o1[i].evaluation)
        dr = np.nonzero(r0[i, :H].view(np.int64) != r1[i, :H].view(np.int64))[0]
        dw = np.nonzero(w0[i, :H] != w1[i, :H])[0]
        if a != b or len(dr) or len(dw):
            nd += 1
            if nd <= 5:
                print("node", i, "cuts", len(nodes[i]), a, b, "rhs diffs", dr[:8], [(r0[i, j], r1[i, j]) for j in dr[:4]], "row diffs", dw[:8], [(w0[i, j], w1[i, j]) for j in dw[:4]])
    print("nodes that differ between the first (five-launch) and the second (one-launch) batch:", nd)
