"""Subprocess body of test_batch_launch_shapes (the launch-shape knobs are read once per process): Monster_II's 151
reference nodes x 3 as one batch on the HIP engine, checked against the reference's per-node outcomes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_util as G  # noqa: E402
from jslpsolver_amd import Tableau, _capi  # noqa: E402

g = G.load(os.path.join(G.GOLDEN, "fixtures", "Monster_II.json.gz"))
tab = g["tableau"]
m, vibr, vibc = G.dense_tableau(tab)
calls = g["simplexCalls"]
t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"],
            row_capacity=tab["height"] + max(len(c["cuts"] or []) for c in calls), lib=_capi.load_hip())
t.applyCuts([], check_cycles=True)
t.save()
nodes = [c["cuts"] or [] for c in calls[1:]]
for _round in range(2):  # the second call restores incrementally (dirty rows + RHS mirror)
    results, rhs, rows = t.applyCutsBatch(nodes * 3, check_cycles=True)
    for j, r in enumerate(results):
        call = calls[1 + j % len(nodes)]
        assert bool(r.feasible) == call["feasible"] and r.height == call["height"], j
        assert r.pivots_phase1 == call["p1"] and r.pivots_phase2 == call["p2"], j
        assert G.sha_rhs(rhs[j, :r.height], rows[j, :r.height]) == call["rhsSha"], j
assert t.last_path() == "workgroup"
t.close()
print("ok")
