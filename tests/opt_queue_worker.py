"""Subprocess body of test_node_batches_with_optional_objectives_through_the_queue_kernel (JSLP_GROUP_MAX is read once per process):
branch-and-bound children of models with optional objectives as batches LARGER than the slots they get -- the queue kernel's OPT build
(`k_node_queue<512, false, true>`, round 4) -- against the oracle engine evaluating the same cut lists one at a time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_util as G  # noqa: E402
from jslpsolver_amd import _capi  # noqa: E402
from jslpsolver_amd.engine import Tableau  # noqa: E402
from test_wide_goldens import _optional_objective_instance  # noqa: E402

assert os.environ.get("JSLP_GROUP_MAX") == "8"
hip = _capi.load_hip()
oracle = _capi.Library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libjslp_oracle.so"))
for n_rows, n_cols, n_opt in ((20, 30, 1), (30, 40, 2), (60, 90, 3)):
    m, vibr, vibc, oo = _optional_objective_instance(31 + n_rows, n_rows, n_cols, n_opt)
    rng = np.random.default_rng(5)
    ts = []
    for lib in (oracle, hip):
        t = Tableau(m, vibr, vibc, [], lib=lib, optional_objectives=oo, row_capacity=n_rows + 1 + 4)
        assert t.simplex(check_cycles=True).feasible
        t.save()
        ts.append(t)
    nodes = []
    for k in range(40):
        cuts = []
        for _ in range(int(rng.integers(1, 4))):
            cuts.append({"type": "max" if rng.random() < 0.5 else "min", "varIndex": int(rng.integers(0, n_cols)), "value": float(rng.integers(0, 12))})
        nodes.append(cuts)
    ref = []
    for cuts in nodes:
        ts[0].restore()
        r, rhs, rows = ts[0].applyCuts(cuts, check_cycles=True)
        ref.append((bool(r.feasible), r.height, r.pivots_phase1, r.pivots_phase2, r.evaluation if r.feasible else None, G.sha_rhs(rhs[:r.height], rows[:r.height])))
    for call in range(3):  # (the first batch of an engine brings the slots in sync through the per-group launches; the queue takes the next ones)
        results, rhs, rows = ts[1].applyCutsBatch(nodes, check_cycles=True)
        got = [(bool(r.feasible), r.height, r.pivots_phase1, r.pivots_phase2, r.evaluation if r.feasible else None,
                G.sha_rhs(rhs[j, :r.height], rows[j, :r.height])) for j, r in enumerate(results)]
        assert got == ref, (n_rows, n_cols, n_opt, call)
    assert ts[1].get_counters()["node_queue_launches"] >= 2, ts[1].get_counters()
    for t in ts:
        t.close()
print("ok")
