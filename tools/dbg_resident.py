import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import golden_util as G
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau, pivot_digest
lib = _capi.load_hip()
for n in (1000, 500, 2000):
    g = G.load(os.path.join(G.GOLDEN, "synthetic", "generateResourceAllocation_%dx%d_seed12345.json.gz" % (n, n)))
    ref = np.array(g["pivots"], dtype=np.int64).reshape(-1, 2)
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    for mode in ("resident", "fused"):
        os.environ["JSLP_FORCE_PATH"] = mode
        for trial in range(3):
            t = Tableau(m, vibr, vibc, lib=lib)
            t0 = time.perf_counter(); res = t.simplex(check_cycles=False); dt = time.perf_counter() - t0
            tr = t.pivot_trace()
            k = min(len(tr), len(ref))
            diff = np.nonzero((tr[:k] != ref[:k]).any(axis=1))[0]
            first = int(diff[0]) if len(diff) else -1
            print(n, mode, trial, "pivots", len(tr), "ref", len(ref), "first_diff", first,
                  (tr[first].tolist(), ref[first].tolist()) if first >= 0 else "", "us/pivot %.2f" % (1e6 * dt / max(len(tr), 1)), flush=True)
            t.close()
