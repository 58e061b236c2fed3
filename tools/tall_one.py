"""One dense LP (rows x cols [randomlp]) through whatever path the environment selects: pivots/s and the pivot digest.
  tools/tall_one.py rows cols            random dense LP with integer data (phase 2 only)
  tools/tall_one.py rows cols randomlp   generateRandomLP(seed 12345, cols variables, rows constraints): phase-1 pivots"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, pivot_digest
m, n = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "randomlp":  # generateRandomLP (config 3b family): >= constraints, phase-1 pivots
    from jslpsolver_amd import generators
    A, vibr, vibc, _op = generators.dense_random_lp_tableau(12345, n, m)
else:
    rng = np.random.default_rng(12345)
    A = np.zeros((m + 1, n + 1))
    A[1:, 1:] = rng.integers(1, 21, (m, n))
    A[0, 1:] = rng.integers(1, 51, n)
    A[1:, 0] = rng.integers(100, 501, m)
    vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
    vibc = np.array([-1] + list(range(n)), dtype=np.int32)
t = Tableau(A, vibr, vibc, lib=_capi.load_hip())
t.save()
CHECK = os.environ.get("CHECK_CYCLES", "0") == "1"
t.simplex(check_cycles=CHECK)
best = 1e9
for _ in range(int(os.environ.get("REPEATS", "2"))):
    t.restore()
    t0 = time.perf_counter(); r = t.simplex(check_cycles=CHECK); best = min(best, time.perf_counter() - t0)
piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
print(json.dumps({"path": t.last_path(), "p1": r.pivots_phase1, "feasible": bool(r.feasible), "pivots": piv, "pivots_per_s": round(piv / best), "digest": pivot_digest(t.pivot_trace()[-piv:])}))
