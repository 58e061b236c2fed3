"""Wall time / pivot rate of the two config-3 instances (3a: all phase 2, 3b: all phase 1) on the current build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau, pivot_digest
lib = _capi.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for kind in ("3a resource allocation", "3b random LP"):
    if kind.startswith("3a"):
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
    for check in (False, True):
        t = Tableau(m, vibr, vibc, lib=lib)
        t.save()
        best = 1e9
        for _ in range(3):
            t.restore()
            t0 = time.perf_counter(); res = t.simplex(check_cycles=check); best = min(best, time.perf_counter() - t0)
        piv = res.pivots_phase1 + max(res.pivots_phase2, 0)
        print("%-24s n=%d cycle-check=%-5s path=%-14s pivots=%5d  %.1f ms  %.2f us/pivot  %.0f pivots/s  feasible=%d digest=%s" % (
            kind, n, check, t.last_path(), piv, best * 1e3, best * 1e6 / max(piv, 1), piv / best, res.feasible,
            pivot_digest(t.pivot_trace()[-piv:])), flush=True)
        t.close()
