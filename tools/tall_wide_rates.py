"""Pivot rate of the dense generateResourceAllocation(12345) LPs on every default register-resident geometry, each solve verified against
the reference's golden first (tools/known_answers.py):  python tools/tall_wide_rates.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import known_answers as KA
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau, pivot_digest
lib = _capi.load_hip()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for n_vars, n_rows in ((2000, 2000), (2000, 4000), (3000, 3000), (4000, 2000)):
    want = KA.expected_dense("ra", n_vars, n_rows)
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n_vars, n_rows)
    for check in (False, True):
        t = Tableau(m, vibr, vibc, lib=lib); t.save()
        best = 1e9
        for i in range(reps):
            t.restore(); t0 = time.perf_counter(); r = t.simplex(check_cycles=check); best = min(best, time.perf_counter() - t0)
            if i == 0: KA.check(KA.solve_signature(t, r, pivot_digest), want, "%d x %d" % (n_rows + 1, n_vars + 1))
        piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
        print("%d x %d cycle-check=%-5s %-9s %6d pivots  %.2f us/pivot  %7.0f pivots/s  aborts %d" % (
            n_rows + 1, n_vars + 1, check, t.last_path(), piv, best * 1e6 / piv, piv / best, t.get_counters()["resident_aborts"]), flush=True)
        t.close()
