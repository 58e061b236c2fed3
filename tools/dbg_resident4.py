import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau, pivot_digest
lib = _capi.load_hip()
os.environ["JSLP_FORCE_PATH"] = "resident"
m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 2000, 2000)
for trial in range(3):
    t = Tableau(m, vibr, vibc, lib=lib)
    t0 = time.perf_counter(); res = t.simplex(check_cycles=False); dt = time.perf_counter() - t0
    n = len(t.pivot_trace())
    print(os.environ.get("JSLP_HIP_LIBRARY", "")[-12:], "pivots", n, pivot_digest(t.pivot_trace()), "us/pivot %.2f" % (1e6 * dt / n), flush=True)
    t.close()
