import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau
lib = _capi.load_hip()
os.environ["JSLP_FORCE_PATH"] = "resident"
m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 1000, 1000)
t = Tableau(m, vibr, vibc, lib=lib)
res = t.simplex(check_cycles=False)
print("pivots", len(t.pivot_trace()))
