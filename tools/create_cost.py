"""Fixed cost of one engine: create + upload + tiny simplex + destroy, for a tiny and a mid-size tableau."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau
lib = _capi.load_hip()
for H, W in ((6, 3), (625, 553), (1722, 1641)):
    m = np.zeros((H, W)); m[1:, 0] = 1.0
    vibr = np.array([-1] + list(range(W - 1, W + H - 2)), dtype=np.int32)
    vibc = np.array([-1] + list(range(W - 1)), dtype=np.int32)
    for rep in range(3):
        t0 = time.perf_counter(); t = Tableau(m, vibr, vibc, lib=lib); t1 = time.perf_counter()
        t.simplex(); t2 = time.perf_counter(); t.read_rhs(); t3 = time.perf_counter(); t.close(); t4 = time.perf_counter()
    print("%5dx%-5d create+upload %.2f ms  simplex %.2f ms  read_rhs %.2f ms  destroy %.2f ms" % (H, W, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
