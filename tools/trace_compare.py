"""Pivot trace of one dense integer LP (tools/tall_one.py's instance) through the current engine path against a stored oracle trace.
  python tools/trace_compare.py make rows cols      (CPU: the oracle's trace -> build/trace_<rows>x<cols>.npy)
  python tools/trace_compare.py check rows cols     (GPU: same instance through the HIP engine; first divergence, if any)"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, pivot_digest
mode, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(12345)
A = np.zeros((m + 1, n + 1))
A[1:, 1:] = rng.integers(1, 21, (m, n))
A[0, 1:] = rng.integers(1, 51, n)
A[1:, 0] = rng.integers(100, 501, m)
vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
vibc = np.array([-1] + list(range(n)), dtype=np.int32)
path = "build/trace_%dx%d.npy" % (m, n)
lib = _capi.Library("oracle/libjslp_oracle.so") if mode == "make" else _capi.load_hip()
t = Tableau(A, vibr, vibc, lib=lib)
t.save()
r = t.simplex(check_cycles=False)
piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
tr = np.asarray(t.pivot_trace())[-piv:]
for rep in range(int(os.environ.get("REPEATS", "0"))):  # the same solve again from the device-side snapshot
    t.restore()
    r = t.simplex(check_cycles=False)
    piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
    tr2 = np.asarray(t.pivot_trace())[-piv:]
    if mode != "make" and (len(tr2) != len(tr) or (tr2 != tr).any()):
        print(json.dumps({"repeat": rep + 1, "pivots": piv, "first_run_pivots": int(len(tr)), "note": "the repeat differs from the first run"}))
        tr = tr2
        break
if mode == "make":
    np.save(path, tr)
    print(json.dumps({"shape": [m, n], "pivots": piv, "digest": pivot_digest(tr[-piv:])}))
else:
    want = np.load(path)
    k = min(len(want), len(tr))
    diff = np.nonzero((want[:k] != tr[:k]).any(axis=1))[0]
    first = int(diff[0]) if len(diff) else (k if len(want) != len(tr) else -1)
    out = {"shape": [m, n], "path": t.last_path(), "pivots": piv, "want_pivots": int(len(want)), "first_divergence": first}
    if first >= 0 and first < k:
        out["want"] = want[first].tolist(); out["got"] = tr[first].tolist()
        out["before"] = want[max(0, first - 2):first].tolist()
    print(json.dumps(out))
