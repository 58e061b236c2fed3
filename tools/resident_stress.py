"""Repeat one dense LP through the register-resident kernel and compare every run's pivot trace and final tableau with the first
run's (and with the oracle's digest when given): python tools/resident_stress.py rows cols runs [fresh]
`fresh` = a new engine per run (upload + first launch each time) instead of restore() on one engine."""
import hashlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, pivot_digest
m, n, runs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fresh = len(sys.argv) > 4 and sys.argv[4] == "fresh"
rng = np.random.default_rng(int(os.environ.get("SEED", "12345")))
A = np.zeros((m + 1, n + 1))
A[1:, 1:] = rng.integers(1, 21, (m, n))
A[0, 1:] = rng.integers(1, 51, n)
A[1:, 0] = rng.integers(100, 501, m)
vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
vibc = np.array([-1] + list(range(n)), dtype=np.int32)
lib = _capi.load_hip()
check = os.environ.get("CHECK_CYCLES", "0") == "1"
ref = None
bad = []
t = None
for i in range(runs):
    if fresh or t is None:
        if t is not None:
            t.close()
        t = Tableau(A, vibr, vibc, lib=lib)
        t.save()
    else:
        t.restore()
    r = t.simplex(check_cycles=check)
    piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
    tr = np.asarray(t.pivot_trace()[-piv:], dtype=np.int64).reshape(-1, 2)
    out = (piv, pivot_digest(tr), hashlib.sha256(t.download()[0].tobytes()).hexdigest()[:16], t.last_path(), bool(r.optimal))
    if ref is None:
        ref = out
        ref_tr = tr
        print("run 0:", out, flush=True)
    elif out != ref:
        bad.append((i, out))
        k = min(len(tr), len(ref_tr))
        d = np.nonzero((tr[:k] != ref_tr[:k]).any(axis=1))[0]
        first = int(d[0]) if len(d) else k
        rpb = -(-(m + 1) // 256)
        print("run %d DIFFERS: %s; first differing pivot %d: got (row %d, col %d) want (row %d, col %d); owner workgroups %d / %d, rows per workgroup %d" % (
            i, out, first, tr[first][0] if first < len(tr) else -1, tr[first][1] if first < len(tr) else -1,
            ref_tr[first][0] if first < len(ref_tr) else -1, ref_tr[first][1] if first < len(ref_tr) else -1,
            (tr[first][0] // rpb) if first < len(tr) else -1, (ref_tr[first][0] // rpb) if first < len(ref_tr) else -1, rpb), flush=True)
        if first > 0:
            print("   previous pivot: (row %d, col %d), owner %d" % (ref_tr[first - 1][0], ref_tr[first - 1][1], ref_tr[first - 1][0] // rpb), flush=True)
print("%d x %d, %d runs (%s): %d differ from the first" % (m + 1, n + 1, runs, "fresh engines" if fresh else "one engine", len(bad)))
