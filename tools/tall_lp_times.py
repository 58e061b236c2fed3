"""Dense LPs taller than 2048 rows: the tall register-resident geometry (512 lanes x 4 columns x 16 rows per workgroup)
against the fused one-launch-per-pivot pipeline and select + update.  usage: tools/tall_lp_times.py [rows] [cols]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, pivot_digest
m, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(12345)
A = np.zeros((m + 1, n + 1))
A[1:, 1:] = rng.integers(1, 21, (m, n))
A[0, 1:] = rng.integers(1, 51, n)
A[1:, 0] = rng.integers(100, 501, m)
vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
vibc = np.array([-1] + list(range(n)), dtype=np.int32)
t = Tableau(A, vibr, vibc, lib=_capi.load_hip())
t.save()
t.simplex(check_cycles=False)
best = 1e9
for _ in range(2):
    t.restore()
    t0 = time.perf_counter(); r = t.simplex(check_cycles=False); best = min(best, time.perf_counter() - t0)
piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
print(json.dumps({"path": t.last_path(), "pivots": piv, "seconds": best, "pivots_per_s": piv / best, "digest": pivot_digest(t.pivot_trace()[-piv:]),
                  "roofline_frac": 16.0 * (m + 1) * (n + 1) * piv / best / 8e12}))
""" % ROOT
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for mode in ("resident", "fused", "sp"):
    env = dict(os.environ, JSLP_FORCE_PATH=mode)
    out = subprocess.run([sys.executable, "-c", CHILD, str(m), str(n)], capture_output=True, text=True, env=env, timeout=1200)
    print(mode, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
