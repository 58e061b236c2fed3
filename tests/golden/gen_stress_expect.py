"""Known answers for the instances tools/resident_stress.py repeats on the GPU (TEST INFRASTRUCTURE: this script runs the C
restatement under oracle/ -- which tests/test_oracle_golden.py pins against the reference's own goldens -- and commits what it
answered as DATA, tests/golden/stress_expect.json; the tool itself never touches oracle/).
  python tests/golden/gen_stress_expect.py [key filter]
Keys: <kind>_<H>x<W>_seed<seed>, H x W = the tableau's shape.  Existing entries are kept (the big ones take minutes)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
from jslpsolver_amd import _capi  # noqa: E402
from jslpsolver_amd.engine import Tableau, pivot_digest  # noqa: E402
from resident_stress import int_instance  # noqa: E402  (the instance builder is shared with the tool: one definition)

OUT = os.environ.get("JSLP_STRESS_EXPECT_OUT") or os.path.join(ROOT, "tests", "golden", "stress_expect.json")  # (the override: several keys generated side by side, merged by hand)
# (kind, constraints m, variables n, seed): every register-resident geometry and both pipelines
#   <1024,2,8> headline: 1000 x 1000, 2000 x 2000;  <512,4,16> tall: 2100 x 300, 4000 x 2000;  <512,6,12>: 1200 x 2100, 3000 x 3000;
#   <512,8,8> wide: 600 x 3000, 2000 x 4000;  int2p = with a phase 1
CASES = [("int", 1000, 1000, 12345), ("int", 2000, 2000, 12345), ("int", 2100, 300, 12345), ("int", 1200, 2100, 12345),
         ("int", 600, 3000, 12345), ("int", 300, 2100, 12345), ("int", 2000, 4000, 12345), ("int", 4000, 2000, 12345),
         ("int", 3000, 3000, 12345),
         ("int2p", 1000, 1000, 12345), ("int2p", 2100, 300, 12345), ("int2p", 300, 2100, 12345), ("int2p", 1200, 2100, 12345),
         ("intunr3", 1200, 2100, 12345), ("intunr3", 1000, 1000, 12345),  # the first 3 variables unrestricted: the general build
         # round 5 -- beyond the register file, the shapes the DEFAULT policy streams: 5001 x 3001 (k_pivot_fused<2>; int2p: k_fused_p1<2> first),
         # 5001 x 2001 (k_pivot_fused<1>), 3001 x 5001 (ld > 4096: k_select + k_update for both phases)
         ("int", 5000, 3000, 12345), ("int2p", 5000, 3000, 12345), ("int", 5000, 2000, 12345), ("int", 3000, 5000, 12345), ("int2p", 3000, 5000, 12345)]


def main(filt=""):
    table = {}
    if os.path.exists(OUT):
        with open(OUT) as fh:
            table = json.load(fh)
    lib = _capi.Library(os.path.join(ROOT, "oracle", "libjslp_oracle.so"))
    for kind, m, n, seed in CASES:
        key = "%s_%dx%d_seed%d" % (kind, m + 1, n + 1, seed)
        if filt not in key or key in table:
            continue
        A, vibr, vibc = int_instance(m, n, seed, kind == "int2p")
        t0 = time.time()
        t = Tableau(A, vibr, vibc, list(range(int(kind[6:]))) if kind.startswith("intunr") else [], lib=lib)
        r = t.simplex(check_cycles=False)
        piv = r.pivots_phase1 + max(r.pivots_phase2, 0)
        tr = np.asarray(t.pivot_trace(), dtype=np.int64).reshape(-1, 2)
        table[key] = {"pivots": int(piv), "pivots_phase1": int(r.pivots_phase1), "digest": pivot_digest(tr),
                      "final_sha": hashlib.sha256(np.ascontiguousarray(t.download()[0]).tobytes()).hexdigest(),
                      "feasible": bool(r.feasible), "bounded": bool(r.bounded), "optimal": bool(r.optimal)}
        t.close()
        print(key, table[key]["pivots"], table[key]["digest"], "%.1f s" % (time.time() - t0), flush=True)
        with open(OUT, "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
