"""Differential fuzz of the register-resident kernels against the oracle on mid-size dense LPs with integer data (exact ties,
degenerate rows, two phases, optional objectives, cycle check on / off), shapes that exercise every geometry and partial grids.
  python tools/fuzz_resident.py make [n]     CPU: instances + the oracle's traces -> build/fuzz_resident.npz
  python tools/fuzz_resident.py check        GPU: the same instances through the HIP engine (JSLP_FORCE_PATH=resident), all compared"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, pivot_digest

PATH = "build/fuzz_resident.npz"


def instance(k):
    rng = np.random.default_rng(9000 + k)
    shapes = [(40, 60), (300, 500), (700, 900), (1000, 1400), (260, 2040), (2040, 300), (511, 513), (1500, 700), (1200, 2100), (2300, 400),
              (2049, 64), (64, 2049), (900, 900), (257, 1000), (1800, 1800), (3000, 600), (600, 3000), (128, 128), (2047, 2047), (1024, 2600)]
    m, n = shapes[k % len(shapes)]
    A = np.zeros((m + 1, n + 1))
    dens = [1.0, 0.6, 0.3][k % 3]
    A[1:, 1:] = np.where(rng.random((m, n)) < dens, rng.integers(1, 13, (m, n)), 0)
    A[0, 1:] = rng.integers(0, 25, n)
    A[1:, 0] = rng.integers(0 if k % 4 == 2 else 40, 300, m)  # k % 4 == 2: some RHS exactly 0 (degenerate rows), no ">=" rows
    n_ge = [0, 5, 0, 9][k % 4]
    if n_ge:  # ">=" rows x_j >= 1..3 (negated, negative RHS): phase-1 pivots; the "<=" rows (RHS >= 40, coefficients <= 12) stay satisfiable
        ge = rng.choice(np.arange(1, m + 1), min(n_ge, m), replace=False)
        A[ge, 0] = -rng.integers(1, 4, len(ge))
        A[ge, 1:] = 0.0
        A[ge, 1 + rng.choice(n, len(ge), replace=False)] = -1.0
    oo = None
    if k % 5 == 2 and n + 1 <= 2048 and m + 1 <= 2048:  # optional objectives: the headline geometry only
        oo = np.zeros((1 + k % 3, n + 1))
        for o in range(oo.shape[0]):
            cols = rng.choice(n, max(2, n // 4), replace=False)
            oo[o, 1 + cols] = rng.integers(-6, 20, len(cols))
    vibr = np.array([-1] + list(range(n, n + m)), dtype=np.int32)
    vibc = np.array([-1] + list(range(n)), dtype=np.int32)
    return A, vibr, vibc, oo, bool((k // 2) % 2)


def run(lib, k, cap=None):
    A, vibr, vibc, oo, check = instance(k)
    t = Tableau(A, vibr, vibc, [], lib=lib, optional_objectives=oo)
    r = t.simplex(check_cycles=check)
    tr = np.asarray(t.pivot_trace(), dtype=np.int32).reshape(-1, 2)
    fm = t.download()[0]
    out = (r.as_dict(), tr, fm.tobytes(), t.last_path())
    t.close()
    return out


mode = sys.argv[1]
if mode == "make":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    lib = _capi.Library("oracle/libjslp_oracle.so")
    store = {}
    for k in range(n):
        t0 = time.time()
        d, tr, fm, _ = run(lib, k)
        import hashlib
        store["trace%d" % k] = tr
        store["meta%d" % k] = np.frombuffer(json.dumps({"res": d, "sha": hashlib.sha256(fm).hexdigest()}).encode(), dtype=np.uint8)
        print(k, instance(k)[0].shape, d["pivots_phase1"], d["pivots_phase2"], d["feasible"], d["cycle_phase"], "%.1fs" % (time.time() - t0), flush=True)
    np.savez_compressed(PATH, n=np.array([n]), **store)
else:
    import hashlib
    os.environ.setdefault("JSLP_FORCE_PATH", "resident")
    lib = _capi.load_hip()
    z = np.load(PATH)
    bad = 0
    paths = {}
    only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]  # FUZZ_ONLY=8,28: those instances only
    for k in range(int(z["n"][0])):
        if only and k not in only:
            continue
        meta = json.loads(bytes(z["meta%d" % k]).decode())
        d, tr, fm, path = run(lib, k)
        paths[path] = paths.get(path, 0) + 1
        want = z["trace%d" % k]
        ok = d == meta["res"] and tr.shape == want.shape and (tr == want).all() and hashlib.sha256(fm).hexdigest() == meta["sha"]
        if not ok:
            bad += 1
            first = -1
            kk = min(len(tr), len(want))
            diff = np.nonzero((tr[:kk] != want[:kk]).any(axis=1))[0]
            first = int(diff[0]) if len(diff) else kk
            print("MISMATCH instance", k, instance(k)[0].shape, path, "pivots", len(tr), "want", len(want), "first divergence", first, {kk2: (d[kk2], meta["res"][kk2]) for kk2 in d if d[kk2] != meta["res"][kk2]})
    print(json.dumps({"instances": int(z["n"][0]), "mismatches": bad, "paths": paths}))
