#!/usr/bin/env python3
"""Where does a branch-and-bound node spend its time inside the one-workgroup kernel?  (debug build: -DJSLP_DEBUG_WGLDS,
JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so; the library prints the per-section cycle table on get_counters)
  tools/wglds_timing.py single|batch|rate"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Model, _capi
from jslpsolver_amd.engine import Tableau
lib = _capi.load_hip()
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
model = Model(g["model"])
m, vibr, vibc = model.build_tableau()
base = [c["cuts"] or [] for c in g["simplexCalls"][1:]]
t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
t.applyCuts([], check_cycles=True)
t.save()
mode = sys.argv[1]
if mode == "one":
    # round 6: the latency shape of bench.py's small_batch_latency -- node 3 of the reference's tree (1 pivot) or nodes 3..10 (17 pivots), compact read-back,
    # one call at a time; with the -DJSLP_DEBUG_WGLDS build get_counters() prints the kernel's section table for exactly these calls
    n_small = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sub = base[3:3 + n_small]
    t.set_watched_variables([int(v) for v in model.integer_index_array])
    packed = t.pack_cut_lists(sub)
    for _ in range(50):
        t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False)
    t.set_counting(True)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%d node(s) per call, 200 calls (counting on): median %.1f us, min %.1f us" % (n_small, 1e6 * ts[100], 1e6 * ts[0]), flush=True)
    print(t.get_counters(), flush=True)
elif mode == "single":
    for cuts in base[:20]:
        t.applyCuts(cuts, check_cycles=True)
    t.set_counting(True)
    t0 = time.perf_counter()
    for cuts in base:
        t.applyCuts(cuts, check_cycles=True)
    dt = time.perf_counter() - t0
    print("single nodes: %.1f us per relaxation (host wall, counting on)" % (1e6 * dt / len(base)), flush=True)
    print(t.get_counters(), flush=True)
    t.set_counting(False)
    t0 = time.perf_counter()
    for _ in range(3):
        for cuts in base:
            t.applyCuts(cuts, check_cycles=True)
    dt = time.perf_counter() - t0
    print("single nodes: %.1f us per relaxation (host wall, counting off)" % (1e6 * dt / (3 * len(base))), flush=True)
else:
    nodes = base * 16
    packed = t.pack_cut_lists(nodes)
    for _ in range(5):
        t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    if mode == "batch":
        t.set_counting(True)
        t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
        print(t.get_counters(), flush=True)
        t.set_counting(False)
    best = 1e9
    want_rows = os.environ.get("WANT_ROWS", "1") == "1"
    watched = os.environ.get("WATCHED", "0") == "1"
    if watched:
        t.set_watched_variables([int(v) for v in model.integer_index_array])
    for _ in range(8):
        t0 = time.perf_counter()
        if watched:
            t.applyCutsBatchWatched(None, check_cycles=True, packed=packed)
        else:
            t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False, want_rows=want_rows)
        best = min(best, time.perf_counter() - t0)
    if watched:
        print("compact read-back: %d watched variables per node" % t.n_watched)
    print("batch %d nodes: %.0f us, %.0f relaxations/s  [threads %s group %s wglds %s]" % (
        len(nodes), best * 1e6, len(nodes) / best, os.environ.get("JSLP_WG_BATCH_THREADS", "512"), os.environ.get("JSLP_GROUP_MAX", "1024"),
        "off" if os.environ.get("JSLP_NO_WGLDS") == "1" else "on"), flush=True)
