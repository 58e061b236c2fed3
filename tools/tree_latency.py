"""Latency of consecutive Solve(Monster_II) trees (the `relaxations.tree` leg of bench.py), call by call.
VERDICT r05 "weak" #5: one of 15 solves of a committed bench line took 592 ms instead of ~11 ms.  This tool repeats the tree N times with
every C-ABI entry point behind a timer, so that a slow solve names the engine call (or the host stretch between two calls) that ate the time.
  python tools/tree_latency.py [N=500] [out.md]
Environment the engine reads (JSLP_BATCH_POLL, JSLP_POOL_SPIN ...) is passed through; AMD_LOG_LEVEL can be set around a re-run.
Exit status 1 when max > 3 x median."""
import gc
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Solve, _capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
OUT = sys.argv[2] if len(sys.argv) > 2 else None
lib = _capi.Library(os.environ["TREE_LIB"]) if os.environ.get("TREE_LIB") else _capi.load_hip()  # (TREE_LIB: a CPU check of the tool itself against the oracle library)
calls = []  # (name, t_begin, seconds) of the current solve


def wrap(name, fn):
    def timed(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        calls.append((name, t0, time.perf_counter() - t0))
        return r
    return timed


for name in _capi.SYMBOLS:
    if name in ("jslp_last_error", "jslp_backend_name"):
        continue
    setattr(lib, name, wrap(name, getattr(lib, name)))

with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
spec = int(os.environ.get("TREE_SPECULATE", "8"))
want_iter = g["final"]["branchAndCutIterations"]
if os.environ.get("TREE_PRELUDE") == "pool":
    # what bench.py's relaxation legs do in front of their tree leg: a 2416-node batch on one engine, then the same over a pool of four virtual
    # devices (four more engines + worker threads), all closed again
    import numpy as np  # noqa: F401
    from jslpsolver_amd import Model
    from jslpsolver_amd.engine import DevicePool, Tableau
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    nodes = [c["cuts"] or [] for c in g["simplexCalls"][1:]] * 16
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    packed = t.pack_cut_lists(nodes)
    for _ in range(10):
        t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    pool = DevicePool(t, [0] * 4)
    for _ in range(10):
        pool.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    pool.set_watched_variables([int(v) for v in model.integer_index_array])
    for _ in range(10):
        pool.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False)
    pool.close()
    t.close()
for _ in range(3):
    Solve(g["model"], full=True, lib=lib, speculate=spec)
gc.collect()
gc.disable()
rows = []
for i in range(N):
    del calls[:]
    t0 = time.perf_counter()
    sol = Solve(g["model"], full=True, lib=lib, speculate=spec)
    el = time.perf_counter() - t0
    assert sol["iter"] == want_iter, (sol["iter"], want_iter)
    in_calls = sum(c[2] for c in calls)
    worst = max(calls, key=lambda c: c[2])
    # the longest host stretch between two engine calls
    gap, gap_after = 0.0, ""
    prev_end, prev_name = t0, "start"
    for nm, tb, d in calls:
        if tb - prev_end > gap:
            gap, gap_after = tb - prev_end, prev_name
        prev_end, prev_name = tb + d, nm
    rows.append({"i": i, "ms": 1e3 * el, "calls": len(calls), "in_calls_ms": 1e3 * in_calls, "worst_call": worst[0], "worst_call_ms": 1e3 * worst[2],
                 "worst_gap_ms": 1e3 * gap, "gap_after": gap_after,
                 "by_name": sorted(((nm, 1e3 * sum(c[2] for c in calls if c[0] == nm), sum(1 for c in calls if c[0] == nm)) for nm in set(c[0] for c in calls)), key=lambda x: -x[1])[:4]})
gc.enable()
ms = sorted(r["ms"] for r in rows)
med, p99, mx = ms[len(ms) // 2], ms[min(len(ms) - 1, int(0.99 * len(ms)))], ms[-1]
lines = ["# tools/tree_latency.py: %d consecutive Solve(Monster_II) (speculative batches of %d nodes, python host), every C-ABI call timed" % (N, spec),
         "", "env: " + " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(("JSLP_", "AMD_LOG", "HIP_", "GPU_MAX"))),
         "", "| solves | min ms | median ms | p99 ms | max ms | max / median |", "|---|---|---|---|---|---|",
         "| %d | %.2f | %.2f | %.2f | %.2f | %.2f |" % (N, ms[0], med, p99, mx, mx / med), "",
         "median solve: %.2f ms inside %d engine calls" % (sorted(r["in_calls_ms"] for r in rows)[len(rows) // 2], rows[0]["calls"]), "",
         "solves slower than 2 x median (index, ms, slowest engine call, longest host stretch between two calls):", ""]
slow = [r for r in rows if r["ms"] > 2 * med]
for r in slow[:40]:
    lines.append("* #%d: %.2f ms; slowest call %s %.2f ms; longest host stretch %.2f ms (after %s); top: %s" % (
        r["i"], r["ms"], r["worst_call"], r["worst_call_ms"], r["worst_gap_ms"], r["gap_after"], ", ".join("%s %.2f ms x%d" % x for x in r["by_name"])))
if not slow:
    lines.append("* none")
text = "\n".join(lines)
print(text)
if OUT:
    with open(OUT, "w") as fh:
        fh.write(text + "\n")
sys.exit(1 if mx > 3 * med else 0)
