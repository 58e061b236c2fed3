"""Round 6: what does the in-process device pool's hand-off cost per call?  The same Monster_II batch through ONE engine and through a pool
of M members on the one visible GPU, at batch sizes from M nodes (one per member: the call is all hand-off) to the bench's 2416; per size the median call of
each and the difference.  On one GPU the members time-share the chip (the large sizes price THAT); the small sizes price the hand-off a multi-GPU node pays.
  python tools/pool_handoff.py [members=4] [out.md]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jslpsolver_amd import Model, _capi
from jslpsolver_amd.engine import DevicePool, Tableau
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _capi.load_hip()
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
model = Model(g["model"])
m, vibr, vibc = model.build_tableau()
t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
t.applyCuts([], check_cycles=True)
t.save()
ints = [int(v) for v in model.integer_index_array]
t.set_watched_variables(ints)
pool = DevicePool(t, [0] * M)
pool.set_watched_variables(ints)
calls = [c["cuts"] or [] for c in g["simplexCalls"][1:]]


def med(fn, n):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts)), 1e6 * float(np.min(ts))


lines = ["| nodes in the batch | one engine, us per call (median / min) | pool of %d, us per call (median / min) | pool - one engine | one engine on nodes / %d (what a member's own GPU would do) |" % (M, M),
         "|---|---|---|---|---|"]
for n in (M, 4 * M, 16 * M, 151, 604, 2416):
    nodes = (calls * 17)[:n]
    packed = t.pack_cut_lists(nodes)
    share = t.pack_cut_lists(nodes[: max(n // M, 1)])
    one = med(lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False), 200)
    r1, rows1, vals1 = t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=True)
    mem = med(lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=share, copy=False), 200)
    pl = med(lambda: pool.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False), 200)
    rp, rowsp, valsp = pool.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=True)
    ok = all(rp[i].height == r1[i].height and rp[i].feasible == r1[i].feasible for i in range(n)) and np.array_equal(rowsp[:n], rows1[:n]) and np.array_equal(
        np.asarray(valsp[:n]).view(np.int64), np.asarray(vals1[:n]).view(np.int64))
    lines.append("| %d | %.1f / %.1f | %.1f / %.1f | %+.1f us | %.1f / %.1f (pool - that: %+.1f us)%s |" % (n, one[0], one[1], pl[0], pl[1], pl[0] - one[0], mem[0], mem[1], pl[0] - mem[0],
                                                                                                  "" if ok else " OUTCOMES DIFFER"))
    print(lines[-1], flush=True)
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
pool.close(); t.close()
