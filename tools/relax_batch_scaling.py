"""Time jslp_engine_relax_batch on the Monster_II node set for several batch sizes (latency- or bandwidth-bound?)."""
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
for n in (1, 8, 32, 128, 256, 512, 1024, 2048):
    nodes = (base * ((n + len(base) - 1) // len(base)))[:n]
    packed = t.pack_cut_lists(nodes)
    t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); res, _, _ = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False); best = min(best, time.perf_counter() - t0)
    piv = sum(res[i].pivots_phase1 + max(res[i].pivots_phase2, 0) for i in range(n))
    print("nodes %5d  %8.1f us  %7.2f us/node  %9.0f relax/s  pivots %d" % (n, best * 1e6, best * 1e6 / n, n / best, piv), flush=True)
