"""Wall time of SMALL node batches (what a real speculative tree issues: <= 16 nodes per call) on Monster_II.
  tools/small_batch_times.py [nodes per batch ...]"""
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
t.set_watched_variables([int(v) for v in model.integer_index_array])
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]:
    packs = [t.pack_cut_lists(base[i:i + n]) for i in range(0, 128, n)][:16]
    for p in packs:
        t.applyCutsBatch(None, check_cycles=True, packed=p, copy=False)
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 0.4:
        for p in packs:
            t.applyCutsBatch(None, check_cycles=True, packed=p, copy=False)
        reps += len(packs)
    dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter(); reps = 0
    while time.perf_counter() - t1 < 0.4:
        for p in packs:
            t.applyCutsBatchWatched(None, check_cycles=True, packed=p, copy=False)
        reps += len(packs)
    dw = (time.perf_counter() - t1) / reps
    print("batch of %3d nodes: %.1f us per call full read-back, %.1f us compact  [JSLP_SMALL_BATCH_1024=%s]" % (n, dt * 1e6, dw * 1e6, os.environ.get("JSLP_SMALL_BATCH_1024", "default")), flush=True)
