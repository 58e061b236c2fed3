"""How much of a batched relaxation call is the PCIe read-back?  Same Monster_II node batch with the full read-back
(RHS column + row map), the RHS column only, and (via the C ABI directly) the per-node states only."""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Model, _capi
from jslpsolver_amd.engine import Tableau
from jslpsolver_amd._capi import SimplexResult
lib = _capi.load_hip()
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
model = Model(g["model"])
m, vibr, vibc = model.build_tableau()
base = [c["cuts"] or [] for c in g["simplexCalls"][1:]]
t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
t.applyCuts([], check_cycles=True)
t.save()
for n in (1208, 2416):
    nodes = (base * ((n + len(base) - 1) // len(base)))[:n]
    packed = t.pack_cut_lists(nodes)
    n_nodes, offs, ty, v, x = packed
    out = (SimplexResult * n_nodes)()
    def states_only():
        lib.check(lib.jslp_engine_relax_batch_pinned(t._h, n_nodes, _capi.ptr_i32(offs), _capi.ptr_i8(ty), _capi.ptr_i32(v), _capi.ptr_f64(x), 1, out, None, None, None), "x")
    variants = {"rhs+rows": lambda: t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False),
                "rhs only": lambda: t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False, want_rows=False),
                "states only": states_only}
    for name, fn in variants.items():
        fn()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        print("nodes %5d  %-12s %8.1f us  %9.0f relax/s" % (n, name, best * 1e6, n / best), flush=True)
