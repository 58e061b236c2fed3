#!/usr/bin/env python3
"""Read-back sensitivity of the node batch inside ONE process (the run-to-run spread between processes is ~15 %): the 2416-node
Monster_II batch with the full read-back (RHS column + row map), the RHS column only, and the compact one (watched variables),
interleaved, several rounds."""
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
nodes = base * int(os.environ.get("REPS", "16"))
packed = t.pack_cut_lists(nodes)
modes = {
    "full (rhs + rows, %d B/node)" % (12 * m.shape[0]): lambda: t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False),
    "rhs only": lambda: t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False, want_rows=False),
    "watched (%d B/node)" % (12 * t.n_watched): lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=packed),
    "watched, pinned": lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False),
}
import numpy as np
_n, _offs, _t, _v, _x = packed
_out = (_capi.SimplexResult * _n)()
_wr = np.empty((_n, t.n_watched), dtype=np.int32)
_wv = np.empty((_n, t.n_watched), dtype=np.float64)
def _raw(fn, *tail):
    lib.check(getattr(lib, fn)(t._h, _n, _capi.ptr_i32(_offs), _capi.ptr_i8(_t), _capi.ptr_i32(_v), _capi.ptr_f64(_x), 1, _out, *tail), fn)
modes["states only"] = lambda: _raw("jslp_engine_relax_batch", None, None, t.row_capacity)
modes["watched values only"] = lambda: _raw("jslp_engine_relax_batch_watched", None, _capi.ptr_f64(_wv))
modes["watched rows only"] = lambda: _raw("jslp_engine_relax_batch_watched", _capi.ptr_i32(_wr), None)
for f in modes.values():
    f(); f()
best = {k: 1e9 for k in modes}
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    line = []
    for k, f in modes.items():
        b = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
        best[k] = min(best[k], b)
        line.append("%s %.0f us" % (k.split("(")[0].strip(), b * 1e6))
    print("round %d: %s" % (rnd, ", ".join(line)), flush=True)
for k, b in best.items():
    print("%-34s %6.0f us  %.2f M relaxations/s" % (k, b * 1e6, len(nodes) / b / 1e6))
