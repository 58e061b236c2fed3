"""Is the bimodal batch time per process or per kernel?  Two engines in ONE process (one-launch groups vs the four-launch
sequence), the same Monster_II node batch, calls interleaved."""
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
nodes = base * 16
def make(no_node_kernel):
    if no_node_kernel:
        os.environ["JSLP_NO_NODE_KERNEL"] = "1"
    else:
        os.environ.pop("JSLP_NO_NODE_KERNEL", None)
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    return t
engines = {"one-launch": make(False), "four-launch": make(True)}
packed = {k: t.pack_cut_lists(nodes) for k, t in engines.items()}
for k, t in engines.items():
    t.applyCutsBatch(None, check_cycles=True, packed=packed[k], copy=False)
    t.applyCutsBatch(None, check_cycles=True, packed=packed[k], copy=False)
times = {k: [] for k in engines}
for _ in range(8):
    for k, t in engines.items():
        t0 = time.perf_counter(); t.applyCutsBatch(None, check_cycles=True, packed=packed[k], copy=False); times[k].append(1e6 * (time.perf_counter() - t0))
for k, v in times.items():
    print(k, " ".join("%.0f" % x for x in v))
