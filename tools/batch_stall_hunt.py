"""Round 6: hunting the rare multi-millisecond stall of a batch call (profiles/r06_tree_latency.md): the 2416-node Monster_II batch through
jslp_engine_relax_batch_watched_pinned N times, every call timed; with JSLP_DEBUG_STALL=<ms> the engine says where a slow call's time went.
  JSLP_DEBUG_STALL=2 python tools/batch_stall_hunt.py [N=3000] [compact|full|sharded]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # (before the engine's library: torch initialises HIP itself)
torch.cuda.init()
from jslpsolver_amd import Model, _capi
from jslpsolver_amd.engine import Tableau
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
mode = sys.argv[2] if len(sys.argv) > 2 else "compact"
lib = _capi.load_hip()
with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
    g = json.load(fh)
model = Model(g["model"])
m, vibr, vibc = model.build_tableau()
t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=lib)
t.applyCuts([], check_cycles=True)
t.save()
nodes = [c["cuts"] or [] for c in g["simplexCalls"][1:]] * 16
packed = t.pack_cut_lists(nodes)
t.set_watched_variables([int(v) for v in model.integer_index_array])
if mode == "sharded":  # bench.py's sharded_batch leg: a one-rank RCCL group, engine call into device memory + all-gather + one D2H
    import socket
    import torch
    import torch.distributed as dist
    from jslpsolver_amd.sharding import evaluate_nodes_sharded_watched
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    fn = lambda: evaluate_nodes_sharded_watched(t, nodes, True, dist.group.WORLD, packed_mine=packed, copy=False)
elif mode == "full":
    fn = lambda: t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
else:
    fn = lambda: t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False)
time.sleep(float(os.environ.get("HUNT_IDLE_S", "0.5")))  # (an idle gap, as between two legs of bench.py: the clocks fall back)
warm = []
for _ in range(40):
    t0 = time.perf_counter(); fn(); warm.append(round((time.perf_counter() - t0) * 1e6))
print("the 40 warm-up calls after the idle gap (us):", warm)
import gc
gc.collect(); gc.disable()
ts = []
for i in range(N):
    t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
srt = np.sort(ts)
print("%s: %d calls, median %.0f us, p99 %.0f us, max %.0f us; calls > 3 x median: %s" % (mode, N, srt[N // 2], srt[int(0.99 * N)], srt[-1],
      [(int(i), int(ts[i])) for i in np.nonzero(ts > 3 * srt[N // 2])[0][:20]]))
