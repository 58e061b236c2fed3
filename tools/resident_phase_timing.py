"""In-kernel phase timing of k_simplex_resident (s_memtime deltas per pivot phase) and micro-costs in its geometry.
Needs the debug build of the library:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DJSLP_DEBUG_RESIDENT \
        -o /tmp/libjslp_hip_dbg.so jslpsolver_amd/csrc/jslp_hip.hip
  JSLP_HIP_LIBRARY=/tmp/libjslp_hip_dbg.so python tools/resident_phase_timing.py 2000
(the debug build dumps its counters to gpurun_out/resident_r0.bin)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jslpsolver_amd import _capi, generators
from jslpsolver_amd.engine import Tableau
lib = _capi.load_hip()
os.environ["JSLP_FORCE_PATH"] = os.environ.get("JSLP_FORCE_PATH", "resident")  # ("xl": the XCD-local geometry)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else n  # (variables, constraints): 2000 4000 = the tall geometry
m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n_rows)
t = Tableau(m, vibr, vibc, lib=lib)
res = t.simplex(check_cycles=os.environ.get("PHASE_TIMING_CHECK", "0") == "1")  # (PHASE_TIMING_CHECK=1: the cycle check on)
print("pivots", len(t.pivot_trace()), "path", t.last_path())
d = np.fromfile("gpurun_out/resident_r0.bin", dtype=np.uint64)[12288:]
names = ["A cands", "B row stores", "C sweep/drain/sync", "D decide/poll", "E0 rowflag", "E row load+norm", "F update", "G price"]
if os.environ.get("JSLP_RES_LEAN", "1") != "0":  # the lean kernel's pipelined loop (jslp_resident_pipe.hip.h) marks other sections
    names = ["S claim + ratio test", "C1 poll+reduce", "U update+publish", "C2 drain+barrier", "D decide + E row fetch", "N/R0/commit", "G pricing round A", "loop top"]
for label, off in (("wg0", 0), ("wg100", 16), ("wgLast", 32)):
    acc = d[off:off + 8].astype(np.float64); ep = float(d[off + 8])
    print(label, "row-fetch retries", int(d[off + 9]), "of", int(ep), "pivots")
    print(label, "epochs", ep, " ".join("%s=%.0f" % (nm, a / max(ep, 1)) for nm, a in zip(names, acc)), "total cyc/pivot %.0f" % (acc.sum() / max(ep, 1)))
mc = d[64:69]
print("micro: syncthreads %d ticks, 1024-way LDS atomicMin %d ticks, dependent agent load %d ticks, dependent f64 div+add %d ticks" % (mc[0], mc[1], mc[2], mc[3]))
