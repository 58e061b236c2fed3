"""Event stamps of eight consecutive pivots of the lean register-resident kernel (debug build: -DJSLP_DEBUG_RESIDENT, RT_STAMP in
jslp_resident_pipe.hip.h), thread 0 of every workgroup, on the 100 MHz clock all CUs share: where a pivot's 5.8 us go ACROSS workgroups --
who stores its summary last, how long after that the gathers close, when the winning row is in.
  JSLP_HIP_LIBRARY=build/lib..._dbg.so python tools/resident_phase_timing.py 2000 && python tools/resident_stamps.py [G]"""
import sys
import numpy as np
G = int(sys.argv[1]) if len(sys.argv) > 1 else 251
d = np.fromfile("gpurun_out/resident_stamps.bin", dtype=np.uint64).reshape(8, 256, 8)[:, :G, :].astype(np.int64)
names = ["summary stored", "update+publish issued", "my 64 summaries in", "gather closed", "row fetched", "at pricing", "priced"]
t0 = d[0, :, 0].min()
print("ticks of 10 ns; per event: min / median / max over %d workgroups, relative to the first summary of the first stamped pivot; [workgroup of the max]" % G)
for e in range(8):
    print("pivot +%d" % e)
    order = [5, 6, 0, 1, 2, 3, 4]  # program order inside an iteration: pricing, priced, summary, update, summaries in, gather closed, row fetched
    for k in order:
        v = d[e, :, k] - t0
        if (d[e, :, k] == 0).any():
            continue
        print("  %-24s %6d %6d %6d   [%3d]  spread %d" % (names[k], v.min(), int(np.median(v)), v.max(), int(v.argmax()), v.max() - v.min()))
per = np.diff(np.median(d[:, :, 0], axis=1))
print("period (median summary-to-summary): %s ticks" % per.tolist())
# the chain: last summary stored -> first / median / last gather closed; last row fetched -> last summary of the next pivot
for e in range(7):
    ls = d[e, :, 0].max(); gc = d[e, :, 3]; rf = d[e, :, 4]; ns = d[e + 1, :, 0]
    print("pivot +%d: last summary -> gather closed min/med/max %d/%d/%d; -> row fetched med/max %d/%d; row fetched(max) -> next summary med/max %d/%d; workgroup last to store its summary: %d (next: %d)" % (
        e, gc.min() - ls, int(np.median(gc)) - ls, gc.max() - ls, int(np.median(rf)) - ls, rf.max() - ls, int(np.median(ns)) - rf.max(), ns.max() - rf.max(), int(d[e, :, 0].argmax()), int(ns.argmax())))
