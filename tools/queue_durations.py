#!/usr/bin/env python3
"""durations of the k_node_queue dispatches of a rocprofv3 --kernel-trace run, in launch order (tools/batch_modes.py ROUNDS=1:
2 warm-up calls per mode, then 4 timed calls per mode)"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
d = [r[0] / 1e3 for r in cur.execute("select end-start from kernels where name like '%k_node_queue%' order by start")]
print(len(d), "dispatches")
n_modes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
timed = d[2 * n_modes:]
for m in range(n_modes):
    print("mode %d: %s us" % (m, " ".join("%.0f" % x for x in timed[4 * m:4 * m + 4])))
