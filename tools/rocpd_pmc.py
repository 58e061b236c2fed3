#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one `--pmc` pass per database).
usage: tools/rocpd_pmc.py <dir with */*.db> [out.json]"""
import glob
import json
import os
import sqlite3
import sys


def summarise(db):
    cur = sqlite3.connect(db).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in views:
        return {"error": "no counters_collection view", "views": views[:80]}
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    cnt_col = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
    val_col = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    if not (name_col and cnt_col and val_col):
        return {"error": "unexpected schema", "columns": cols}
    out = {}
    q = "select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s" % (
        name_col, cnt_col, val_col, val_col, name_col, cnt_col)
    for kname, cname, n, avg, tot in cur.execute(q):
        out.setdefault(kname, {})[cname] = {"dispatches": n, "avg": avg, "sum": tot}
    return out


def main(root, out_path=None):
    result = {}
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        result[os.path.relpath(db, root)] = summarise(db)
    text = json.dumps(result, indent=1)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)
    print(text[:6000])


if __name__ == "__main__":
    main(*sys.argv[1:3])
