#!/usr/bin/env python3
"""Fold the two --pmc passes of one workload (tools/gpu_round.sh pmc / pmcrelax) into profiles/pmc_latest.json[key]:
HBM bytes per unit of work of the dominant kernel.
  tools/pmc_latest.py <run dir> <key: pivots|relax> <source label>
<run dir>/pmc_<key>_fetch and .../pmc_<key>_write hold the rocpd databases, <run dir>/pmc_<key>_{fetch,write}.log the
workload's JSON line (kernel, dispatches, units).  FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); FETCH_SIZE gets the
gfx950 x2 correction only for kernels that stream with 16 B/lane (MI355X_MICROARCH.md, HBM): stated per entry."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter_sum(run_dir, sub, counter, kernel):
    tot, n, wgs = 0.0, 0, 0
    for db in glob.glob(os.path.join(run_dir, sub, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
        val_col = "value" if "value" in cols else "counter_value"
        q = "select count(*), sum(%s), sum(grid_size / workgroup_size) from counters_collection where %s like ? and %s = ?" % (val_col, name_col, cnt_col)
        c, s, g = cur.execute(q, ("%" + kernel + "%", counter)).fetchone()
        tot += s or 0.0
        n += c or 0
        wgs += g or 0
    return tot, n, int(wgs)


def workload_line(path):
    with open(path) as fh:
        return json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])


def main(run_dir, key, source):
    w = workload_line(os.path.join(run_dir, "pmc_%s_fetch.log" % key))
    w2 = workload_line(os.path.join(run_dir, "pmc_%s_write.log" % key))
    assert w["units"] == w2["units"] and w["kernel"] == w2["kernel"]
    fetch_kib, nf, gf = counter_sum(run_dir, "pmc_%s_fetch" % key, "FETCH_SIZE", w["kernel"])
    write_kib, nw, gw = counter_sum(run_dir, "pmc_%s_write" % key, "WRITE_SIZE", w["kernel"])
    if key == "relax":  # one dispatch per call of the workload: the units are the nodes those calls evaluated
        assert nf == nw == w["dispatches"], (nf, nw, w["dispatches"])
    wide = key == "relax"  # the node kernels stream rows with 16 B per lane; the resident kernel issues 8-byte agent-scope loads
    fetch = fetch_kib * 1024.0 * (2.0 if wide else 1.0) / w["units"]
    write = write_kib * 1024.0 / w["units"]
    entry = {"kernel": w["kernel"], "workload": w["workload"], "unit": w["unit"], "units_counted": w["units"],
             "dispatches_fetch_pass": nf, "dispatches_write_pass": nw, "dispatches_expected": w["dispatches"],
             "fetch_bytes_per_unit": fetch, "write_bytes_per_unit": write, "traffic_bytes_per_unit": fetch + write,
             "algorithmic_bytes_per_unit": w["algorithmic_bytes_per_unit"],
             "traffic_over_algorithmic": (fetch + write) / w["algorithmic_bytes_per_unit"],
             "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py %s; KiB units; sums over the "
                       "kernel's dispatches divided by the units of work they did; FETCH_SIZE %s" % (
                           key, "x2 (gfx950 correction for 16 B/lane streaming reads)" if wide else
                           "RAW (the x2 correction is calibrated for wide streaming reads; this kernel issues 8-byte agent-scope loads: at most 2x the raw value)"),
             "source": source,
             # which kernel sources the passes ran on: bench.py reports `traffic` only when this equals its own tree's hash
             "kernel_sources_sha": __import__("bench").kernel_sources_sha(),
             "verified": w.get("verified")}
    if not w.get("verified") or not w2.get("verified"):
        raise SystemExit("refusing to fold an unverified workload into profiles/pmc_latest.json (tools/pmc_workload.py checks every solve)")
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if "kernel" in doc:  # the round-1 single-entry layout
            doc = {}
    except Exception:
        doc = {}
    doc["relaxations" if key == "relax" else "pivots"] = entry
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
