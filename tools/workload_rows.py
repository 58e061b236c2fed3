#!/usr/bin/env python3
"""One row per (kernel, workload): duration per dispatch and per unit of work from a rocprofv3 kernel trace, HBM bytes per unit
from the two PMC passes, and the three fractions bench.py reports -- so that the headline can be recomputed from profiles/ alone.
  tools/workload_rows.py <run dir> <out.md> [out.json]
<run dir>/<workload>/{kt,fetch,write}/**.db are the three rocprofv3 runs of `tools/pmc_workload.py <workload>` (kernel trace,
--pmc FETCH_SIZE, --pmc WRITE_SIZE: separate passes), <run dir>/<workload>/{kt,fetch,write}.log hold the workload's JSON line.
(tools/gpu_round.sh profw produces the layout.)"""
import glob
import json
import os
import sqlite3
import sys

HBM_PEAK = 8.0e12


def line(path):
    with open(path) as fh:
        return json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])


def kernel_durations(run, kernel):
    out = []
    for db in glob.glob(os.path.join(run, "kt", "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        out += [r[0] for r in cur.execute("select end - start from kernels where name like ? order by start", ("%" + kernel + "%",))]
    return out


def counter(run, sub, name, kernel):
    tot, n = 0.0, 0
    for db in glob.glob(os.path.join(run, sub, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
        val_col = "value" if "value" in cols else "counter_value"
        c, s = cur.execute("select count(*), sum(%s) from counters_collection where %s like ? and %s = ?" % (val_col, name_col, cnt_col),
                           ("%" + kernel + "%", name)).fetchone()
        tot += s or 0.0
        n += c or 0
    return tot, n


def main(root, out_md, out_json=None):
    rows = []
    for run in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.exists(os.path.join(run, "kt.log")):
            continue
        w = line(os.path.join(run, "kt.log"))
        # a row is only written for a workload whose solves were checked against the reference's answer (tools/pmc_workload.py
        # exits non-zero on a wrong one, so a log without a JSON line never gets here) -- in ALL three passes
        unverified = [p for p in ("kt", "fetch", "write") if os.path.exists(os.path.join(run, p + ".log")) and not line(os.path.join(run, p + ".log")).get("verified")]
        if unverified:
            print("SKIPPED %s: unverified solve in pass(es) %s" % (w.get("key"), ", ".join(unverified)))
            continue
        d = kernel_durations(run, w["kernel"])
        if w.get("one_dispatch_per") == "pivot":
            d = [x for x in d if x > 3000]  # the engine over-launches a little past the end of a solve; those kernels exit at once
        if not d:
            continue
        total_ns = float(sum(d))
        us_per_unit = total_ns / 1e3 / w["units"]
        row = {"workload": w["key"], "what": w["workload"], "kernel": w["kernel"], "dispatches": len(d), "dispatches_expected": w["dispatches"],
               "avg_dispatch_us": total_ns / 1e3 / len(d), "units": w["units"], "unit": w["unit"], "us_per_unit": us_per_unit,
               "units_per_s": 1e6 / us_per_unit, "algorithmic_bytes_per_unit": w["algorithmic_bytes_per_unit"],
               "algorithmic_gb_s": w["algorithmic_bytes_per_unit"] / us_per_unit / 1e3,
               "algorithmic_frac": w["algorithmic_bytes_per_unit"] / (us_per_unit * 1e-6) / HBM_PEAK,
               "verified": w.get("verified"), "health": w.get("health"),
               # (the streaming-path workloads also clock the whole solve on the host: selection kernels, launch gaps and read-backs included)
               "whole_solve_units_per_s": w.get("whole_solve_units_per_s"), "path": w.get("path"), "phase1_pivots": w.get("phase1_pivots")}
        try:
            wide = w["kernel"] != "k_simplex_resident"  # streaming kernels read 16 B per lane: gfx950 FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)
            f_kib, nf = counter(run, "fetch", "FETCH_SIZE", w["kernel"])
            w_kib, nw = counter(run, "write", "WRITE_SIZE", w["kernel"])
            uf, uw = line(os.path.join(run, "fetch.log"))["units"], line(os.path.join(run, "write.log"))["units"]
            fetch = f_kib * 1024.0 * (2.0 if wide else 1.0) / uf
            write = w_kib * 1024.0 / uw
            row.update({"pmc_fetch_bytes_per_unit": fetch, "pmc_write_bytes_per_unit": write, "pmc_traffic_bytes_per_unit": fetch + write,
                        "fetch_correction": "x2 (16 B/lane streaming reads)" if wide else "raw (narrow agent-scope loads)",
                        "hbm_gb_s": (fetch + write) / us_per_unit / 1e3, "hbm_frac": (fetch + write) / (us_per_unit * 1e-6) / HBM_PEAK,
                        "traffic_over_algorithmic": (fetch + write) / w["algorithmic_bytes_per_unit"]})
        except Exception as e:  # a run without the PMC passes still gives the timing rows
            row["pmc_error"] = repr(e)
        rows.append(row)
    lines = ["| workload | kernel | dispatches | avg dispatch us | units | us / unit | units / s | algorithmic MB / unit | algorithmic frac of 8 TB/s | PMC HBM MB / unit | hbm frac | whole solve, units / s (host clock) | verified against |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        v = r.get("verified") or {}
        lines.append("| %s | `%s` | %d | %.1f | %d %ss | %.3f | %.0f | %.2f | %.3f | %s | %s | %s | %s |" % (
            r["workload"], r["kernel"], r["dispatches"], r["avg_dispatch_us"], r["units"], r["unit"], r["us_per_unit"], r["units_per_s"],
            r["algorithmic_bytes_per_unit"] / 1e6, r["algorithmic_frac"],
            "%.3f" % (r["pmc_traffic_bytes_per_unit"] / 1e6) if "pmc_traffic_bytes_per_unit" in r else "-",
            "%.4f" % r["hbm_frac"] if "hbm_frac" in r else "-",
            "%.0f" % r["whole_solve_units_per_s"] if r.get("whole_solve_units_per_s") else "-",
            ("%s pivots, digest %s" % (v["pivots"], v["digest"])) if "digest" in v else ("%s reference node outcomes per call" % v.get("nodes_per_call", "?"))))
    text = "\n".join(lines) + "\n"
    with open(out_md, "w") as fh:
        fh.write("Every solve behind a row was checked against the reference's answer (pivot count, pivot digest, sha256 of the final tableau; node\n"
                 "outcomes for the relaxation batch) by tools/pmc_workload.py in each of the three passes; an unverified workload gets no row.\n"
                 "One row per (kernel, workload).  Durations: rocprofv3 --kernel-trace of `tools/pmc_workload.py <workload>` (sum of the kernel's dispatch\n"
                 "durations / units of work they did); HBM bytes: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the same command\n"
                 "(KiB units; FETCH_SIZE x2 for the 16 B/lane streaming kernels, raw for the resident kernel's narrow agent-scope loads).\n\n" + text)
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(rows, fh, indent=1)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:4])
