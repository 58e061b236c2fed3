#!/usr/bin/env python3
"""Fold the SQ-counter passes of one workload (tools/gpu_round.sh sq / sqp) into profiles/pmc_latest.json[<key>_sq]: what the waves of the
dominant kernel DID per unit of work -- instructions issued by class, the share of wave-cycles spent waiting, and (pivots) the vector
instructions one wave issues per pivot next to the 32 that are the Gauss-Jordan update itself (16 cells per lane x v_mul_f64 + v_add_f64).
  tools/pmc_sq.py <run dir> <key: pivots|relax> <source label> [out.md [summary.json, default profiles/pmc_latest.json]]
<run dir>/pmc_sq_<key>*/ hold the rocpd databases (one --pmc pass each, <= 8 counters), <run dir>/pmc_sq_<key>.log the workload's JSON
line (tools/pmc_workload.py: kernel, dispatches, units, every solve checked against the reference's answer).  The entry is stamped with
the kernel sources' hash like the HBM entries: bench.py reports it only for the tree it was taken on."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

USEFUL_VALU_PER_WAVE_PER_PIVOT = 32  # headline geometry <1024, 2, 8>: 16 cells per lane, one v_mul_f64 + one v_add_f64 each (simplex.ts:376-387)


def counters(run_dir, key, kernel):
    """{counter: (sum over the kernel's dispatches, dispatches)} over every pass directory of this key"""
    out = {}
    for db in glob.glob(os.path.join(run_dir, "pmc_sq_%s*" % key, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
        val_col = "value" if "value" in cols else "counter_value"
        q = "select %s, count(*), sum(%s) from counters_collection where %s like ? group by %s" % (cnt_col, val_col, name_col, cnt_col)
        for cname, n, tot in cur.execute(q, ("%" + kernel + "%",)):
            out[cname] = (float(tot or 0.0), int(n or 0))
    return out


def main(run_dir, key, source, out_md=None, latest_path=None):
    with open(os.path.join(run_dir, "pmc_sq_%s.log" % key)) as fh:
        w = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
    if not w.get("verified"):
        raise SystemExit("refusing to fold an unverified workload into profiles/pmc_latest.json")
    c = counters(run_dir, key, w["kernel"])
    if not c:
        raise SystemExit("no counters of %s under %s/pmc_sq_%s*" % (w["kernel"], run_dir, key))
    for name, (_, n) in c.items():
        assert n == w["dispatches"], (name, n, w["dispatches"])
    units = float(w["units"])
    per_unit = {k: v[0] / units for k, v in sorted(c.items())}
    entry = {"kernel": w["kernel"], "workload": w["workload"], "unit": w["unit"], "units_counted": w["units"], "dispatches": w["dispatches"],
             "per_unit": per_unit, "source": source, "kernel_sources_sha": __import__("bench").kernel_sources_sha(), "verified": w.get("verified"),
             "method": "rocprofv3 --pmc <SQ counters>, one pass per <= 8 counters over tools/pmc_workload.py %s; sums over the kernel's "
                       "dispatches divided by the units of work they did" % key}
    g = lambda k: c.get(k, (None,))[0]
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_ANY") is not None:
            entry["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_ACTIVE_INST_ANY") is not None:
            entry["issuing_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        waves_per_dispatch = g("SQ_WAVES") / w["dispatches"]
        entry["waves_per_dispatch"] = waves_per_dispatch
        if key == "pivots":  # every wave lives for the whole solve: instructions per wave per pivot
            per_wave = g("SQ_INSTS_VALU") / (waves_per_dispatch * units)
            entry["valu_per_wave_per_pivot"] = per_wave
            entry["useful_valu_per_wave_per_pivot"] = USEFUL_VALU_PER_WAVE_PER_PIVOT
            entry["useful_valu_frac"] = USEFUL_VALU_PER_WAVE_PER_PIVOT / per_wave
            if g("SQ_INSTS_SALU") is not None:
                entry["salu_per_wave_per_pivot"] = g("SQ_INSTS_SALU") / (waves_per_dispatch * units)
            entry["note"] = ("DYNAMIC counts: the polls of the hand-over loops are in them (a wave that waits for the candidate row issues "
                             "instructions while it waits); the static mix of the same loop is tools/isa_mix.py's")
    path = latest_path or os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except Exception:
        doc = {}
    doc[("relaxations" if key == "relax" else "pivots") + "_sq"] = entry
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    lines = ["# SQ counters of %s (%s; %s)" % (w["kernel"], w["workload"], source), "",
             "kernel sources: %s; every solve of the workload checked against the reference's answer" % entry["kernel_sources_sha"], "",
             "| counter | sum over %d dispatches | per %s |" % (w["dispatches"], w["unit"]), "|---|---|---|"]
    for k, (tot, _) in sorted(c.items()):
        lines.append("| %s | %.4g | %.4g |" % (k, tot, tot / units))
    lines.append("")
    for k in ("wait_any_share_of_wave_cycles", "issuing_share_of_wave_cycles", "waves_per_dispatch", "valu_per_wave_per_pivot", "salu_per_wave_per_pivot",
              "useful_valu_frac"):
        if k in entry:
            lines.append("* %s = %.4g" % (k, entry[k]))
    text = "\n".join(lines) + "\n"
    if out_md:
        with open(out_md, "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:6])
