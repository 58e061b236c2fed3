#!/usr/bin/env python3
"""fp32-vs-fp64 tolerance sweep.  For each model and each tolerance (the reference's `precision`)
the root LP is solved (a) by the fp64 engine created with that tolerance and (b) by the fp32 twin of the same kernels
(jslp_engine_simplex_f32) from the same initial tableau; reported: flags, pivots, objective error against the fp64 run at
the reference's default 1e-8, whether the final basis is the same, device time of the pivot loop.
usage: tools/fp32_sweep.py [out.md]"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jslpsolver_amd import Model, Tableau, _capi, generators  # noqa: E402

TOLERANCES = [1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8]


def fixture_tableau(name):
    with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz"), "rt") as fh:
        g = json.load(fh)
    m = Model(g["model"])
    matrix, vibr, vibc = m.build_tableau()
    return matrix, vibr, vibc, m.unrestricted


def flags(r):
    return "feasible" if r.feasible and r.optimal else ("infeasible" if not r.feasible else ("unbounded" if not r.bounded else "stopped"))


def main(out_path=None):
    lib = _capi.load_hip()
    os.environ.setdefault("JSLP_FORCE_PATH", "sp")  # same launch shape for both widths: select + update per pivot
    cases = [("Vendor Selection root LP (config 5)",) + fixture_tableau("Vendor_Selection"),
             ("Monster LP (config 2)",) + fixture_tableau("Monster_Problem"),
             ("Monster_II root LP (config 4)",) + fixture_tableau("Monster_II")]
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 500, 500)
    cases.append(("dense resource allocation 500x500 (config 3a generator)", m, vibr, vibc, []))
    lines = ["| model | tolerance | fp64: outcome / pivots / objective | fp32: outcome / pivots / objective | fp32 objective rel. error vs fp64@1e-8 | same final basis | device ms fp64 / fp32 (both through k_select + k_update) | fp64 through the engine's DEFAULT path: device ms (path) |",
             "|---|---|---|---|---|---|---|---|"]
    for label, matrix, vibr, vibc, unr in cases:
        ref = Tableau(matrix, vibr, vibc, unr, precision=1e-8, lib=lib)
        r_ref = ref.simplex(check_cycles=False)
        ref.close()
        for tol in TOLERANCES:
            t = Tableau(matrix, vibr, vibc, unr, precision=tol, lib=lib)
            r32, rhs32, rows32, ms32 = t.simplex_f32(tol, check_cycles=False)
            t.set_timing(True)
            r64 = t.simplex(check_cycles=False)
            _, _, ms64 = t.get_timing()
            rhs64, rows64 = t.read_rhs()
            t.close()
            # the same solve the way the engine runs it without knobs (register-resident / one LDS workgroup / fused): what an
            # fp32 variant would have to beat, not the select + update pair
            forced = os.environ.pop("JSLP_FORCE_PATH", None)
            td = Tableau(matrix, vibr, vibc, unr, precision=tol, lib=lib)
            td.set_timing(True)
            td.simplex(check_cycles=False)
            _, _, msd = td.get_timing()
            pathd = td.last_path()
            td.close()
            if forced is not None:
                os.environ["JSLP_FORCE_PATH"] = forced
            err = abs(r32.obj_cell - r_ref.obj_cell) / max(1.0, abs(r_ref.obj_cell))
            lines.append("| %s | %.0e | %s / %d / %.9g | %s / %d / %.9g | %.2e | %s | %.1f / %.1f | %.2f (%s) |" % (
                label, tol, flags(r64), r64.pivots_phase1 + max(r64.pivots_phase2, 0), r64.obj_cell,
                flags(r32), r32.pivots_phase1 + max(r32.pivots_phase2, 0), r32.obj_cell, err,
                "yes" if np.array_equal(rows32, rows64) else "no", ms64, ms32, msd, pathd))
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:2])
