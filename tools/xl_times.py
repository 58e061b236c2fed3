"""Per-pivot time of the mid-size LPs the XCD-local register-resident geometry takes, next to what round 3 shipped for them
(JSLP_XL=0: the chip-wide resident kernel or, for the sparse Monster LP, one LDS workgroup).  Every solve is checked against the
reference's golden (digest / first-call pivots) before its time is printed.   python tools/xl_times.py [out.md]"""
import gzip
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def cases():
    from jslpsolver_amd import Model, generators
    import known_answers as KA
    out = []
    for kind, n in (("ra", 500), ("lp", 500), ("ra", 1000), ("lp", 1000)):
        want = KA.expected_dense(kind, n, n)
        if kind == "ra":
            m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
        else:
            m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
        out.append(("%s %dx%d" % ("generateResourceAllocation" if kind == "ra" else "generateRandomLP", n + 1, n + 1), m, vibr, vibc, [], 1e-8, None, want["digest"], want["pivots"]))
    for name in ("Monster_Problem", "Monster_II"):
        with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", name + ".json.gz"), "rt") as fh:
            g = json.load(fh)
        model = Model(g["model"])
        m, vibr, vibc = model.build_tableau()
        call = g["simplexCalls"][0]
        n0 = call["p1"] + call["p2"]
        from jslpsolver_amd.engine import pivot_digest
        import numpy as np
        out.append(("%s root LP %dx%d" % (name, m.shape[0], m.shape[1]), m, vibr, vibc, model.unrestricted, model.precision,
                    m.shape[0] + 2 * len(model.integerVariables), pivot_digest(np.asarray(g["pivots"], dtype=np.int64).reshape(-1, 2)[:n0]), n0))
    return out


def run_mode():
    from jslpsolver_amd import _capi
    from jslpsolver_amd.engine import Tableau, pivot_digest
    lib = _capi.load_hip()
    rows = []
    for label, m, vibr, vibc, unr, prec, cap, digest, npiv in cases():
        for check in (False, True):
            t = Tableau(m, vibr, vibc, unr, precision=prec, row_capacity=cap, lib=lib)
            t.save()
            t.set_timing(True)
            best_wall, reps = 1e9, 7
            for _ in range(reps):
                t.restore()
                t0 = time.perf_counter()
                res = t.simplex(check_cycles=check)
                best_wall = min(best_wall, time.perf_counter() - t0)
                piv = res.pivots_phase1 + max(res.pivots_phase2, 0)
                d = pivot_digest(t.pivot_trace()[-piv:])
                if piv != npiv or d != digest:
                    raise SystemExit("WRONG ANSWER on %s: %d pivots %s, want %d %s" % (label, piv, d, npiv, digest))
            ms, n_units, _tot = t.get_timing()
            c = t.get_counters()
            if c["resident_aborts"]:
                raise SystemExit("resident abort on " + label)
            rows.append({"case": label, "check": check, "path": t.last_path(), "pivots": piv, "wall_us": best_wall * 1e6,
                         "kernel_us_per_pivot": (ms * 1e3 / n_units) if n_units else None, "wall_us_per_pivot": best_wall * 1e6 / max(piv, 1)})
            t.close()
    print(json.dumps(rows))


def main(out_md=None):
    res = {}
    for mode, env in (("xl", {"JSLP_XL": "1"}), ("round3", {"JSLP_XL": "0"})):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode"], capture_output=True, text=True, env=e, timeout=900)
        lines = [l for l in out.stdout.splitlines() if l.startswith("[")]
        if not lines:
            raise SystemExit(out.stdout[-2000:] + out.stderr[-2000:])
        res[mode] = json.loads(lines[-1])
    lines = ["| LP (every solve checked against the reference's pivots) | cycle check | path | pivots | simplex() wall us | us / pivot (wall) | kernel us / pivot | round-3 path | its wall us | its us / pivot |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for a, b in zip(res["xl"], res["round3"]):
        lines.append("| %s | %s | %s | %d | %.0f | %.2f | %s | %s | %.0f | %.2f |" % (
            a["case"], "on" if a["check"] else "off", a["path"], a["pivots"], a["wall_us"], a["wall_us_per_pivot"],
            "%.2f" % a["kernel_us_per_pivot"] if a["kernel_us_per_pivot"] else "-", b["path"], b["wall_us"], b["wall_us_per_pivot"]))
    text = "\n".join(lines) + "\n"
    print(text)
    if out_md:
        with open(out_md, "w") as fh:
            fh.write("Mid-size LPs: the XCD-local register-resident geometry (JSLP_XL=1) against the default policy (chip-wide resident kernel / one LDS workgroup).\n"
                     "`simplex()` wall = best of 7 from a device-side restore(); kernel time = HIP events around the cooperative launch.\n\n" + text)


if __name__ == "__main__":
    if "--mode" in sys.argv:
        run_mode()
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else None)
