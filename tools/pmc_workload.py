#!/usr/bin/env python3
"""The two workloads of bench.py in a form whose units of work can be matched to kernel dispatches exactly, for the
rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs: tools/gpu_round.sh pmc).
  tools/pmc_workload.py pivots [n]   -> K solves of config 3a (k_simplex_resident: one dispatch per solve)
  tools/pmc_workload.py relax        -> the 2416-node Monster_II batch, K one-launch calls (k_node_queue: one dispatch per call)
Prints one JSON line: which kernel, how many dispatches of it to expect, how many units (pivots / relaxations) they did."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jslpsolver_amd import Model, _capi, generators  # noqa: E402
from jslpsolver_amd.engine import Tableau, pivot_digest  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import known_answers as KA  # noqa: E402


def _expectation(kind, n_vars, n_rows):
    """the reference's answer for this instance; without one no row may be written (JSLP_ALLOW_UNVERIFIED=1 overrides)"""
    want = KA.expected_dense(kind, n_vars, n_rows)
    if want is None and not KA.unverified_allowed():
        raise SystemExit("no reference answer for %s %d x %d under tests/golden: refusing to profile an unverified solve "
                         "(JSLP_ALLOW_UNVERIFIED=1 to override)" % (kind, n_vars, n_rows))
    return want


def _health(t):
    """the register-resident kernels must not have been rolled back or handed on behind the tool's back"""
    c = t.get_counters()
    if c["resident_aborts"]:
        raise SystemExit("resident kernel aborted %d time(s): the timing would be the streaming fallback's" % c["resident_aborts"])
    return {"resident_launches": c["resident_launches"], "resident_aborts": c["resident_aborts"], "resident_handovers": c["resident_handovers"]}


def pivots(n=2000, solves=2):
    lib = _capi.load_hip()
    want = _expectation("ra", n, n)
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    t = Tableau(m, vibr, vibc, lib=lib)
    t.save()
    total = 0
    for i in range(solves):
        t.restore()
        r = t.simplex(check_cycles=False)
        total += r.pivots_phase1 + max(r.pivots_phase2, 0)
        if want:
            KA.check(KA.solve_signature(t, r, pivot_digest), want, "config 3a %d x %d, solve %d" % (n, n, i))
    path = t.last_path()
    health = _health(t)
    t.close()
    H, W = m.shape
    print(json.dumps({"key": "pivots", "kernel": "k_simplex_resident" if path == "resident" else path, "dispatches": solves,
                      "units": total, "unit": "pivot", "algorithmic_bytes_per_unit": 16.0 * H * W,
                      "verified": ({"pivots": want["pivots"], "digest": want["digest"], "final_sha": want["final_sha"][:16], "source": want["source"]} if want else None),
                      "health": health,
                      "workload": "config 3a %dx%d fp64, %d solves" % (H, W, solves)}))


def dense(kind, n_vars, n_rows, check, solves=2, key=None):
    """one dense LP solved `solves` times from the device-side snapshot; the kernel that ran is read back from the engine"""
    lib = _capi.load_hip()
    want = _expectation(kind, n_vars, n_rows)  # (the cycle check changes nothing on these instances unless it HITS -- which would change the answer)
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n_vars, n_rows)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n_vars, n_rows)
    t = Tableau(m, vibr, vibc, lib=lib)
    t.save()
    total, p1 = 0, 0
    for i in range(solves):
        t.restore()
        r = t.simplex(check_cycles=check)
        total += r.pivots_phase1 + max(r.pivots_phase2, 0)
        p1 += r.pivots_phase1
        if want:
            KA.check(KA.solve_signature(t, r, pivot_digest), want, "%s %d x %d, solve %d" % (kind, n_vars, n_rows, i))
    path = t.last_path()
    health = _health(t)
    t.close()
    H, W = m.shape
    one_launch = path == "resident"
    kernel = {"resident": "k_simplex_resident", "fused": "k_fused_p1" if p1 == total else "k_pivot_fused", "select+update": "k_update"}.get(path, path)
    print(json.dumps({"key": key or ("%s_%dx%d_%s" % (kind, H, W, "check" if check else "nocheck")), "kernel": kernel,
                      # the resident kernel runs a whole solve per dispatch; the streaming kernels one pivot per dispatch (plus a few
                      # empty over-launches past the end of a solve, which the summary drops by duration)
                      "dispatches": solves if one_launch else total, "units": total, "unit": "pivot", "one_dispatch_per": "solve" if one_launch else "pivot",
                      "algorithmic_bytes_per_unit": 16.0 * H * W, "path": path, "phase1_pivots": p1,
                      "verified": ({"pivots": want["pivots"], "digest": want["digest"], "final_sha": want["final_sha"][:16], "source": want["source"]} if want else None),
                      "health": health,
                      "workload": "%s %dx%d fp64, cycle check %s, %d solves" % (
                          "generateResourceAllocation" if kind == "ra" else "generateRandomLP", H, W, "on" if check else "off", solves)}))


def stream(kind, m, n, key):
    """Round 5: a dense LP BEYOND the register file (tools/resident_stress.py's integer instances: m constraints x n variables, seed 12345)
    through the DEFAULT policy -- k_pivot_fused<1|2> (+ k_fused_p1 when the instance has a phase 1) or, past ld = 4096, k_select +
    k_update -- against the known answer of tests/golden/stress_expect.json.  This is the kernel north_star's ">= 40 % of the HBM roofline
    in rocprof" literally describes: 16 x H x W algorithmic bytes per pivot, one dispatch per pivot."""
    import time
    from resident_stress import int_instance
    lib = _capi.load_hip()
    want = KA.expected_stress(kind, m + 1, n + 1, 12345)
    if want is None and not KA.unverified_allowed():
        raise SystemExit("no known answer for %s %d x %d (tests/golden/gen_stress_expect.py): refusing to profile an unverified solve" % (kind, m + 1, n + 1))
    A, vibr, vibc = int_instance(m, n, 12345, kind == "int2p")
    t = Tableau(A, vibr, vibc, lib=lib)
    t.save()
    t0 = time.perf_counter()
    r = t.simplex(check_cycles=False)
    wall = time.perf_counter() - t0
    total, p1 = r.pivots_phase1 + max(r.pivots_phase2, 0), r.pivots_phase1
    if want:
        KA.check(KA.solve_signature(t, r, pivot_digest), want, "%s %d x %d" % (kind, m + 1, n + 1))
    path = t.last_path()
    c = t.get_counters()
    t.close()
    H, W = A.shape
    if c["resident_launches"]:
        raise SystemExit("%d x %d took a register-resident launch: not a streaming-path workload" % (H, W))
    kernel = {"fused": "k_pivot_fused", "select+update": "k_update"}.get(path, path)
    print(json.dumps({"key": key, "kernel": kernel, "dispatches": total - (p1 if path == "fused" else 0), "units": total - (p1 if path == "fused" else 0), "unit": "pivot",
                      "one_dispatch_per": "pivot", "algorithmic_bytes_per_unit": 16.0 * H * W, "path": path, "phase1_pivots": p1,
                      "whole_solve_units_per_s": total / wall, "whole_solve_seconds": wall,
                      "verified": ({"pivots": want["pivots"], "digest": want["digest"], "final_sha": want["final_sha"][:16], "source": want["source"]} if want else None),
                      "health": {"resident_launches": c["resident_launches"], "resident_aborts": c["resident_aborts"], "resident_handovers": c["resident_handovers"]},
                      "workload": "dense integer LP %dx%d fp64 (tools/resident_stress.py %s, seed 12345), default policy, cycle check off, 1 solve" % (H, W, kind)}))


WORKLOADS = {
    "stream_5001x3001": lambda: stream("int", 5000, 3000, "stream_5001x3001"),      # k_pivot_fused<2>
    "stream_5001x2001": lambda: stream("int", 5000, 2000, "stream_5001x2001"),      # k_pivot_fused<1>
    "stream_3001x5001": lambda: stream("int", 3000, 5000, "stream_3001x5001"),      # ld > 4096: k_pivot_fused<3> (JSLP_FUSED_WIDE=0: k_select + k_update)
    "stream2p_5001x3001": lambda: stream("int2p", 5000, 3000, "stream2p_5001x3001"),  # k_fused_p1<2> first
    "3a": lambda: dense("ra", 2000, 2000, False, key="3a"),
    "3a_check": lambda: dense("ra", 2000, 2000, True, key="3a_check"),
    "3b": lambda: dense("lp", 2000, 2000, False, solves=4, key="3b"),
    "tall_4001x2001": lambda: dense("ra", 2000, 4000, False, solves=1, key="tall_4001x2001"),
    "wide_2001x4001": lambda: dense("ra", 4000, 2000, False, solves=1, key="wide_2001x4001"),
    "big_3001x3001": lambda: dense("ra", 3000, 3000, False, solves=1, key="big_3001x3001"),
}


def relax(calls=12, reps=16):
    lib = _capi.load_hip()
    with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
        g = json.load(fh)
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    H, W = m.shape
    cap = H + 2 * len(model.integerVariables)
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=cap, lib=lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    nodes = [c["cuts"] or [] for c in g["simplexCalls"][1:]] * reps
    packed = t.pack_cut_lists(nodes)
    t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)  # first call: slots allocated + restored in full (other kernels)
    import hashlib
    import numpy as np
    ref_calls = g["simplexCalls"][1:]
    for _ in range(calls):
        results, rhs, rows = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
        for i in range(len(nodes)):  # every node of every profiled call against the reference's own relaxation outcome
            call = ref_calls[i % len(ref_calls)]
            h = results[i].height
            sha = hashlib.sha256(np.ascontiguousarray(rhs[i, :h]).tobytes() + np.ascontiguousarray(rows[i, :h]).tobytes()).hexdigest()
            if h != call["height"] or bool(results[i].feasible) != call["feasible"] or sha != call["rhsSha"]:
                raise SystemExit("WRONG ANSWER: node %d of the Monster_II batch differs from the reference's relaxation outcome" % i)
    # the same batch once more with the work counters on (host-side only here: the counting pass is not profiled separately)
    t.set_counting(True)
    t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
    c = t.get_counters()
    t.set_counting(False)
    t.close()
    sys.path.insert(0, ROOT)
    import bench
    per_node = bench.gated_bytes(c, W, W + 2 * cap + 2, H) / c["relaxations"]
    # every call after the first is ONE dispatch of k_node_queue (resident workgroups pulling the nodes from a queue)
    print(json.dumps({"key": "relaxations", "kernel": "k_node_queue", "dispatches": calls + 1, "units": len(nodes) * (calls + 1),
                      "unit": "LP relaxation", "algorithmic_bytes_per_unit": per_node,
                      "verified": {"nodes_per_call": len(nodes), "source": "tests/golden/fixtures/Monster_II.json.gz: rhsSha / height / feasible of every relaxation (the reference's own run)"},
                      "workload": "Monster_II %d-node batch, %d one-launch calls" % (len(nodes), calls + 1)}))


if __name__ == "__main__":
    if sys.argv[1] == "pivots":
        pivots(int(sys.argv[2]) if len(sys.argv) > 2 else 2000)
    elif sys.argv[1] in WORKLOADS:
        WORKLOADS[sys.argv[1]]()
    else:
        relax()
