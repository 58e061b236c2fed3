#!/usr/bin/env python3
"""bench.py -- headline measurement (BASELINE.json): simplex pivots/sec on the synthetic dense LP
generateResourceAllocation({seed:12345, numVariables:2000, numConstraints:2000, density:1.0}) (config 3a:
2001 x 2001 fp64 tableau, 9726 pivots, cycle check off -- the FASTER reference setting), 1 GPU; with
--gpus N every rank solves its own replica (a single LP does not shard: SURVEY.md 8e "replicas only"), and
the LP-relaxation throughput of the sharded branch-and-bound workload (config 4, Monster_II node batch,
nodes split across ranks, no data-path collective) rides along in "relaxations".

A "step" = one complete simplex() of the workload with the tableau already resident in HBM (restored from
the device-side snapshot; the 32 MB host upload happens once, outside the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)


def cpu_baseline(n, sample_pivots):
    """The reference itself (oracle/_ref, type-erased TypeScript under node, 1 thread) on the SAME instance,
    stopped after `sample_pivots` pivots of its simplex(); time = simplex entry -> last sampled pivot."""
    script = os.path.join(ROOT, "oracle", "ref_pivot_rate.js")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "src", "solver.js")):
        return None
    try:
        out = subprocess.run(["node", "--max-old-space-size=8192", script, str(n), str(sample_pivots)],
                             capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {"value": r["pivots_per_sec"], "unit": "pivots/s", "cores": 1, "kind": "reference",
                "sample": "first %d of the %d pivots of the same %dx%d instance, reference TS (type-erased) under node %s, "
                          "options.exitOnCycles=false, simplex() time only; host has %d cores"
                          % (r["pivots"], 9726 if n == 2000 else -1, n + 1, n + 1, r["node"], os.cpu_count())}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "pivots/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}


def pmc_traffic(H, W, kernel):
    """HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 FETCH_SIZE x2 correction): counters cannot be read from inside this process, so the
    committed summary of the latest pass on this workload is reported, with its provenance."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        if abs(d["algorithmic_bytes_per_launch"] - 16.0 * H * W) > 1 or d["kernel"] != kernel:
            return None, "profiles/pmc_latest.json is for another workload / kernel (%s)" % d["kernel"]
        return d["traffic_bytes_per_launch"], "profiles/pmc_latest.json (%s; %s)" % (d["kernel"], d["source"])
    except Exception:
        return None, "no PMC summary committed"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lp-size", dest="n", type=int, default=2000, help="variables = constraints of the dense LP (2000 = config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pivots", type=int, default=1200)
    ap.add_argument("--no-relaxations", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU over RCCL.  JSLP_BENCH_BACKEND=gloo (tests only) lets several ranks share the one GPU of the
    # test box so that the N > 1 code path of this file can be exercised there.
    backend = os.environ.get("JSLP_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    red_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from jslpsolver_amd import _capi, generators
    from jslpsolver_amd.engine import Tableau, pivot_digest

    lib = _capi.load_hip()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # ---- workload: config 3a ------------------------------------------------------------------------
    n = args.n
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    H, W = m.shape
    t = Tableau(m, vibr, vibc, device=device_index, lib=lib)
    t.save()  # device-resident copy of the initial tableau: every step restarts from it without touching PCIe

    def step():
        t.restore()
        return t.simplex(check_cycles=False)

    res = None
    for _ in range(args.warmup):
        res = step()
    import gc
    gc.collect()
    gc.disable()  # keep CPython's cyclic collector out of the timed region (re-enabled right after it)
    barrier()
    t0 = time.perf_counter()
    pivots = 0
    for _ in range(args.steps):
        res = step()
        pivots += res.pivots_phase1 + max(res.pivots_phase2, 0)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    total_pivots = sum_over_ranks(float(pivots))
    value = total_pivots / elapsed
    digest = pivot_digest(t.pivot_trace()[-(res.pivots_phase1 + max(res.pivots_phase2, 0)):])
    pivots_per_solve = res.pivots_phase1 + max(res.pivots_phase2, 0)

    # ---- roofline of the dominant kernel (k_update), measured live with HIP events on the engine's stream ---
    roofline = None
    if rank == 0:
        t.set_timing(True)
        step()
        upd_ms, launches, total_ms = t.get_timing()
        t.set_timing(False)
        bytes_per_launch = 16.0 * H * W  # read + write every fp64 cell of the H x W tableau
        avg_s = (upd_ms / 1e3) / max(launches, 1)
        achieved = bytes_per_launch / avg_s if launches else 0.0
        # the register-resident kernel runs ALL pivots of phase 2 in one launch; its unit of work stays one pivot
        # (16*H*W algorithmic bytes), so "launch" below means "pivot" for it
        kernel_name = {"resident": "k_simplex_resident", "fused": "k_pivot_fused", "select+update": "k_update",
                       "workgroup": "k_simplex_wg"}.get(t.last_path(), t.last_path())
        traffic, traffic_note = pmc_traffic(H, W, kernel_name)
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_note,
                    "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_s * 1e6, "launches": launches,
                    "whole_pivot_frac": (bytes_per_launch * value / max(world, 1)) / HBM_PEAK}
    t.close()

    # ---- the other config-3 instance (3b: generateRandomLP, every pivot is a phase-1 pivot; ends infeasible) ------------
    phase1 = None
    if rank == 0:
        m1, vibr1, vibc1, _op = generators.dense_random_lp_tableau(12345, n, n)
        t1 = Tableau(m1, vibr1, vibc1, device=device_index, lib=lib)
        t1.save()
        t1.simplex(check_cycles=False)
        t1.restore()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r1 = t1.simplex(check_cycles=False)
        dt = time.perf_counter() - t0
        phase1 = {"workload": "config 3b: generateRandomLP(seed 12345, %d x %d, density 1.0), phase 1 only" % (n, n),
                  "pivots": r1.pivots_phase1, "feasible": bool(r1.feasible), "pivots_per_s": r1.pivots_phase1 / dt,
                  "pivot_digest": pivot_digest(t1.pivot_trace()[-r1.pivots_phase1:]), "kernel": t1.last_path()}
        t1.close()

    # ---- LP relaxations/sec: Monster_II node batch sharded over ranks (config 4, throughput variant) -----
    relax = None
    if not args.no_relaxations:
        relax = relaxation_throughput(lib, device_index, rank, world, barrier, max_over_ranks, sum_over_ranks)

    if rank == 0:
        line = {
            "metric": "simplex pivots/sec", "value": value, "unit": "pivots/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 3a: generateResourceAllocation(seed 12345, %d vars x %d constraints, density 1.0), "
                                   "%dx%d fp64 tableau, %d pivots per solve, cycle check off; one replica per GPU"
                                   % (n, n, H, W, pivots_per_solve),
                       "pivot_digest": digest, "result_evaluation": t.evaluation, "parallelism": "replicas%d" % world},
            "roofline": roofline,
        }
        if relax is not None:
            line["relaxations"] = relax
        if phase1 is not None:
            line["phase1_instance"] = phase1
        if not args.no_cpu_baseline and world == 1:
            if relax is not None:
                relax["cpu_baseline"] = cpu_relaxation_baseline()
                if relax["cpu_baseline"] and relax["cpu_baseline"].get("value"):
                    relax["speedup_vs_cpu_1_thread"] = relax["value"] / relax["cpu_baseline"]["value"]
                    if relax["cpu_baseline"].get("aggregate_value"):
                        relax["speedup_vs_cpu_all_cores"] = relax["value"] / relax["cpu_baseline"]["aggregate_value"]
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_sample_pivots)
            if line["cpu_baseline"] and line["cpu_baseline"].get("value"):
                line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def relaxation_throughput(lib, device, rank, world, barrier, max_over_ranks, sum_over_ranks, reps=16):
    """Config 4 throughput variant: the 151 cut lists the reference visits on Monster_II,
    replicated `reps` times, evaluated as independent nodes; rank r takes nodes r, r+world, ... (no collective
    in the data path).  Needs the committed golden fixture for the cut lists only."""
    import gzip
    from jslpsolver_amd import Model
    from jslpsolver_amd.engine import Tableau
    path = os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz")
    if not os.path.exists(path):
        return None
    with gzip.open(path, "rt") as fh:
        g = json.load(fh)
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    # weak scaling: every rank evaluates reps x 151 nodes whatever the world size
    nodes = [c["cuts"] or [] for c in g["simplexCalls"][1:]] * (reps * world)
    mine = nodes[rank::world]
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision,
                row_capacity=m.shape[0] + 2 * len(model.integerVariables), device=device, lib=lib)
    t.applyCuts([], check_cycles=True)  # root relaxation
    t.save()
    # the cut lists are flattened once (host-side input preparation, like the tableau build); the timed call is the
    # engine entry point itself: restore + add cuts + simplex + RHS / row-map read-back for every node
    packed = t.pack_cut_lists(mine)
    per_call = []
    # warm-up: the first call allocates the slots and restores them in full; and the GPU has been idle while the host
    # built the model: ~10-15 ms into a new burst of work one call stalls for ~6 ms (clock ramp), so warm up past that
    for _ in range(25):
        t0 = time.perf_counter()
        t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
        per_call.append(time.perf_counter() - t0)
    calls = 10
    import gc
    gc.collect()
    gc.disable()  # a generation-2 pass of CPython's collector (~6 ms over the model's dicts) otherwise lands in a random call
    barrier()
    t0 = time.perf_counter()
    for _ in range(calls):
        t1 = time.perf_counter()
        results, rhs, rows = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
        per_call.append(time.perf_counter() - t1)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0) / calls
    gc.enable()
    total = sum_over_ranks(float(len(mine)))
    my_pivots = [results[i].pivots_phase1 + max(results[i].pivots_phase2, 0) for i in range(len(mine))]
    piv = sum_over_ranks(float(sum(my_pivots)))
    # algorithmic bytes of one relaxation: restore = 16*H*W, then p pivots of 16*H'*W with H' = H + #cuts
    H, W = m.shape
    my_bytes = sum(16.0 * H * W + p * 16.0 * results[i].height * W for i, p in enumerate(my_pivots))
    alg_bytes = sum_over_ranks(float(my_bytes))
    t.close()
    return {"value": total / el, "unit": "LP relaxations/s", "nodes": int(total), "pivots": int(piv), "seconds": el,
            "calls_averaged": calls, "per_call_us": [round(1e6 * x) for x in per_call[-calls:]],
            "roofline": {"bound": "latency (per-node kernel); hbm for reference", "achieved": alg_bytes / el / 1e9 / world,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s per GPU", "frac": alg_bytes / el / world / HBM_PEAK,
                         "note": "algorithmic bytes per relaxation = 16*H*W (restore) + pivots * 16*H'*W as SURVEY.md 8d defines "
                                 "them (dense update of every cell); the per-node kernel restores only dirty rows and updates "
                                 "only the rows/columns the reference's zero gate touches, so its real HBM traffic is far "
                                 "below this figure and the fraction can exceed 1"},
            "workload": "config 4: Monster_II (935x925 root, 112 ints), the reference's 151 visited cut lists x%d as one "
                        "batch of independent nodes per rank, sharded round-robin over %d rank(s) (weak scaling)" % (reps, world)}


def cpu_relaxation_baseline(seconds=4.0):
    """The reference itself on config 4 on this box's host cores: LP relaxations/s of solver.Solve(Monster_II) with one
    thread, and the aggregate of one such process per core (the generous CPU figure of SURVEY.md 8d)."""
    script = os.path.join(ROOT, "oracle", "ref_relax_rate.js")
    fixture = os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "src", "solver.js")):
        return None

    def run(n_proc):
        procs = [subprocess.Popen(["node", script, fixture, str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  stdin=subprocess.DEVNULL, text=True) for _ in range(n_proc)]
        rates = []
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=120)
                rates.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1])["relaxations_per_sec"])
            except Exception:
                pr.kill()
        return rates

    try:
        one = run(1)
        cores = os.cpu_count() or 1
        many = run(cores)
        return {"value": one[0] if one else None, "unit": "LP relaxations/s", "cores": 1, "kind": "reference",
                "aggregate_value": sum(many) if many else None, "aggregate_processes": len(many),
                "sample": "solver.Solve(Monster_II) repeated for ~%.0f s per process after one warm-up (151 B&B relaxations + root + "
                          "final per solve), clock = Tableau.solve() only; aggregate = one independent node process per host core "
                          "(%d), all running concurrently" % (seconds, cores)}
    except Exception as e:
        return {"value": None, "unit": "LP relaxations/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}


if __name__ == "__main__":
    main()
