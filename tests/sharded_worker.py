"""Worker for tests/test_sharded_gloo.py: launched by torch.distributed.run with WORLD_SIZE ranks (gloo, CPU).
Each rank owns an engine (the TEST-ONLY oracle library stands in for a GPU), nodes are sharded across ranks."""
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as G  # noqa: E402
from jslpsolver_amd import Solve, _capi  # noqa: E402
from jslpsolver_amd.engine import Tableau  # noqa: E402
from jslpsolver_amd.sharding import evaluate_nodes_sharded, evaluate_nodes_sharded_watched, watched_block_bytes  # noqa: E402


def main():
    backend = os.environ.get("JSLP_TEST_BACKEND", "gloo")
    if backend == "nccl":  # RCCL: one rank per GPU (the test box has one: world_size 1 still runs the real collectives)
        import torch
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    # CPU run: the test-only oracle stands in for the GPUs; GPU run ("virtual shards"): every rank drives the HIP engine
    # on the one visible MI355X, the exchange still goes over gloo
    if os.environ.get("JSLP_TEST_ENGINE") == "hip":
        lib = _capi.load_hip()
    else:
        lib = _capi.Library(os.path.join(ROOT, "oracle", "libjslp_oracle.so"))
    report_backend = lib.backend
    report = {"rank": rank, "world": world, "backend": report_backend, "cases": []}
    # 1. whole solves: sharded speculative B&B == the reference's result
    for name in ("Monster_II", "Knapsack_1", "Integer_Wood_Shop_Problem", "Sudoku4x4"):
        g = G.load(os.path.join(G.GOLDEN, "fixtures", name + ".json.gz"))
        ref = {k: (G.num(v) if not isinstance(v, bool) else v) for k, v in g["result"].items()}
        # (round 5: the tree's exchange is COMPACT by default -- the integer variables' rows and values; JSLP_SHARD_COMPACT=0: whole columns)
        for compact in ("1", "0"):
            os.environ["JSLP_SHARD_COMPACT"] = compact
            out = Solve(g["model"], full=True, lib=lib, speculate=4 * world, group=dist.group.WORLD)
            ok = out["result"] == ref and out["iter"] == g["final"]["branchAndCutIterations"]
            report["cases"].append({"name": name + (" (compact exchange)" if compact == "1" else " (full exchange)"), "ok": bool(ok)})
        os.environ.pop("JSLP_SHARD_COMPACT", None)
    # 2. the throughput unit: a batch of independent nodes sharded round-robin, outcomes all-gathered
    g = G.load(os.path.join(G.GOLDEN, "fixtures", "Monster_II.json.gz"))
    tab = g["tableau"]
    m, vibr, vibc = G.dense_tableau(tab)
    calls = g["simplexCalls"]
    t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"],
                row_capacity=tab["height"] + max(len(c["cuts"] or []) for c in calls), lib=lib)
    t.applyCuts([], check_cycles=True)
    t.save()
    nodes = [c["cuts"] or [] for c in calls[1:]]
    # both forms of the exchange step: host payload (gloo's default) and outcomes left in "device" memory until they are gathered
    # (RCCL's default, jslp_engine_relax_batch_device; over gloo with CPU tensors -- the oracle's device memory is host memory)
    for form in ("0", "1"):
        os.environ["JSLP_SHARD_DEVICE_PATH"] = form
        out = evaluate_nodes_sharded(t, nodes, True, dist.group.WORLD)
        ok = len(out) == len(nodes) and [int(h) for h in out.heights()] == [c["height"] for c in calls[1:]]
        ok = ok and type(out).__name__ == ("ShardedOutcomesDevice" if form == "1" else "ShardedOutcomes")
        for ev, call in zip([out.node(i) for i in range(len(out))], calls[1:]):
            ok = ok and bool(ev.res.feasible) == call["feasible"] and ev.res.height == call["height"]
            ok = ok and G.sha_rhs(ev.rhs, ev.vibr) == call["rhsSha"]
        report["cases"].append({"name": "Monster_II node batch, exchange form %s" % form, "ok": bool(ok)})
    os.environ.pop("JSLP_SHARD_DEVICE_PATH", None)
    # 2b. the COMPACT exchange (jslp_engine_relax_batch_watched_device): per node the state record + row / RHS cell of the watched (integer)
    #     variables; every node against the reference's golden through the full outcome it must agree with
    import numpy as np
    watched = np.array(tab["integerVarIndexes"], dtype=np.int32)
    t.set_watched_variables(watched)
    full = evaluate_nodes_sharded(t, nodes, True, dist.group.WORLD)
    comp = evaluate_nodes_sharded_watched(t, nodes, True, dist.group.WORLD)
    ok = len(comp) == len(nodes) and type(comp).__name__ == "ShardedOutcomesWatched"
    for i, call in enumerate(calls[1:]):
        fe, ce = full.node(i), comp.node(i)
        ok = ok and G.sha_rhs(fe.rhs, fe.vibr) == call["rhsSha"]  # (the full outcome IS the reference's)
        # (a node without an optimum keeps whatever evaluation its engine held before: compare the bound only where there is one)
        ok = ok and bool(ce.res.feasible) == call["feasible"] and ce.res.height == call["height"] and (not ce.res.optimal or ce.res.evaluation == fe.res.evaluation)
        ok = ok and bool(ce.res.optimal) == bool(fe.res.optimal)
        row_of = {int(v): r for r, v in enumerate(fe.vibr) if r > 0}
        want_rows = np.array([row_of.get(int(v), -1) for v in watched], dtype=np.int32)
        want_vals = np.array([fe.rhs[r] if r > 0 else 0.0 for r in want_rows])
        ok = ok and np.array_equal(ce.wrows, want_rows) and ce.wvals.tobytes() == want_vals.tobytes()
    per = max((len(nodes) + world - 1) // world, 1)
    payload = watched_block_bytes(per, len(watched), t.state_record_bytes())
    report["cases"].append({"name": "Monster_II node batch, compact exchange (%d B per rank, %d B per node)" % (payload, payload // per), "ok": bool(ok)})
    for n_take in (0, 1, world + 1):  # ragged batches through the compact form
        sub = nodes[:n_take]
        comp = evaluate_nodes_sharded_watched(t, sub, True, dist.group.WORLD)
        ok = len(comp) == len(sub)
        for i, call in enumerate(calls[1:1 + n_take]):
            ce = comp.node(i)
            ok = ok and bool(ce.res.feasible) == call["feasible"] and ce.res.height == call["height"] and len(ce.wrows) == len(watched)
        report["cases"].append({"name": "ragged compact batch of %d node(s) over %d rank(s)" % (n_take, world), "ok": bool(ok)})
    # 3. ragged batches: fewer nodes than ranks, a batch that does not divide by the world size, an empty batch (ranks without a
    #    node still take part in the exchange step with a padded, empty contribution)
    for n_take in (0, 1, max(world - 1, 1), world + 1):
        sub = nodes[:n_take]
        out = evaluate_nodes_sharded(t, sub, True, dist.group.WORLD)
        ok = len(out) == len(sub)
        for i, call in enumerate(calls[1:1 + n_take]):
            ev = out.node(i)
            ok = ok and bool(ev.res.feasible) == call["feasible"] and ev.res.height == call["height"] and G.sha_rhs(ev.rhs, ev.vibr) == call["rhsSha"]
        report["cases"].append({"name": "ragged batch of %d node(s) over %d rank(s)" % (n_take, world), "ok": bool(ok)})
    t.close()
    # 4. a MILP whose ROOT relaxation is infeasible (no tree, no batch, hence no exchange step at all: no rank may wait for one),
    #    and one that is integral at the root; both against the same host without sharding
    base = {"optimize": "profit", "opType": "max",
            "constraints": {"cap": {"max": 10}, "need": {"min": 12}},
            "variables": {"x": {"profit": 3, "cap": 1, "need": 1}, "y": {"profit": 2, "cap": 1, "need": 1}},
            "ints": {"x": 1, "y": 1}}
    feasible_root = dict(base, constraints={"cap": {"max": 10}, "need": {"min": 4}})
    for label, model in (("infeasible root", base), ("integral root", feasible_root)):
        sharded = Solve(model, full=True, lib=lib, speculate=2 * world, group=dist.group.WORLD)
        local = Solve(model, full=True, lib=lib)
        ok = sharded["result"] == local["result"] and sharded["iter"] == local["iter"]
        if label == "infeasible root":
            ok = ok and sharded["result"]["feasible"] is False
        report["cases"].append({"name": label, "ok": bool(ok)})
    gathered = [None] * world
    dist.all_gather_object(gathered, report)
    if rank == 0:
        print("REPORT " + json.dumps(gathered))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
