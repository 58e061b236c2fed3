"""N > 1 path on CPU: world_size 2 over gloo, one engine per rank (oracle library as the fake device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


def _run(engine, nproc, port, backend="gloo"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", JSLP_TEST_ENGINE=engine, JSLP_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT ")][-1]
    return json.loads(line[len("REPORT "):])


@pytest.mark.gpu
def test_virtual_shards_on_one_gpu(hip_lib):
    """4 ranks, each with its own HIP engine on the one visible GPU ("N virtual shards"), nodes sharded round-robin,
    outcomes all-gathered over gloo: every rank reproduces the reference results"""
    reports = _run("hip", 4, 29547)
    assert len(reports) == 4
    for rep in reports:
        assert rep["backend"] == "hip-gfx950" and rep["world"] == 4
        assert not [c["name"] for c in rep["cases"] if not c["ok"]], [c["name"] for c in rep["cases"] if not c["ok"]]


@pytest.mark.gpu
def test_rccl_exchange_on_one_gpu(hip_lib):
    """the exchange step over RCCL itself (backend "nccl"): the box has one GPU, so one rank -- the process group, the
    device-side all-gather of the outcome payload and the tree replay are the code the 8-GPU run executes"""
    reports = _run("hip", 1, 29551, backend="nccl")
    assert len(reports) == 1 and reports[0]["backend"] == "hip-gfx950" and reports[0]["world"] == 1
    assert not [c["name"] for c in reports[0]["cases"] if not c["ok"]], [c["name"] for c in reports[0]["cases"] if not c["ok"]]


def test_two_ranks_shard_nodes_and_agree(oracle_lib):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "sharded_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT ")][-1]
    reports = json.loads(line[len("REPORT "):])
    assert len(reports) == 2
    for rep in reports:
        assert rep["world"] == 2
        assert not [c["name"] for c in rep["cases"] if not c["ok"]], [c["name"] for c in rep["cases"] if not c["ok"]]


def test_four_ranks_ragged_batches_and_infeasible_root(oracle_lib):
    """world_size 4 on CPU (gloo, oracle engines): batches with fewer nodes than ranks / not divisible by the world size / empty,
    a MILP whose root relaxation is infeasible and one that is integral at the root (no exchange step must be waited for)"""
    reports = _run("oracle", 4, 29561)
    assert len(reports) == 4
    for rep in reports:
        assert rep["world"] == 4 and rep["backend"] == "oracle-c"
        names = [c["name"] for c in rep["cases"]]
        assert "infeasible root" in names and "ragged batch of 0 node(s) over 4 rank(s)" in names and "ragged batch of 5 node(s) over 4 rank(s)" in names
        assert not [c["name"] for c in rep["cases"] if not c["ok"]], [c["name"] for c in rep["cases"] if not c["ok"]]


@pytest.mark.gpu
def test_bench_multi_rank_code_path_on_one_gpu(hip_lib):
    """bench.py's N > 1 branch (process group, barrier, max / sum over ranks, replica + sharded-relaxation accounting)
    with 2 ranks sharing the one GPU of the test box (gloo instead of RCCL): the JSON line must be well formed and the
    aggregate must count both ranks' pivots"""
    # `python bench.py --gpus 2` as the driver types it for N = 1: bench.py spawns the two ranks itself
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", JSLP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--lp-size", "500"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["unit"] == "pivots/s" and line["value"] > 0
    assert line["config"]["pivot_digest"] == "1cda2607"  # the reference's digest for n = 500
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 2 * 657) < 1e-6 * 2 * 657
    assert line["relaxations"]["weak_scaling"]["nodes"] == 2 * 16 * 151 and line["relaxations"]["weak_scaling"]["scaling"] == "weak"
    # round 6: the STRONG-scaling leg -- one fixed batch split over the ranks, the exchange inside the timed region
    sb = line["relaxations"]["sharded_batch"]
    assert sb["scaling"] == "strong" and sb["nodes"] == 16 * 151 and sb["ranks"] == 2 and sb["value"] > 0, sb
    assert sb["exchange_ms"] > 0 and sb["bytes_per_rank"] >= (16 * 151 // 2) * (128 + 12 * 112), sb  # (state record + 12 B per integer variable per node of the rank's share)
    assert abs(sb["value"] * sb["seconds"] - sb["nodes"]) < 1e-6 * sb["nodes"] and 1e-3 * sb["exchange_ms"] < sb["seconds"], sb  # (the rate is over the median call's time, which contains the exchange)
    assert sb["value_over_mean_call"] > 0 and sb["max_call_us"] >= 1e6 * sb["seconds"] * 0.999 and "health" in sb, sb
    # ... and at N > 1 it IS the relaxation figure (the weak leg moved into `weak_scaling`)
    assert line["relaxations"]["value"] == sb["value"] and line["relaxations"]["scaling"] == "strong" and line["relaxations"]["nodes"] == sb["nodes"], line["relaxations"]["value_leg"]
    assert line["relaxations"]["tree"]["scaling"] == "strong" and line["relaxations"]["tree"]["result"] == 20631
    assert line["cycle_check_on"]["pivot_digest"] == "1cda2607"
    assert "cpu_baseline" not in line


@pytest.mark.gpu
def test_bench_pool_devices_leg_runs_as_a_child_process():
    """bench.py runs the real-devices pool leg in a child process under a time limit (never run on several GPUs by the builder): the child's entry
    point, here over the one visible device (a pool of one member), prints one JSON object with every node checked against the single engine"""
    import json, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pool-devices-leg", "0", "1", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    leg = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert leg["value"] > 0 and leg["members"] == 1 and leg["devices"] == [0] and len(leg["per_call_us"]) == 10
