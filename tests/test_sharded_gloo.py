"""N > 1 path on CPU: world_size 2 over gloo, one engine per rank (oracle library as the fake device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        assert all(c["ok"] for c in rep["cases"]), rep
