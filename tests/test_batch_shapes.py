"""The node batch under every launch shape the engine has for it (tools/queue_check.py in a subprocess per setting: the JSLP_*
knobs are read once per process): outcome digests and the per-node work counters must not depend on the shape -- the queue
kernel with copy-on-write slots (default), with eager restores, without the transposed root, one launch per group of slots,
outcomes through the staging buffer + copies, most-cuts-first hand-out, and a batch cut into many small groups.  The digests
themselves are pinned to the reference by tests/test_gpu_parity.py (rhsSha of every Monster_II relaxation) and by bench.py."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [{}, {"JSLP_NODE_COW": "0"}, {"JSLP_SNAPSHOT_TRANSPOSE": "0"}, {"JSLP_NODE_COW": "0", "JSLP_SNAPSHOT_TRANSPOSE": "0"},
            {"JSLP_NODE_QUEUE": "0"}, {"JSLP_NODE_QUEUE": "1"}, {"JSLP_NODE_QUEUE": "2"}, {"JSLP_ZERO_COPY": "0"}, {"JSLP_GROUP_MAX": "100"},
            {"JSLP_NODE_QUEUE": "0", "JSLP_NO_WGLDS": "1"}]


def _run(extra):
    env = dict(os.environ, REPS="8", **extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "queue_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    shas = re.findall(r"rep (\d) outcome sha (\w+)", out.stdout)
    counters = eval(out.stdout.strip().splitlines()[-1])
    return dict(shas), counters


@pytest.mark.gpu
def test_node_batch_is_the_same_under_every_launch_shape(hip_lib):
    base_sha, base_cnt = _run({})
    assert base_sha["1"] == base_sha["2"]  # (rep 0 reports the engine's previous evaluation for nodes that end infeasible)
    assert base_cnt["relaxations"] == 8 * 151 and base_cnt["pivots"] == 8 * 842 and base_cnt["gated_rows"] == 8 * 8146
    for extra in SETTINGS[1:]:
        sha, cnt = _run(extra)
        assert sha["2"] == base_sha["2"], extra
        for k in ("relaxations", "simplex_calls", "pivots", "gated_cells", "gated_rows", "cut_rows", "height_sum"):
            assert cnt[k] == base_cnt[k], (extra, k)


@pytest.mark.gpu
@pytest.mark.parametrize("cow_small", ["1", "0"])
def test_small_dependent_batches_copy_on_write_or_eager_restore_give_the_verified_outcomes(cow_small):
    """round 6: a one-group batch of the 1024-thread node kernel (<= 16 nodes: the tree's speculative batches) starts copy-on-write by default
    (JSLP_NODE_COW_SMALL=0: eager restores, as until round 6); tools/node_latency.py checks every node of its 1 / 8 / 16-node batches against the verified
    full read-back -- repeated calls, so that a slot's rows dirtied by the previous call are what the next call starts from"""
    env = dict(os.environ, JSLP_NODE_COW_SMALL=cow_small)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "node_latency.py"), os.path.join("/tmp", "node_latency_cow_%s.md" % cow_small)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("== the verified full read-back") == 3, out.stdout[-2000:]
