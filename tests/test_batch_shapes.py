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


@pytest.mark.gpu
def test_mixed_batch_shapes_back_to_back_on_one_engine_are_the_reference(hip_lib):
    """one engine, one saved Monster_II root, and calls of every shape one after the other -- single nodes, one-group batches of 2 .. 16 nodes (copy-on-write start
    since round 6), a 40-node and a 300-node batch, the 2416-node queue batch, compact and full read-back mixed -- so that what one call leaves dirty in its slots
    is what the next one starts from: every node's outcome must be the reference's (height, feasibility, sha256 of RHS column + row map; the compact read-back
    against the same full outcome)"""
    import gzip
    import hashlib
    import json

    import numpy as np

    from jslpsolver_amd import Model
    from jslpsolver_amd.engine import Tableau
    with gzip.open(os.path.join(ROOT, "tests", "golden", "fixtures", "Monster_II.json.gz"), "rt") as fh:
        g = json.load(fh)
    model = Model(g["model"])
    m, vibr, vibc = model.build_tableau()
    calls = g["simplexCalls"][1:]
    t = Tableau(m, vibr, vibc, model.unrestricted, precision=model.precision, row_capacity=m.shape[0] + 2 * len(model.integerVariables), lib=hip_lib)
    try:
        t.applyCuts([], check_cycles=True)
        t.save()
        ints = np.asarray([int(v) for v in model.integer_index_array])
        t.set_watched_variables([int(v) for v in ints])

        def check_full(which, res, rhs, rows):
            for i, k in enumerate(which):
                call, h = calls[k % len(calls)], res[i].height
                sha = hashlib.sha256(np.ascontiguousarray(rhs[i, :h]).tobytes() + np.ascontiguousarray(rows[i, :h]).tobytes()).hexdigest()
                assert h == call["height"] and bool(res[i].feasible) == call["feasible"] and sha == call["rhsSha"], ("full", len(which), k)

        def check_compact(which, res, wrows, wvals):
            # against a full read-back of the same nodes taken right now (itself checked against the reference)
            full = t.applyCutsBatch([calls[k % len(calls)]["cuts"] or [] for k in which], check_cycles=True)
            check_full(which, *full)
            fres, frhs, frows = full
            for i in range(len(which)):
                h = fres[i].height
                row_of = np.full(int(max(frows[i, :h].max(), ints.max())) + 1, -1, dtype=np.int64)
                row_of[frows[i, 1:h]] = np.arange(1, h)
                r = row_of[ints]
                want = np.where(r > 0, frhs[i, np.maximum(r, 0)], 0.0)
                assert res[i].height == h and bool(res[i].feasible) == bool(fres[i].feasible), ("compact", len(which), i)
                assert np.array_equal(np.asarray(wrows[i]), r.astype(np.int32)) and np.array_equal(np.asarray(wvals[i]).view(np.int64), want.view(np.int64)), ("compact", len(which), i)

        start = 0
        for n, compact in ((1, True), (8, True), (16, False), (2, True), (1, False), (40, True), (16, True), (300, False), (3, True), (151 * 16, True), (8, False),
                           (1, True), (16, True), (151 * 16, False), (5, True)):
            which = [(start + i) % len(calls) for i in range(n)]
            start += 7
            nodes = [calls[k]["cuts"] or [] for k in which]
            for rep in range(2):  # (twice: the second call starts from what the first left in the slots)
                if compact:
                    res, wrows, wvals = t.applyCutsBatchWatched(nodes, check_cycles=True)
                    check_compact(which, res, wrows, wvals)
                else:
                    check_full(which, *t.applyCutsBatch(nodes, check_cycles=True))
    finally:
        t.close()
