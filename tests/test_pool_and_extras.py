"""Round-2 additions of the C ABI: the device pool, the compact read-back, the work counters, the pinned
host-build buffer (8f.4) and the resident kernel's roll-back.  CPU: the test-only oracle library behind the same ABI;
-m gpu: the HIP library, with "virtual devices" (several pool members on the one visible MI355X)."""
import os

import numpy as np
import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from jslpsolver_amd import generators
from jslpsolver_amd.engine import DevicePool, Tableau, pivot_digest

MONSTER_II = os.path.join(G.GOLDEN, "fixtures", "Monster_II.json.gz")


def _root(lib, g, extra_rows=0):
    tab = g["tableau"]
    m, vibr, vibc = G.dense_tableau(tab)
    calls = g["simplexCalls"]
    t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"],
                row_capacity=tab["height"] + max(len(c["cuts"] or []) for c in calls) + extra_rows, lib=lib)
    res, rhs, rows = t.applyCuts([], check_cycles=True)
    assert G.sha_rhs(rhs, rows) == calls[0]["rhsSha"]
    t.save()
    return t, calls


def _check_pool(lib, n_members, reps):
    """a pool of n members reproduces the reference's outcome (sha256 of RHS column + row map) of every node Monster_II's
    branch-and-bound visits, `reps` times over, in both read-back flavours"""
    g = G.load(MONSTER_II)
    t, calls = _root(lib, g)
    nodes = [c["cuts"] or [] for c in calls[1:]] * reps
    want = [c for c in calls[1:]] * reps
    pool = DevicePool(t, [0] * n_members)
    assert pool.size == n_members
    for copy in (True, False):
        results, rhs, rows = pool.applyCutsBatch(nodes, check_cycles=True, copy=copy)
        assert len(nodes) == 151 * reps
        for i, call in enumerate(want):
            h = results[i].height
            assert bool(results[i].feasible) == call["feasible"] and h == call["height"]
            assert G.sha_rhs(rhs[i, :h], rows[i, :h]) == call["rhsSha"], (copy, i)
    # a new root (here: the same tableau saved again after one more relaxation) is fanned out again by itself
    first = nodes[0]
    t.applyCuts(first, check_cycles=True)
    t.save()
    results, rhs, rows = pool.applyCutsBatch([[], []] * n_members, check_cycles=True)
    ref_res, ref_rhs, ref_rows = t.applyCuts([], check_cycles=True)
    for i in range(2 * n_members):
        h = results[i].height
        assert h == ref_res.height and G.sha_rhs(rhs[i, :h], rows[i, :h]) == G.sha_rhs(ref_rhs, ref_rows)
    pool.close()
    t.close()


def test_pool_reproduces_monster_ii_nodes_oracle(oracle_lib):
    _check_pool(oracle_lib, 3, 1)


def _check_pool_watched(lib, n_members, reps):
    """round 4: the COMPACT read-back over the pool (jslp_pool_relax_batch_watched[_pinned]): per node the row and the RHS cell of
    the integer variables -- what mip-utils.ts:43-61, 100-126 read -- against the reference's own node outcomes: each node's full
    RHS column + row map (checked against the reference's sha256 above) yields the expected compact answer"""
    from jslpsolver_amd import Model
    g = G.load(MONSTER_II)
    t, calls = _root(lib, g)
    ints = [int(v) for v in Model(g["model"]).integer_index_array]
    nodes = [c["cuts"] or [] for c in calls[1:]] * reps
    want = [c for c in calls[1:]] * reps
    pool = DevicePool(t, [0] * n_members)
    with pytest.raises(Exception):  # the watched set must be given to the POOL (every member), not just to the primary
        t.set_watched_variables(ints)
        pool.n_watched = len(ints)
        pool.applyCutsBatchWatched(nodes[:4], check_cycles=True)
    pool.set_watched_variables(ints)
    full_res, rhs, rows = pool.applyCutsBatch(nodes, check_cycles=True, copy=True)
    ints_a = np.asarray(ints)
    for copy in (True, False):
        out, rows_w, vals_w = pool.applyCutsBatchWatched(nodes, check_cycles=True, copy=copy)
        for i, call in enumerate(want):
            h = out[i].height
            assert bool(out[i].feasible) == call["feasible"] and h == call["height"]
            a_, b_ = out[i].as_dict(), full_res[i].as_dict()
            if not call["feasible"]:  # (an infeasible node keeps the evaluation its ENGINE had before the call -- tableau.evaluation is
                a_.pop("evaluation"), b_.pop("evaluation")  # left alone, simplex.ts:73-76 -- and the members' histories differ between the two calls)
            assert a_ == b_
            assert G.sha_rhs(rhs[i, :h], rows[i, :h]) == call["rhsSha"]
            row_of = np.full(int(max(rows[i, :h].max(), ints_a.max())) + 1, -1, dtype=np.int64)
            row_of[rows[i, 1:h]] = np.arange(1, h)
            r = row_of[ints_a]
            assert np.array_equal(rows_w[i], r.astype(np.int32)), (copy, i)
            assert np.array_equal(vals_w[i].view(np.int64), np.where(r > 0, rhs[i, np.maximum(r, 0)], 0.0).view(np.int64)), (copy, i)
    pool.close()
    t.close()


def test_pool_compact_read_back_oracle(oracle_lib):
    _check_pool_watched(oracle_lib, 3, 1)


@pytest.mark.gpu
def test_pool_compact_read_back_of_four_virtual_devices(hip_lib):
    _check_pool_watched(hip_lib, 4, 3)  # 453 nodes over 4 engines on the one GPU, outcomes in ONE pinned [453 x 112] buffer


@pytest.mark.gpu
def test_pool_of_four_virtual_devices_reproduces_monster_ii_nodes(hip_lib):
    _check_pool(hip_lib, 4, 3)  # 453 nodes over 4 engines / streams / host threads on the one GPU


def _check_watched(lib):
    g = G.load(MONSTER_II)
    t, calls = _root(lib, g)
    from jslpsolver_amd import Model
    ints = [int(v) for v in Model(g["model"]).integer_index_array]  # model.integerVariables' indexes, as the host has them
    t.set_watched_variables(ints)
    for call in calls[1:40]:
        cuts = call["cuts"] or []
        res_w, rows_w, vals_w = t.applyCutsWatched(cuts, check_cycles=True)
        res, rhs, vibr = t.applyCuts(cuts, check_cycles=True)
        assert res_w.as_dict() == res.as_dict()
        row_of = {int(v): r for r, v in enumerate(vibr) if r > 0}
        for i, v in enumerate(ints):
            r = row_of.get(v, -1)
            assert rows_w[i] == r
            assert vals_w[i] == (rhs[r] if r > 0 else 0.0)
    t.close()


def _check_watched_batch(lib, reps, duplicate=False):
    """the compact read-back of a whole batch (jslp_engine_relax_batch_watched) against the full one, node by node; reps > 1 makes
    the batch larger than the resident workgroups of the HIP engine's queue kernel (several nodes per slot, copy-on-write)"""
    g = G.load(MONSTER_II)
    t, calls = _root(lib, g)
    from jslpsolver_amd import Model
    ints = [int(v) for v in Model(g["model"]).integer_index_array]
    if duplicate:  # a variable listed twice (the API allows it): no variable -> position table on the device, every entry looks its row up
        ints = ints + ints[:5] + [ints[0]]
    t.set_watched_variables(ints)
    nodes = [c["cuts"] or [] for c in calls[1:]] * reps
    packed = t.pack_cut_lists(nodes)
    for _ in range(2):  # (the first batch of an engine goes through the separate restore / cut / simplex launches)
        out_w, rows_w, vals_w = t.applyCutsBatchWatched(None, check_cycles=True, packed=packed)
        out_p, rows_p, vals_p = t.applyCutsBatchWatched(None, check_cycles=True, packed=packed, copy=False)
        assert np.array_equal(rows_p, rows_w) and np.array_equal(vals_p.view(np.int64), vals_w.view(np.int64))
        out, rhs, vibr = t.applyCutsBatch(None, check_cycles=True, packed=packed)
        ints_a = np.asarray(ints)
        for i in range(len(nodes)):
            assert (out_w[i].feasible, out_w[i].bounded, out_w[i].pivots_phase1, out_w[i].pivots_phase2, out_w[i].height) == \
                   (out[i].feasible, out[i].bounded, out[i].pivots_phase1, out[i].pivots_phase2, out[i].height)
            h = out[i].height
            row_of = np.full(int(max(vibr[i, :h].max(), ints_a.max())) + 1, -1, dtype=np.int64)
            row_of[vibr[i, 1:h]] = np.arange(1, h)
            r = row_of[ints_a]
            assert np.array_equal(rows_w[i], r.astype(np.int32)), i
            want = np.where(r > 0, rhs[i, np.maximum(r, 0)], 0.0)
            assert np.array_equal(vals_w[i].view(np.int64), want.view(np.int64)), i
    t.close()


def test_watched_batch_equals_full_batch_oracle(oracle_lib):
    _check_watched_batch(oracle_lib, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("reps", [1, 8])
def test_watched_batch_equals_full_batch_hip(hip_lib, reps):
    _check_watched_batch(hip_lib, reps)


def test_watched_batch_with_a_variable_listed_twice_oracle(oracle_lib):
    _check_watched_batch(oracle_lib, 1, duplicate=True)


@pytest.mark.gpu
def test_watched_batch_with_a_variable_listed_twice_hip(hip_lib):
    """1208 nodes > the resident workgroups of the queue kernel: nodes land on copy-on-write slots whose GLOBAL maps are stale,
    so the compact read-back must come from the LDS row map"""
    _check_watched_batch(hip_lib, 8, duplicate=True)


def test_watched_read_back_equals_full_read_back_oracle(oracle_lib):
    _check_watched(oracle_lib)


@pytest.mark.gpu
def test_watched_read_back_equals_full_read_back_hip(hip_lib):
    _check_watched(hip_lib)


def _counted_batch(lib, n_nodes=60):
    g = G.load(MONSTER_II)
    t, calls = _root(lib, g)
    nodes = [c["cuts"] or [] for c in calls[1:1 + n_nodes]]
    t.applyCutsBatch(nodes, check_cycles=True)  # the first batch after save() fills every tableau copy in full
    t.set_counting(True)
    t.applyCutsBatch(nodes, check_cycles=True)
    c = t.get_counters()
    t.set_counting(False)
    t.close()
    want_pivots = sum(c2["p1"] + max(c2["p2"], 0) for c2 in calls[1:1 + n_nodes])
    return c, want_pivots


def test_work_counters_oracle(oracle_lib):
    c, want_pivots = _counted_batch(oracle_lib)
    assert c["relaxations"] == 60 and c["simplex_calls"] == 60 and c["pivots"] == want_pivots > 0
    assert c["gated_cells"] >= c["gated_rows"] > 0 and c["cut_rows"] > 0
    assert c["restored_rows"] >= 60 * 935  # the reference restores every row of the root


@pytest.mark.gpu
def test_work_counters_hip_equal_the_oracles(hip_lib, oracle_lib):
    """the kernels' own count of the cells simplex.ts:376-387 touches equals the sequential restatement's; only the
    restore differs by design (dirty rows instead of the whole matrix)"""
    a, _ = _counted_batch(hip_lib)
    b, _ = _counted_batch(oracle_lib)
    for k in ("relaxations", "simplex_calls", "pivots", "gated_cells", "gated_rows", "cut_rows", "height_sum"):
        assert a[k] == b[k], (k, a, b)
    assert 0 < a["restored_rows"] < b["restored_rows"]


def _check_host_matrix(lib):
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 100, 100)
    t = Tableau(m, vibr, vibc, lib=lib)
    ref = t.simplex(check_cycles=True)
    ref_final = t.download()[0]
    buf = t.host_matrix()  # pinned build buffer: zero-filled, the host writes the tableau cell by cell
    assert buf.shape == m.shape and not buf.any()
    buf[:, :] = m
    t.upload(buf, vibr, vibc)
    res = t.simplex(check_cycles=True)
    assert res.as_dict() == ref.as_dict() and t.download()[0].tobytes() == ref_final.tobytes()
    assert pivot_digest(t.pivot_trace()[-20:]) == "b5dedd09"  # the reference's digest for N = 100
    assert not t.host_matrix().any()  # zero-filled again
    t.close()


def test_host_matrix_oracle(oracle_lib):
    _check_host_matrix(oracle_lib)


@pytest.mark.gpu
def test_host_matrix_hip(hip_lib):
    _check_host_matrix(hip_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("abort_at", [0, 7, 150])
def test_resident_abort_rolls_back_and_resolves(hip_hooks_lib, abort_at, monkeypatch):
    """a timed-out hand-off inside the register-resident kernel (forced: the last workgroup gives up at pivot `abort_at`)
    must not leave the engine with advanced index maps over the old matrix: the solve is rolled back and re-run through
    the streaming kernels, pivot for pivot the reference's"""
    monkeypatch.setenv("JSLP_TEST_RESIDENT_ABORT", str(abort_at))
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 200, 200)
    t = Tableau(m, vibr, vibc, lib=hip_hooks_lib)
    res = t.simplex(check_cycles=True)
    assert t.last_path() in ("fused", "select+update")
    assert res.feasible and res.optimal and res.pivots_phase2 == 242
    assert pivot_digest(t.pivot_trace()) == "27aaaa0b"
    monkeypatch.delenv("JSLP_TEST_RESIDENT_ABORT")
    t2 = Tableau(m, vibr, vibc, lib=hip_hooks_lib)
    t2.simplex(check_cycles=True)
    assert t2.last_path() == "resident"
    a, b = t.download(), t2.download()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    t.close()
    t2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,unrestricted", [("ra", 1000, False), ("lp", 500, False), ("lp", 1000, False), ("int", 1000, True)])
def test_resident_row_fetch_with_a_late_wave(hip_hooks_lib, kind, n, unrestricted, monkeypatch):
    """The register-resident kernels fetch the winning row speculatively, next to its flag.  Round 3 found the flag looked at by
    thread 0 only: a wave that ran ahead of thread 0's wave (a cold instruction cache is enough) could load the row before it was
    visible and have it accepted on thread 0's LATER look -- rare, timing-dependent wrong pivots.  Every wave now looks at the flag
    itself before its own loads.  JSLP_TEST_RESIDENT_LATE_WAVE0 makes wave 0 of every workgroup reach every fetch ~8 k cycles late:
    with the old protocol no solve survives that (2942 of 20755 pivots on a 3001 x 3001 LP).  Round 4: every run against the KNOWN
    answer of its instance -- the reference's own goldens; for the general build (3 unrestricted variables) the C restatement's
    (tests/golden/stress_expect.json) -- not against another HIP run.  Lean build (phase 2 / phase 1 pipelines) and the general build."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    from resident_stress import int_instance
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
        want = KA.expected_dense("ra", n, n)
    elif kind == "lp":
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
        want = KA.expected_dense("lp", n, n)
    else:
        m, vibr, vibc = int_instance(n, n, 12345)
        want = KA.expected_stress("intunr3", n + 1, n + 1, 12345)
    assert want is not None
    unr = [0, 1, 2] if unrestricted else []
    for late in ("1", "0"):
        monkeypatch.setenv("JSLP_TEST_RESIDENT_LATE_WAVE0", late)
        t = Tableau(m, vibr, vibc, unr, lib=hip_hooks_lib)
        res = t.simplex(check_cycles=False)
        assert t.last_path() == "resident" and t.get_counters()["resident_aborts"] == 0
        tr = t.pivot_trace()
        got = (len(tr), pivot_digest(tr), G.sha_matrix(t.download()[0]), bool(res.feasible))
        t.close()
        assert got == (want["pivots"], want["digest"], want["final_sha"], want["feasible"]), (late, got)


@pytest.mark.gpu
def test_wide_resident_geometry_with_a_late_wave(hip_hooks_lib, monkeypatch):
    """the same for a 512-thread geometry (1201 x 2101: <512,6,12>, phase 2 only), late wave 0 and not: the known answer of
    tests/golden/stress_expect.json both times"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    from resident_stress import int_instance
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    A, vibr, vibc = int_instance(1200, 2100, 12345)
    want = KA.expected_stress("int", 1201, 2101, 12345)
    for late in ("1", "0"):
        monkeypatch.setenv("JSLP_TEST_RESIDENT_LATE_WAVE0", late)
        t = Tableau(A, vibr, vibc, lib=hip_hooks_lib)
        res = t.simplex(check_cycles=False)
        assert t.last_path() == "resident" and t.get_counters()["resident_aborts"] == 0
        tr = t.pivot_trace()
        got = (res.pivots_phase2, pivot_digest(tr), G.sha_matrix(t.download()[0]))
        t.close()
        assert got == (want["pivots"], want["digest"], want["final_sha"]), (late, got)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("ra", 1000), ("lp", 1000), ("ra", 2000)])
def test_checksummed_row_hand_over_survives_a_flag_that_overtakes_its_row(hip_hooks_lib, kind, n, monkeypatch):
    """Litmus for the lean kernels' row hand-over (jslp_resident_pipe.hip.h, JSLP_PIPE_ROW_CHECKSUM): nothing orders a wave's flag word
    behind its 16-byte stores of the candidate row on their way through the fabric -- the reader loads both, recomputes the row's
    checksum and looks again until it matches the word.  JSLP_TEST_RESIDENT_LATE_WAVE0=2 produces the bad order on purpose: wave 0 of
    every publishing workgroup raises its flag word ~8 k cycles BEFORE it stores its columns of the row, at every pivot.  A reader that
    trusted the flag would take the row of two pivots ago; the solve must still be the reference's to the bit (phase 2 and phase 1
    pipelines), and the repeated looks must show up in jslp_work_counters.resident_fetch_retries (the test is not vacuous)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
    want = KA.expected_dense(kind, n, n)
    assert want is not None
    monkeypatch.setenv("JSLP_TEST_RESIDENT_LATE_WAVE0", "2")
    t = Tableau(m, vibr, vibc, lib=hip_hooks_lib)
    res = t.simplex(check_cycles=False)
    c = t.get_counters()
    assert t.last_path() == "resident" and c["resident_aborts"] == 0
    tr = t.pivot_trace()
    got = (len(tr), pivot_digest(tr), G.sha_matrix(t.download()[0]), bool(res.feasible))
    t.close()
    assert got == (want["pivots"], want["digest"], want["final_sha"], want["feasible"]), got
    assert c["resident_fetch_retries"] >= len(tr) // 2, c  # (every pivot's winner held its row back: its readers looked more than once)


def _litmus_instance(which):
    """(matrix, vibr, vibc, unrestricted, optional objectives, check_cycles, known answer) of one build that hands rows over checksummed"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    wide = os.path.join(G.GOLDEN, "wide")
    if which == "tall":        # <512,4,16>: two pairs per lane, the geometry that naturally repeats looks
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 2000, 4000)
        return m, vibr, vibc, [], None, False, KA.expected_dense("ra", 2000, 4000)
    if which == "check":       # CHK build of the headline geometry (the reference's default cycle check on)
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 1000, 1000)
        return m, vibr, vibc, [], None, True, KA.expected_dense("ra", 1000, 1000)
    if which == "soft":        # <512,4,16,OPT>: three optional objective rows in registers; publishes from inside the update pass
        g = G.load(os.path.join(wide, "soft_RA_2000x3000_k30.json.gz"))
        m, vibr, vibc, oo = generators.soft_resource_allocation_tableau(12345, g["meta"]["n"], g["meta"]["m"], g["meta"]["k"])
        return m, vibr, vibc, [], oo, False, {"pivots": g["nPivots"], "digest": g["pivotDigest"], "final_sha": g["final"]["matrixSha"], "feasible": g["final"]["feasible"]}
    if which == "unrestricted":  # <512,4,16,UNR>: per-lane masks of the unrestricted columns
        g = G.load(os.path.join(wide, "unrestricted_RA_2000x3950_k50.json.gz"))
        m, vibr, vibc, unr = generators.unrestricted_resource_allocation_tableau(12345, g["meta"]["n"], g["meta"]["m"], g["meta"]["k"])
        return m, vibr, vibc, unr, None, bool(g["tableau"]["checkForCycles"]), {"pivots": g["nPivots"], "digest": g["pivotDigest"], "final_sha": g["final"]["matrixSha"], "feasible": g["final"]["feasible"]}
    raise KeyError(which)


@pytest.mark.gpu
@pytest.mark.parametrize("hook", ["2", "4", "6"])
@pytest.mark.parametrize("which", ["tall", "check", "soft", "unrestricted"])
def test_checksummed_hand_over_litmus_on_every_build_that_uses_it(hip_hooks_lib, which, hook, monkeypatch):
    """the litmus above ran the 2-column headline build only.  The checksummed hand-over is also live in the tall geometry
    (`<512,4,16>`: two pairs per lane), its OPT build (publishes from inside the old update pass), its UNR build and the CHK builds.
    Hooks (test library only): 2 = wave 0's flag word overtakes its whole part of the row; 4 / 6 = a TORN row -- the even / odd lanes of
    wave 0 store, the word goes up, the other half follows ~8 k cycles later, so a reader's first looks see half of the wave's columns
    from this epoch and half from two epochs back, which nothing but the checksum can tell.  Every instance against the reference's own
    golden (simplex.ts:271-296, 394-412), repeats counted."""
    m, vibr, vibc, unr, oo, check, want = _litmus_instance(which)
    assert want is not None
    monkeypatch.setenv("JSLP_TEST_RESIDENT_LATE_WAVE0", hook)
    kw = {"optional_objectives": oo} if oo is not None else {}
    t = Tableau(m, vibr, vibc, unr, lib=hip_hooks_lib, **kw)
    res = t.simplex(check_cycles=check)
    c = t.get_counters()
    path = t.last_path()
    tr = t.pivot_trace()
    got = (len(tr), pivot_digest(tr), G.sha_matrix(t.download()[0]), bool(res.feasible))
    t.close()
    assert path == "resident" and c["resident_aborts"] == 0 and c["resident_handovers"] == 0, (path, c)
    assert got == (want["pivots"], want["digest"], want["final_sha"], want["feasible"]), got
    assert c["resident_fetch_retries"] > 0, c  # (the hook did hold rows back: the test is not vacuous)


@pytest.mark.gpu
@pytest.mark.parametrize("hook", ["4", "6"])
@pytest.mark.parametrize("kind,n", [("ra", 1000), ("lp", 1000), ("ra", 2000)])
def test_checksummed_hand_over_survives_a_torn_row(hip_hooks_lib, kind, n, hook, monkeypatch):
    """the TORN-row hook on the headline geometry (one pair per lane: half of wave 0's LANES are held back), phase 2 and phase 1 pipelines"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    monkeypatch.setenv("JSLP_FORCE_PATH", "resident")
    if kind == "ra":
        m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    else:
        m, vibr, vibc, _ = generators.dense_random_lp_tableau(12345, n, n)
    want = KA.expected_dense(kind, n, n)
    assert want is not None
    monkeypatch.setenv("JSLP_TEST_RESIDENT_LATE_WAVE0", hook)
    t = Tableau(m, vibr, vibc, lib=hip_hooks_lib)
    res = t.simplex(check_cycles=False)
    c = t.get_counters()
    assert t.last_path() == "resident" and c["resident_aborts"] == 0
    tr = t.pivot_trace()
    got = (len(tr), pivot_digest(tr), G.sha_matrix(t.download()[0]), bool(res.feasible))
    t.close()
    assert got == (want["pivots"], want["digest"], want["final_sha"], want["feasible"]), got
    assert c["resident_fetch_retries"] > 0, c


@pytest.mark.gpu
@pytest.mark.parametrize("n,after_us,check", [(2000, 9000, False), (2000, 20000, True), (1000, 2500, False)])
def test_host_requested_abort_rolls_back_on_the_shipped_library(hip_lib, n, after_us, check, monkeypatch):
    """the library users load has no test hooks, so its abort / rollback path never ran.  JSLP_INJECT_RESIDENT_ABORT_US makes the
    ENGINE (host side) raise the pinned host-abort word that many microseconds into the cooperative launch; the last workgroup of the lean
    kernel looks at the word every 1024 pivots, raises the device-wide abort flag and leaves, everybody else meets its silence in the next
    gather and leaves too (registers abandoned, no epilogue); the host restores slot 0 from the safety-net copy and solves through the
    streaming kernels: the reference's pivots and final tableau, and the abort COUNTED.  (A kernel that finished before the word went up
    simply reports no abort: the sizes below leave a wide margin -- 9726 pivots take ~57 ms, 2833 take ~17 ms.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import known_answers as KA
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    want = KA.expected_dense("ra", n, n)
    monkeypatch.setenv("JSLP_INJECT_RESIDENT_ABORT_US", str(after_us))
    t = Tableau(m, vibr, vibc, lib=hip_lib)
    res = t.simplex(check_cycles=check)
    c = t.get_counters()
    path = t.last_path()
    tr = t.pivot_trace()
    got = (len(tr), pivot_digest(tr), G.sha_matrix(t.download()[0]), bool(res.feasible))
    assert got == (want["pivots"], want["digest"], want["final_sha"], want["feasible"]), got
    assert c["resident_launches"] == 1 and c["resident_aborts"] == 1 and path in ("fused", "select+update"), (path, c)
    # ... and the same engine, knob off, takes the register-resident path again and gives the same answer
    monkeypatch.delenv("JSLP_INJECT_RESIDENT_ABORT_US")
    t2 = Tableau(m, vibr, vibc, lib=hip_lib)
    t2.simplex(check_cycles=check)
    assert t2.last_path() == "resident" and t2.get_counters()["resident_aborts"] == 0
    assert G.sha_matrix(t2.download()[0]) == want["final_sha"]
    t.close()
    t2.close()


# (round 4: the repeated-solve stress of the tall / wide shapes moved to tests/test_resident_pins.py, where every run is compared with the
#  instance's KNOWN answer -- not with the first run -- and a rolled-back resident launch fails the test)


@pytest.mark.gpu
def test_pool_creation_fails_loudly_when_peer_access_is_refused(hip_lib):
    """jslp_pool_create on a platform that refuses peer access (injected: JSLP_TEST_PEER_REFUSED=1 makes every member look
    refused): a clean error that names the devices, nothing leaked, and the primary engine is as usable as before"""
    code = r"""
import os, sys
os.environ["JSLP_TEST_PEER_REFUSED"] = "1"
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import golden_util as G
from jslpsolver_amd import _capi
from jslpsolver_amd.engine import Tableau, DevicePool
lib = _capi.load_hip()
g = G.load(os.path.join(G.GOLDEN, "fixtures", "Monster_II.json.gz"))
tab = g["tableau"]; m, vibr, vibc = G.dense_tableau(tab)
t = Tableau(m, vibr, vibc, tab["unrestricted"], precision=tab["precision"], row_capacity=tab["height"] + 40, lib=lib)
t.applyCuts([], check_cycles=True); t.save()
for _ in range(3):
    try:
        DevicePool(t, [0, 0, 0])
        print("NO ERROR"); sys.exit(1)
    except Exception as e:
        assert "peer access refused" in str(e), str(e)
r1, _rhs, _rows = t.applyCuts(g["simplexCalls"][1]["cuts"] or [], check_cycles=True)
assert r1.height == g["simplexCalls"][1]["height"] and bool(r1.feasible) == g["simplexCalls"][1]["feasible"]
os.environ["JSLP_POOL_ALLOW_STAGED"] = "1"  # (read once per process: still refused here)
t.close(); print("OK")
""" % (ROOT, ROOT)
    import subprocess, sys
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_pivot_trace_overflow_is_an_error(hip_lib):
    """the trace holds 2^20 pivots since upload(); past that the pairs are refused instead of handed out truncated"""
    from jslpsolver_amd._capi import EngineError
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, 500, 500)
    t = Tableau(m, vibr, vibc, lib=hip_lib)
    t.save()
    solves = (1 << 20) // 657 + 1
    for _ in range(solves):
        t.restore()
        res = t.simplex(check_cycles=False)
    assert res.pivots_phase2 == 657
    with pytest.raises(EngineError, match="pivot_trace"):
        t.pivot_trace()
    t.upload(m, vibr, vibc)  # a new hand-over starts a new trace
    t.simplex(check_cycles=False)
    assert pivot_digest(t.pivot_trace()) == "1cda2607"
    t.close()


@pytest.mark.gpu
def test_large_batch_repeated(hip_lib):
    """bench.py's relaxation workload as a test: 2416 independent nodes (three groups of tableau copies), a dozen calls in a
    row on the same engine, every node's outcome checked against the reference's on the first and on the last call"""
    g = G.load(MONSTER_II)
    t, calls = _root(hip_lib, g, extra_rows=2 * 112)
    nodes = [c["cuts"] or [] for c in calls[1:]] * 16
    packed = t.pack_cut_lists(nodes)
    for k in range(12):
        results, rhs, rows = t.applyCutsBatch(None, check_cycles=True, packed=packed, copy=False)
        if k in (0, 11):
            for i in range(len(nodes)):
                call = calls[1 + i % 151]
                h = results[i].height
                assert h == call["height"] and bool(results[i].feasible) == call["feasible"]
                assert G.sha_rhs(rhs[i, :h], rows[i, :h]) == call["rhsSha"], (k, i)
    t.close()
