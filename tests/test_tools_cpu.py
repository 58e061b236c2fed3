"""CPU checks of the measurement tooling (no GPU, no rocprofv3): the folding of SQ-counter passes into the PMC summary that bench.py
reports as `roofline.issue`, on a hand-made rocpd database with known sums."""
import json
import os
import sqlite3
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _fake_pass(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    db = sqlite3.connect(path)
    db.execute("create table counters_collection(kernel_name text, counter_name text, value real, grid_size int, workgroup_size int)")
    for name, values in rows.items():
        for v in values:
            db.execute("insert into counters_collection values(?,?,?,?,?)", ("void k_simplex_resident<1024, 2, 8>(ResCtx)", name, v, 257024, 1024))
        db.execute("insert into counters_collection values(?,?,?,?,?)", ("k_res_backup(f64::Slots)", name, 7.0, 64, 64))  # another kernel: ignored
    db.commit()
    db.close()


def test_sq_counters_fold_into_the_summary(tmp_path):
    import pmc_sq
    run = str(tmp_path)
    _fake_pass(os.path.join(run, "pmc_sq_pivots", "x", "a.db"),
               {"SQ_INSTS_VALU": [4.0e9, 4.2e9], "SQ_WAVES": [4016, 4016], "SQ_WAVE_CYCLES": [1.0e11, 1.0e11], "SQ_WAIT_ANY": [6.0e10, 6.0e10],
                "SQ_ACTIVE_INST_ANY": [2.5e10, 2.5e10], "SQ_INSTS_SALU": [3.0e9, 3.0e9]})
    with open(os.path.join(run, "pmc_sq_pivots.log"), "w") as fh:
        fh.write("noise\n" + json.dumps({"key": "pivots", "kernel": "k_simplex_resident", "dispatches": 2, "units": 2000, "unit": "pivot",
                                         "verified": {"pivots": 1000}, "workload": "fake"}) + "\n")
    latest = os.path.join(run, "latest.json")
    with open(latest, "w") as fh:
        json.dump({"pivots": {"kernel": "k_simplex_resident", "traffic_bytes_per_unit": 1.0}}, fh)
    pmc_sq.main(run, "pivots", "test", os.path.join(run, "out.md"), latest)
    doc = json.load(open(latest))
    assert set(doc) == {"pivots", "pivots_sq"}  # the HBM entry stays
    e = doc["pivots_sq"]
    import bench
    assert e["kernel_sources_sha"] == bench.kernel_sources_sha()
    assert e["waves_per_dispatch"] == 4016
    assert e["valu_per_wave_per_pivot"] == pytest.approx(8.2e9 / (4016 * 2000))
    assert e["useful_valu_frac"] == pytest.approx(32 / e["valu_per_wave_per_pivot"])
    assert e["wait_any_share_of_wave_cycles"] == pytest.approx(0.6)
    assert e["issuing_share_of_wave_cycles"] == pytest.approx(0.25)
    assert "valu_per_wave_per_pivot" in open(os.path.join(run, "out.md")).read()


def test_sq_fold_refuses_an_unverified_workload(tmp_path):
    import pmc_sq
    run = str(tmp_path)
    _fake_pass(os.path.join(run, "pmc_sq_relax", "x", "a.db"), {"SQ_WAVES": [6144]})
    with open(os.path.join(run, "pmc_sq_relax.log"), "w") as fh:
        fh.write(json.dumps({"key": "relax", "kernel": "k_node_queue", "dispatches": 1, "units": 2416, "unit": "LP relaxation", "verified": None,
                             "workload": "fake"}) + "\n")
    with pytest.raises(SystemExit):
        pmc_sq.main(run, "relax", "test", None, os.path.join(run, "latest.json"))


def test_bench_reports_the_committed_issue_budget_only_for_its_own_tree():
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    got = bench.pmc_sq("pivots", "k_simplex_resident")
    if d.get("pivots_sq", {}).get("kernel_sources_sha") == bench.kernel_sources_sha():
        assert got["valu_per_wave_per_pivot"] == d["pivots_sq"]["valu_per_wave_per_pivot"] and 0 < got["useful_valu_frac"] < 1
    else:
        assert set(got) == {"note"}
    assert set(bench.pmc_sq("pivots", "another_kernel")) == {"note"}
