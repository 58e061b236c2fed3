"""Drop-in boundary through the reference's OWN host: oracle/_ref (the type-erased reference: JSON parsing,
presolve, branch-and-bound, result assembly) + host/gpu-tableau.js + addon/jslp_napi.node + an engine library.
Every fixture must give the very result object the unpatched reference gave, with the same pivot digest.
CPU run: the engine behind the addon is the test-only oracle library; GPU run: the HIP library."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "src", "solver.js")
ADDON = os.path.join(ROOT, "addon", "jslp_napi.node")


def _prepare():
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    if not os.path.exists(REF):
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(["python3", os.path.join(ROOT, "oracle", "build_ref.py")])
        else:
            pytest.skip("oracle/_ref is not built (needs the reference sources; built in the dev container)")
    if not os.path.exists(ADDON) or os.path.getmtime(ADDON) < os.path.getmtime(os.path.join(ROOT, "addon", "jslp_napi.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "addon")])


def _run(lib, sub, flt=""):
    out = subprocess.run(["node", os.path.join(ROOT, "host", "test", "dropin.js"), lib,
                          os.path.join(ROOT, "tests", "golden", sub), flt], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_addon_exports_the_abi():
    _prepare()
    out = subprocess.run(["node", "-e", "console.log(Object.keys(require(%r)).sort().join(','))" % ADDON],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == ("addCuts,applyMirCuts,checkpointCreate,checkpointRelease,checkpointRestore,create,destroy,detach,"
                                  "deviceCount,deviceMs,dims,download,getCounters,getOptionalObjectives,hostMatrix,load,pivot,pivotTrace,"
                                  "poolCreate,poolDestroy,poolRelaxBatch,poolRelaxBatchWatched,poolSetWatchedVariables,poolSize,poolSyncRoot,readRhs,relax,relaxBatch,relaxBatchWatched,relaxFrom,"
                                  "relaxWatched,releasePooledResources,restore,save,setCounting,setIntegerVariables,"
                                  "setOptionalObjectives,setWatchedVariables,simplex,timings,upload")


# exact counts (not lower bounds): a regression that turns cases into skips must fail
FIXTURE_COUNTS = {"fail": 0, "pass": 47, "solved_on_engine": 43, "strategy_variants_ok": 30, "incremental_ok": 114,
                  "device_checkpoints": 577, "mir_ok": 67, "speculative_ok": 15, "lookahead_ok": 15, "lookahead_ran": True, "size_policy_ok": 10, "defer_ok": 2, "fuzz_ok": 1063,
                  "edit_ok": 20, "released_ok": 1, "instance_ok": 16, "cycle_ok": 13, "pool_ok": 13, "pool_full_ok": 13, "watched_ok": 25, "pool_watched_ok": 25, "packed_ok": 25}


def _check_fixture_counts(r, backend):
    assert r["backend"] == backend
    assert {k: r[k] for k in FIXTURE_COUNTS} == FIXTURE_COUNTS


def test_reference_host_with_oracle_engine(oracle_lib):
    _prepare()
    # 47 reference fixtures; 30 enhanced-service variants; the incremental service over device checkpoints; MIR cuts;
    # 16-node speculative batches; minCells size policy; 1063 random models incl. the ones the reference's presolve touches;
    # the post-solve editing API (5 edits x 4 models); a released tableau refuses to solve; speculative batches over a pool
    _check_fixture_counts(_run(oracle_lib.path, "fixtures"), "oracle-c")
    r = _run(oracle_lib.path, "synthetic", "40x")
    assert r["fail"] == 0 and r["pass"] == 18


@pytest.mark.gpu
def test_reference_host_with_hip_engine(hip_lib):
    _prepare()
    _check_fixture_counts(_run(hip_lib.path, "fixtures"), "hip-gfx950")
    r = _run(hip_lib.path, "synthetic", "_")
    assert r["fail"] == 0 and r["pass"] == 40  # every synthetic golden that stores its model
