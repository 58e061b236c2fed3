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
    assert out.stdout.strip() == ("addCuts,applyMirCuts,checkpointCreate,checkpointRelease,checkpointRestore,create,destroy,"
                                  "deviceCount,dims,download,getOptionalObjectives,load,pivot,pivotTrace,readRhs,relax,"
                                  "relaxBatch,relaxFrom,releasePooledResources,restore,save,setIntegerVariables,setOptionalObjectives,simplex,upload")


def test_reference_host_with_oracle_engine(oracle_lib):
    _prepare()
    r = _run(oracle_lib.path, "fixtures")
    assert r["backend"] == "oracle-c" and r["fail"] == 0 and r["pass"] == 47 and r["solved_on_engine"] >= 43
    assert r["strategy_variants_ok"] == 30  # enhanced B&B services over the same seam
    # options.useIncremental: the reference's policy over device checkpoints (host/gpu-incremental-service.js)
    assert r["incremental_ok"] >= 100 and r["device_checkpoints"] > 300
    assert r["mir_ok"] >= 60  # options.useMIRCuts under the default, enhanced and incremental services
    assert r["speculative_ok"] >= 12  # install(..., {speculate: 16}): same results and relaxation counts as the sequential run
    assert r["size_policy_ok"] == 8  # install(..., {minCells}): small tableaus stay on the reference's own path
    assert r["fuzz_ok"] >= 1000  # random MILPs / soft-constraint models incl. the ones the reference's presolve touches
    r = _run(oracle_lib.path, "synthetic", "40x")
    assert r["fail"] == 0 and r["pass"] >= 12


@pytest.mark.gpu
def test_reference_host_with_hip_engine(hip_lib):
    _prepare()
    r = _run(hip_lib.path, "fixtures")
    assert r["backend"] == "hip-gfx950" and r["fail"] == 0 and r["pass"] == 47 and r["solved_on_engine"] >= 43
    assert r["strategy_variants_ok"] == 30
    assert r["incremental_ok"] >= 100 and r["device_checkpoints"] > 300
    assert r["mir_ok"] >= 60 and r["speculative_ok"] >= 12 and r["fuzz_ok"] >= 1000
    r = _run(hip_lib.path, "synthetic", "_")
    assert r["fail"] == 0 and r["pass"] >= 40
