"""The restated synthetic generators must produce the reference's instances bit for bit."""
import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Model, generators


def _golden(name):
    import os
    return G.load(os.path.join(G.GOLDEN, "synthetic", name + ".json.gz"))


@pytest.mark.parametrize("n", [20, 100])
def test_models_equal_reference_models(n):
    g = _golden("generateResourceAllocation_%dx%d_seed12345" % (n, n))
    m = generators.resource_allocation_model(12345, n, n, density=1.0)
    ref = {k: v for k, v in g["model"].items() if k != "options"}
    assert m == ref
    g = _golden("generateRandomLP_%dx%d_seed12345" % (n, n))
    m = generators.random_lp_model(12345, n, n, density=1.0)
    ref = {k: v for k, v in g["model"].items() if k != "options"}
    assert m == ref


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sparse_models_equal_reference_models(seed):
    g = _golden("generateResourceAllocation_40x25_seed%d" % seed)
    assert generators.resource_allocation_model(seed, 40, 25, density=0.6) == g["model"]
    g = _golden("generateRandomLP_40x30_seed%d" % seed)
    assert generators.random_lp_model(seed, 40, 30, density=0.5) == g["model"]


@pytest.mark.parametrize("n", [20, 100, 200, 500, 1000, 2000])
def test_dense_tableaus_match_reference_sha(n):
    """2000 x 2000 is BASELINE.json config 3: the seed accumulator passes 2^53 there"""
    g = _golden("generateResourceAllocation_%dx%d_seed12345" % (n, n))
    m, vibr, vibc = generators.dense_resource_allocation_tableau(12345, n, n)
    assert G.sha_matrix(m) == g["tableau"]["matrixSha"]
    assert vibr.tolist() == [-1 if v is None else v for v in g["tableau"]["varIndexByRow"]]
    assert vibc.tolist() == [-1 if v is None else v for v in g["tableau"]["varIndexByCol"]]
    g = _golden("generateRandomLP_%dx%d_seed12345" % (n, n))
    m, vibr, vibc, op = generators.dense_random_lp_tableau(12345, n, n)
    assert G.sha_matrix(m) == g["tableau"]["matrixSha"]
    assert (op == "min") == g["tableau"]["isMinimization"]


def test_dense_fast_path_equals_model_path():
    m1, r1, c1 = generators.dense_resource_allocation_tableau(7, 30, 20)
    m2, r2, c2 = Model(generators.resource_allocation_model(7, 30, 20, density=1.0)).build_tableau()
    assert np.array_equal(m1.view(np.uint64), m2.view(np.uint64)) and r1.tolist() == r2.tolist() and c1.tolist() == c2.tolist()
    m1, r1, c1, op = generators.dense_random_lp_tableau(7, 30, 20)
    mod = generators.random_lp_model(7, 30, 20, density=1.0)
    m2, r2, c2 = Model(mod).build_tableau()
    assert op == mod["opType"]
    assert np.array_equal(m1.view(np.uint64), m2.view(np.uint64)) and r1.tolist() == r2.tolist() and c1.tolist() == c2.tolist()
