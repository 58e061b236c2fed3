"""The fp32 twin of the simplex core (jslp_engine_simplex_f32; SURVEY.md 8d config 5's fp32-vs-fp64 sweep).  It has no
reference to be exact against: these tests pin what it must guarantee -- the fp64 state is untouched, on small
well-conditioned models the fp32 run agrees with the fp64 run on the flags and, within fp32 accuracy, on the optimum --
and that the test library refuses it."""
import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Tableau
from jslpsolver_amd._capi import EngineError


def _tableau(lib, seed, H=14, W=11):
    rng = np.random.default_rng(seed)
    m = np.zeros((H, W))
    m[0, 1:] = rng.integers(1, 20, W - 1)
    m[1:, 1:] = rng.integers(1, 12, (H - 1, W - 1))
    m[1:, 0] = rng.integers(30, 90, H - 1)
    vibr = np.array([-1] + list(range(W - 1, W + H - 2)), dtype=np.int32)
    vibc = np.array([-1] + list(range(W - 1)), dtype=np.int32)
    return Tableau(m, vibr, vibc, precision=1e-8, row_capacity=H + 4, lib=lib)


def test_oracle_refuses_fp32(oracle_lib):
    t = _tableau(oracle_lib, 1)
    with pytest.raises(EngineError, match="simplex_f32"):
        t.simplex_f32(1e-6)
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_fp32_agrees_on_small_models_and_leaves_fp64_alone(hip_lib, seed):
    t = _tableau(hip_lib, seed)
    before = t.download()
    r32, rhs32, rows32, ms = t.simplex_f32(1e-5)
    after = t.download()
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    r64 = t.simplex()
    rhs64, rows64 = t.read_rhs()
    assert (r32.feasible, r32.bounded, r32.optimal) == (r64.feasible, r64.bounded, r64.optimal)
    assert ms >= 0
    assert abs(r32.obj_cell - r64.obj_cell) <= 1e-4 * max(1.0, abs(r64.obj_cell))
    if np.array_equal(rows32, rows64):  # same vertex: the values agree to fp32 accuracy
        assert np.allclose(rhs32, rhs64, rtol=1e-4, atol=1e-3)
    t.close()


@pytest.mark.gpu
def test_fp32_on_a_reference_fixture(hip_lib):
    g = G.load([p for p in G.fixture_paths() if "Monster_Problem" in p][0])
    m, vibr, vibc = G.dense_tableau(g["tableau"])
    t = Tableau(m, vibr, vibc, g["tableau"]["unrestricted"], precision=1e-8, lib=hip_lib)
    r32, _, _, _ = t.simplex_f32(1e-5, check_cycles=False)
    r64 = t.simplex(check_cycles=False)
    assert r64.feasible and r64.optimal
    assert r32.feasible and r32.optimal
    assert abs(r32.obj_cell - r64.obj_cell) <= 5e-3 * abs(r64.obj_cell)
    t.close()
