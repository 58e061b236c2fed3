"""Speculative batched branch-and-bound (in-order commit) must reproduce the sequential reference run."""
import pytest

import golden_util as G
from jslpsolver_amd import Model, Solve, UnsupportedModel
from test_host_solve import check_fixture


def _mips():
    out = []
    for p in G.fixture_paths() + G.synthetic_paths():
        g = G.load(p)
        if g.get("model") and g["tableau"] and g["tableau"]["integerVarIndexes"] and not g["tableau"]["optionalObjectives"]:
            out.append(p)
    return out


@pytest.mark.parametrize("path", _mips(), ids=G.ident)
@pytest.mark.parametrize("spec", [2, 8])
def test_speculation_gives_the_sequential_result(oracle_lib, path, spec):
    g = G.load(path)
    model = g["model"]
    if g["presolve"] and g["presolve"]["nFixed"] > 0:
        pytest.skip("reference presolve fixed variables")
    if model.get("timeout") or (model.get("options") or {}).get("timeout"):
        if "LargeFarm" in path:
            pytest.skip("wall-clock bounded run")
    try:
        Model(model)
    except UnsupportedModel as e:
        pytest.skip(str(e))
    seq = Solve(model, full=True, lib=oracle_lib)
    par = Solve(model, full=True, lib=oracle_lib, speculate=spec)
    assert par["result"] == seq["result"] and list(par["result"]) == list(seq["result"])
    assert par["iter"] == seq["iter"]
    assert par["matrix"].tobytes() == seq["matrix"].tobytes()
    ref = {k: (G.num(v) if not isinstance(v, bool) else v) for k, v in g["result"].items()}
    assert par["result"] == ref
