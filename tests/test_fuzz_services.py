"""630 reference runs on random small MILPs (the reference's own generators, seeds 100..129, seven service policies each;
tests/golden/gen_golden_fuzz.js): the Python host + engine must take the same pivots in the same order, the same number
of relaxations and return the same result object.  CPU: oracle engine; `-m gpu`: the HIP engine."""
import gzip
import json
import os

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Solve, UnsupportedModel, pivot_digest

with gzip.open(os.path.join(G.GOLDEN, "fuzz_services.jsonl.gz"), "rt") as fh:
    CASES = [json.loads(line) for line in fh]
REPLAYABLE = [c for c in CASES if c["fixed"] == 0 and not c["infeasPre"]]  # the rest needs the reference's presolve pre-pass


def replay(lib, cases):
    done = 0
    for c in cases:
        try:
            out = Solve(c["model"], full=True, lib=lib)
        except UnsupportedModel:
            continue
        where = (c["gen"], c["seed"], c["model"]["options"])
        assert len(out["pivots"]) == c["nPivots"], where
        assert pivot_digest(out["pivots"]) == c["digest"], where
        assert c["iter"] is None or out["iter"] == c["iter"], where
        res = out["result"]
        assert list(res.keys()) == c["keys"], where
        for k, v in c["result"].items():
            ref = v if isinstance(v, bool) else G.num(v)
            assert res[k] == ref or (isinstance(ref, float) and np.isnan(ref) and np.isnan(res[k])), (where, k)
        done += 1
    return done


def test_fuzz_through_oracle_engine(oracle_lib):
    assert len(REPLAYABLE) > 500
    assert replay(oracle_lib, REPLAYABLE) > 500


@pytest.mark.gpu
def test_fuzz_on_gpu(hip_lib):
    assert replay(hip_lib, REPLAYABLE) > 500
