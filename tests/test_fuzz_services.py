"""Reference runs on random models: 630 on small MILPs (the reference's own generators, seeds 100..129, seven service policies each;
tests/golden/gen_golden_fuzz.js): the Python host + engine must take the same pivots in the same order, the same number
of relaxations and return the same result object.  CPU: oracle engine; `-m gpu`: the HIP engine."""
import gzip
import json
import os

import numpy as np
import pytest

import golden_util as G
from jslpsolver_amd import Solve, UnsupportedModel, pivot_digest

def _load(name):
    with gzip.open(os.path.join(G.GOLDEN, name), "rt") as fh:
        cases = [json.loads(line) for line in fh if line.startswith("{")]
    return [c for c in cases if c["fixed"] == 0 and not c["infeasPre"]]  # the rest needs the reference's presolve pre-pass


REPLAYABLE = _load("fuzz_services.jsonl.gz")
# random models with soft constraints (weight / priority -> optional objectives), equalities, ranges, unrestricted
# variables, a third of them with integers under three service policies (tests/golden/gen_golden_fuzz_soft.js)
SOFT = _load("fuzz_soft.jsonl.gz")


# exact counts: 630 service runs of which 7 need the reference's presolve pre-pass (skipped above), 433 soft-constraint
# runs of which 39 do; none of the rest may turn into a skip (UnsupportedModel) or the test fails
N_REPLAYABLE, N_SOFT = 623, 394


def replay(lib, cases):
    done = 0
    unsupported = []
    for c in cases:
        try:
            out = Solve(c["model"], full=True, lib=lib)
        except UnsupportedModel as e:
            unsupported.append((c["gen"], c["seed"], str(e)))
            continue
        where = (c["gen"], c["seed"], c["model"].get("options"))
        assert len(out["pivots"]) == c["nPivots"], where
        assert pivot_digest(out["pivots"]) == c["digest"], where
        assert c["iter"] is None or out["iter"] == c["iter"], where
        res = out["result"]
        assert list(res.keys()) == c["keys"], where
        for k, v in c["result"].items():
            ref = v if isinstance(v, bool) else G.num(v)
            assert res[k] == ref or (isinstance(ref, float) and np.isnan(ref) and np.isnan(res[k])), (where, k)
        done += 1
    assert unsupported == [], unsupported
    return done


def test_fuzz_through_oracle_engine(oracle_lib):
    assert len(REPLAYABLE) == N_REPLAYABLE
    assert replay(oracle_lib, REPLAYABLE) == N_REPLAYABLE


def test_soft_constraint_fuzz_through_oracle_engine(oracle_lib):
    assert len(SOFT) == N_SOFT
    assert replay(oracle_lib, SOFT) == N_SOFT


@pytest.mark.gpu
def test_fuzz_on_gpu(hip_lib):
    assert replay(hip_lib, REPLAYABLE) == N_REPLAYABLE


@pytest.mark.gpu
def test_soft_constraint_fuzz_on_gpu(hip_lib):
    assert replay(hip_lib, SOFT) == N_SOFT
