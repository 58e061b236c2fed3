"""options.nodeSelection / options.branching (the reference's enhanced service, src/tableau/enhanced-branch-and-cut.ts)
through the Python host, against the reference run by `tests/golden/gen_golden_incremental.js enhanced`: same pivots in
the same order, same number of relaxations, same final tableau, same result object.

CPU: the oracle library behind the ABI.  `-m gpu`: libjslp_hip.so.  (Under the reference's own host the unchanged
service runs over the binding: host/test/dropin.js, strategy_variants_ok.)"""
import gzip
import json
import os

import pytest

import golden_util as G
from test_incremental_bnb import case_id, run_case

with gzip.open(os.path.join(G.GOLDEN, "enhanced.json.gz"), "rt") as fh:
    CASES = json.load(fh)


@pytest.mark.parametrize("case", [c for c in CASES if c["iterations"] > 1], ids=case_id)
def test_enhanced_service_through_oracle_engine(oracle_lib, case):
    run_case(oracle_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["iterations"] > 3 and c["nPivots"] < 20000], ids=case_id)
def test_enhanced_service_on_gpu(hip_lib, case):
    run_case(hip_lib, case)
