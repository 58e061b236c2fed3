"""Known-answer pins the reference's OWN unit tests hold for pieces of the path, restated against the
components that replace them here: same inputs, same expected values, file:line of the reference test next to each.

  src/tableau/min-heap.test.ts            -> jslpsolver_amd.branch_and_cut.BranchMinHeap
  src/tableau/mip-utils.test.ts           -> is_integral / most_fractional_var / fractional_volume
  src/tableau/cutting-strategies.test.ts  -> jslp_engine_add_cuts (oracle on CPU, HIP under -m gpu)
  src/tableau/backup.test.ts              -> jslp_engine_save / jslp_engine_restore
  src/tableau/solution.test.ts            -> the result rounding of solver.Solve
"""
import numpy as np
import pytest

from jslpsolver_amd import Tableau
from jslpsolver_amd.branch_and_cut import (BranchMinHeap, _rows_by_var, fractional_volume, is_integral, most_fractional_var)
from jslpsolver_amd.solver import _round_value
from jslpsolver_amd.branch_and_cut import js_round


# ---- min-heap.test.ts ---------------------------------------------------------------------------------------------
def test_heap_lifo_tie_breaking():
    """:198-211: equal relaxed evaluations leave most-recently-pushed first"""
    h = BranchMinHeap()
    for tag in ("first", "second", "third"):
        h.push(5, tag)
    assert [h.pop()[2] for _ in range(3)] == ["third", "second", "first"]


def test_heap_combines_min_and_lifo():
    """:213-234"""
    h = BranchMinHeap()
    for key, tag in ((10, "a"), (5, "b"), (5, "c"), (3, "d")):
        h.push(key, tag)
    assert [h.pop()[2] for _ in range(4)] == ["d", "c", "b", "a"]


def test_heap_property_after_interleaved_operations():
    """:237-254"""
    h = BranchMinHeap()
    h.push(50, None); h.push(30, None)
    assert h.pop()[0] == 30
    h.push(20, None); h.push(40, None)
    assert h.pop()[0] == 20
    h.push(10, None)
    assert h.pop()[0] == 10
    assert h.pop()[0] == 40 and h.pop()[0] == 50 and len(h) == 0


def test_heap_negative_and_fractional_keys():
    """:256-277"""
    h = BranchMinHeap()
    for k in (-10, -30, -20, 0):
        h.push(k, None)
    assert [h.pop()[0] for _ in range(4)] == [-30, -20, -10, 0]
    for k in (1.5, 1.1, 1.3):
        h.push(k, None)
    assert [h.pop()[0] for _ in range(3)] == [1.1, 1.3, 1.5]


# ---- mip-utils.test.ts ----------------------------------------------------------------------------------------------
class _M:
    """the slice of Model the helpers read"""

    def __init__(self, int_indexes):
        self.integerVariables = [{"index": i} for i in int_indexes]
        self.integer_index_set = frozenset(int_indexes)
        self.integer_index_array = np.array(int_indexes, dtype=np.int64)


def _state(values_by_var):
    """RHS column / varIndexByRow of a tableau whose row r holds variable r (the mock tableaus of the reference tests)"""
    n = max(values_by_var) if values_by_var else 0
    rhs = np.zeros(n + 1)
    vibr = np.full(n + 1, -1, dtype=np.int32)
    for v, x in values_by_var.items():
        rhs[v], vibr[v] = x, v
    return rhs, vibr


def test_is_integral_pins():
    rhs, vibr = _state({1: 5.0, 2: 3.0})
    assert is_integral(_M([1, 2]), rhs, _rows_by_var(vibr), 1e-9) is True           # :173-197
    rhs, vibr = _state({1: 5.0, 2: 3.5})
    assert is_integral(_M([1, 2]), rhs, _rows_by_var(vibr), 1e-9) is False          # :199-233
    rhs, vibr = _state({1: 5.5})
    assert is_integral(_M([]), rhs, _rows_by_var(vibr), 1e-9) is True               # :235-250 no integer variables
    rhs, vibr = _state({1: 5.5})
    assert is_integral(_M([2]), rhs, _rows_by_var(vibr), 1e-9) is True              # :252-276 integer variable not in the basis
    rhs, vibr = _state({1: 4.9999999})
    assert is_integral(_M([1]), rhs, _rows_by_var(vibr), 1e-6) is True              # :278-302 within precision


def test_fractional_volume_pins():
    """computeFractionalVolume: the services call it with ignoreIntegerValues = true (:383-411); the plain cases
    (:305-381, :413-428) agree with that mode whenever no integer variable sits at an integer value"""
    rhs, vibr = _state({1: 5.5})
    assert fractional_volume(_M([]), rhs, vibr, 1e-9) == 0                          # :305-319
    assert fractional_volume(_M([1]), rhs, vibr, 1e-9) == 5.5                       # :337-351
    rhs, vibr = _state({1: 2.5, 2: 3.5})
    assert fractional_volume(_M([1, 2]), rhs, vibr, 1e-9) == 2.5 * 3.5              # :353-381
    rhs, vibr = _state({1: 5.0, 2: 3.5})
    assert fractional_volume(_M([1, 2]), rhs, vibr, 1e-9) == 3.5                    # :383-411
    rhs, vibr = _state({1: -2.5})
    assert fractional_volume(_M([1]), rhs, vibr, 1e-9) == 2.5                       # :413-428


def test_most_fractional_var_pins():
    rhs, vibr = _state({1: 5.5})
    assert most_fractional_var(_M([]), rhs, _rows_by_var(vibr)) == (None, 0.0)      # :431-446
    rhs, vibr = _state({1: 5.3, 2: 3.7})
    assert most_fractional_var(_M([1, 2]), rhs, _rows_by_var(vibr)) == (1, 5.3)     # :448-485 (3.7's fraction is 0.2999..)
    rhs, vibr = _state({1: 5.1, 2: 3.5})
    assert most_fractional_var(_M([1, 2]), rhs, _rows_by_var(vibr)) == (2, 3.5)     # :487-522
    rhs, vibr = _state({1: 5.3})
    assert most_fractional_var(_M([1, 2]), rhs, _rows_by_var(vibr))[0] == 1         # :524-548 variable 2 not basic
    rhs, vibr = _state({1: 5.0})
    assert most_fractional_var(_M([1]), rhs, _rows_by_var(vibr)) == (None, 0.0)     # :550-567


# ---- cutting-strategies.test.ts ------------------------------------------------------------------------------------
def _mock_tableau(lib, rows, cols, matrix=None):
    """createMockTableau({width: 4, height: 2}) (:14-45): rows = varIndexByRow, cols = varIndexByCol"""
    m = np.zeros((2, 4)) if matrix is None else np.array(matrix, dtype=np.float64).reshape(2, 4)
    return Tableau(m, np.array(rows, dtype=np.int32), np.array(cols, dtype=np.int32), row_capacity=6, lib=lib)


def check_add_cut_pins(lib):
    # :60-70 one "max" cut on a non-basic variable: the height grows by one
    t = _mock_tableau(lib, [-1, 1], [-1, 2, 3, 4])
    t.addCutConstraints([{"type": "max", "varIndex": 2, "value": 10}])
    assert t.height == 3
    m = t.download()[0]
    assert m[2].tolist() == [10.0, 1.0, 0.0, 0.0]  # sign * value, sign at the variable's column (:46-53 of the source)
    t.close()
    # :72-86 variable in the basis: rhs = sign * (cut.value - varValue)
    t = _mock_tableau(lib, [-1, 1], [-1, 2, 3, 4], matrix=[0, 0, 0, 0, 3, 0, 0, 0])
    t.addCutConstraints([{"type": "min", "varIndex": 1, "value": 5}])
    assert t.height == 3 and t.download()[0][2, 0] == -1 * (5 - 3)
    t.close()
    # :88-99 variable not in the basis: rhs = sign * cut.value
    t = _mock_tableau(lib, [-1, 9], [-1, 1, 2, 3])
    t.addCutConstraints([{"type": "min", "varIndex": 1, "value": 5}])
    assert t.height == 3 and t.download()[0][2, 0] == -5
    t.close()
    # :101-117 several cuts at once, each with a fresh slack index (getNewElementIndex, tableau.ts:393-401)
    t = _mock_tableau(lib, [-1, 9], [-1, 1, 2, 3])
    t.addCutConstraints([{"type": "min", "varIndex": 1, "value": 5}, {"type": "max", "varIndex": 2, "value": 10},
                         {"type": "min", "varIndex": 3, "value": 2}])
    assert t.height == 5
    m, rows, _, _, _ = t.download()
    assert m[2:, 0].tolist() == [-5.0, 10.0, -2.0]
    assert rows[2:].tolist() == [4, 5, 6]  # lastElementIndex starts at width + height - 2 = 4
    t.close()


def test_add_cut_pins_oracle(oracle_lib):
    check_add_cut_pins(oracle_lib)


@pytest.mark.gpu
def test_add_cut_pins_hip(hip_lib):
    check_add_cut_pins(hip_lib)


# ---- backup.test.ts ----------------------------------------------------------------------------------------------------
def check_backup_pins(lib):
    rng = np.random.default_rng(3)
    m = rng.integers(1, 9, (4, 5)).astype(np.float64)
    t = Tableau(m, np.array([-1, 4, 5, 6], dtype=np.int32), np.array([-1, 0, 1, 2, 3], dtype=np.int32), row_capacity=7, lib=lib)
    # :190-198 restore() without a saved state does nothing
    t.restore()
    assert np.array_equal(t.download()[0], m)
    t.save()                                                       # :177-188
    t.pivot(2, 3)
    t.addCutConstraints([{"type": "max", "varIndex": 0, "value": 1}])
    assert t.height == 5 and not np.array_equal(t.download()[0][:4], m)
    t.restore()                                                    # :200-334: dimensions, matrix, the four maps come back
    back, rows, cols, rbv, cbv = t.download()
    assert t.height == 4 and np.array_equal(back, m)
    assert rows.tolist() == [-1, 4, 5, 6] and cols.tolist() == [-1, 0, 1, 2, 3]
    assert [int(rbv[v]) for v in (4, 5, 6)] == [1, 2, 3] and [int(cbv[v]) for v in (0, 1, 2, 3)] == [1, 2, 3, 4]
    # the saved state is a COPY (:164-175): changing the live tableau and restoring again gives the same answer
    t.pivot(1, 1)
    t.restore()
    assert np.array_equal(t.download()[0], m)
    # and the element-index counter is part of the state (:200-229): a cut after restore reuses the same slack index
    t.addCutConstraints([{"type": "max", "varIndex": 0, "value": 1}])
    first = int(t.download()[1][4])
    t.restore()
    t.addCutConstraints([{"type": "max", "varIndex": 0, "value": 1}])
    assert int(t.download()[1][4]) == first == 5 + 4 - 2
    t.close()


def test_backup_pins_oracle(oracle_lib):
    check_backup_pins(oracle_lib)


@pytest.mark.gpu
def test_backup_pins_hip(hip_lib):
    check_backup_pins(hip_lib)


# ---- solution.test.ts --------------------------------------------------------------------------------------------------
def test_solution_rounding_pins():
    def rounded(value, precision):
        return _round_value(value, js_round(1 / precision))
    assert rounded(5.0000001, 1e-6) == 5        # :161-184
    assert rounded(2.5, 1e-9) == 2.5            # :186-202
    assert rounded(0.0, 1e-9) == 0              # :204-219
    assert rounded(-7.5, 1e-9) == -7.5          # :221-236
    assert rounded(5.0, 1e-9) == 5 and rounded(3.0, 1e-9) == 3  # :86-111
