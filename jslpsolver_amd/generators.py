"""Synthetic LP/MIP inputs for measurement: the reference's own seeded generators
(src/test-utils/problem-generator.ts) restated so the GPU box can build BASELINE.json's configs without the
reference.  Same Mulberry32 stream (:42-49), same draw order, so the instances -- and therefore the pivot
sequences -- are identical (tests pin the tableau sha256 and the pivot digest against the goldens).

The k-th Mulberry32 output depends only on k, so the stream is produced vectorised with numpy.  Detail that
matters at 2000 x 2000 (8M draws): the reference's `seed += 0x6d2b79f5` accumulates in a JavaScript double
and passes 2^53, where additions start to round; `np.add.accumulate` on float64 performs the same sequential
IEEE additions.
"""
import numpy as np

_INC = 0x6D2B79F5


def mulberry32_stream(seed, n):
    """first n outputs of createRng(seed) (problem-generator.ts:42-49) as float64 in [0, 1)"""
    a = np.full(n + 1, float(_INC), dtype=np.float64)
    a[0] = float(seed)
    acc = np.add.accumulate(a)[1:]                         # seed += 0x6d2b79f5 (double arithmetic)
    t = (acc.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)  # ToInt32 of the double
    with np.errstate(over="ignore"):
        t = (t ^ (t >> np.uint32(15))) * (t | np.uint32(1))            # Math.imul
        t = t ^ (t + (t ^ (t >> np.uint32(7))) * (t | np.uint32(61)))  # t ^= t + Math.imul(...)
        out = (t ^ (t >> np.uint32(14))).astype(np.float64) / 4294967296.0
    return out


def _rand_int(u, lo, hi):
    """Math.floor(lo + u * (hi - lo + 1))"""
    return np.floor(lo + u * (hi - lo + 1))


def resource_allocation_model(seed, num_variables=8, num_constraints=4, density=0.6, coefficient_range=(1, 50),
                              rhs_range=(100, 500)):
    """generateResourceAllocation (problem-generator.ts:297-340) as a JSON model dict"""
    u = mulberry32_stream(seed, num_variables * (1 + 2 * num_constraints) + num_constraints)
    k = 0
    variables, constraints = {}, {}
    for a in range(num_variables):
        v = {"profit": float(_rand_int(u[k], *coefficient_range))}
        k += 1
        for r in range(num_constraints):
            hit = u[k] < density
            k += 1
            if hit:
                v["resource%d" % r] = float(_rand_int(u[k], 1, 20))
                k += 1
        variables["activity%d" % a] = v
    for r in range(num_constraints):
        constraints["resource%d" % r] = {"max": float(_rand_int(u[k], *rhs_range))}
        k += 1
    return {"name": "ResourceAllocation_%dx%d_seed%d" % (num_variables, num_constraints, seed), "optimize": "profit",
            "opType": "max", "constraints": constraints, "variables": variables}


def random_lp_model(seed, num_variables=10, num_constraints=5, density=0.7, coefficient_range=(1, 100),
                    rhs_range=(10, 1000)):
    """generateRandomLP (problem-generator.ts:54-108) as a JSON model dict"""
    u = mulberry32_stream(seed, num_variables + num_constraints * (2 * num_variables + 2) + 1)
    k = 0
    variables, constraints = {}, {}
    for v in range(num_variables):
        variables["x%d" % v] = {"objective": float(_rand_int(u[k], *coefficient_range))}
        k += 1
    for c in range(num_constraints):
        for v in range(num_variables):
            hit = u[k] < density
            k += 1
            if hit:
                variables["x%d" % v]["c%d" % c] = float(_rand_int(u[k], *coefficient_range))
                k += 1
        rhs = float(_rand_int(u[k], *rhs_range))
        k += 1
        constraints["c%d" % c] = {"max": rhs} if u[k] < 0.5 else {"min": rhs}
        k += 1
    op = "max" if u[k] < 0.5 else "min"
    return {"name": "RandomLP_%dx%d_seed%d" % (num_variables, num_constraints, seed), "optimize": "objective",
            "opType": op, "constraints": constraints, "variables": variables}


def dense_resource_allocation_tableau(seed, n, m):
    """The tableau Tableau.setModel builds for generateResourceAllocation({seed, numVariables: n,
    numConstraints: m, density: 1.0}) -- BASELINE.json config 3a at n = m = 2000 -- without materialising the
    JSON model (4M coefficients).  Returns (matrix (m+1)x(n+1), varIndexByRow, varIndexByCol)."""
    u = mulberry32_stream(seed, n * (1 + 2 * m) + m)
    per = u[: n * (1 + 2 * m)].reshape(n, 1 + 2 * m)
    profit = _rand_int(per[:, 0], 1, 50)
    coef = _rand_int(per[:, 2::2], 1, 20)            # draws 2r+2 (draw 2r+1 is the always-true density test)
    rhs = _rand_int(u[n * (1 + 2 * m):], 100, 500)
    matrix = np.zeros((m + 1, n + 1), dtype=np.float64)
    matrix[0, 1:] = profit                             # max => cost row = +cost (tableau.ts:330-339)
    matrix[1:, 0] = rhs
    matrix[1:, 1:] = coef.T                            # "<=" rows copied as they are (:364-370)
    vibr = np.concatenate(([-1], np.arange(m))).astype(np.int32)       # constraints are created first
    vibc = np.concatenate(([-1], m + np.arange(n))).astype(np.int32)
    return matrix, vibr, vibc


def unrestricted_resource_allocation_tableau(seed, n, m, k):
    """generateResourceAllocation({seed, numVariables: n, numConstraints: m, density: 1.0}) whose first k activities are
    declared `unrestricted`, get their profit negated and a lower bound `lower<i>: {min: -(1 + i % 7)}` each -- the model
    tests/golden/gen_golden_wide.js hands to the reference.  Tableau (m + k + 1) x (n + 1); returns (matrix,
    varIndexByRow, varIndexByCol, unrestricted variable indexes)."""
    base, _, _ = dense_resource_allocation_tableau(seed, n, m)
    matrix = np.zeros((m + k + 1, n + 1), dtype=np.float64)
    matrix[:m + 1] = base
    matrix[0, 1:k + 1] = -base[0, 1:k + 1]          # negated profit, "max" => cost row = +cost
    for i in range(k):                               # "min" rows are negated, RHS included (tableau.ts:371-378)
        matrix[m + 1 + i, 1 + i] = -1.0
        matrix[m + 1 + i, 0] = float(1 + i % 7)
    vibr = np.concatenate(([-1], np.arange(m + k))).astype(np.int32)   # constraints are created first
    vibc = np.concatenate(([-1], m + k + np.arange(n))).astype(np.int32)
    return matrix, vibr, vibc, [m + k + i for i in range(k)]


def soft_resource_allocation_tableau(seed, n, m, k):
    """generateResourceAllocation({seed, numVariables: n, numConstraints: m, density: 1.0}) whose first k resources are SOFT
    (`priority` strong / medium / weak in turn, weight 1, limit halved) -- tests/golden/gen_golden_wide.js `softRA`.  Every relaxed
    constraint gets a relaxation variable created right behind it (expressions.ts:73-94), so the k relaxation columns come first
    (column 1 + i: -1 in row 1 + i, cost 0 on the main row; their costs live in the optional objective rows, tableau.ts:278-290).
    Tableau (m + 1) x (n + k + 1); returns (matrix, varIndexByRow, varIndexByCol, optional objective rows [3 x W], strong first)."""
    base, _, _ = dense_resource_allocation_tableau(seed, n, m)
    W = n + k + 1
    matrix = np.zeros((m + 1, W), dtype=np.float64)
    matrix[:, 0] = base[:, 0]
    matrix[1:k + 1, 0] = np.floor(base[1:k + 1, 0] / 2.0)
    matrix[:, k + 1:] = base[:, 1:]
    matrix[1 + np.arange(k), 1 + np.arange(k)] = -1.0
    # element indexes: constraint i and its relaxation variable alternate for i < k (2i, 2i + 1), then the other constraints, then the activities
    vibr = np.concatenate(([-1], 2 * np.arange(k), 2 * k + np.arange(m - k))).astype(np.int32)
    vibc = np.concatenate(([-1], 2 * np.arange(k) + 1, m + k + np.arange(n))).astype(np.int32)
    oo = np.zeros((3, W), dtype=np.float64)
    oo[np.arange(k) % 3, 1 + np.arange(k)] = -1.0
    return matrix, vibr, vibc, oo


def dense_random_lp_tableau(seed, n, m):
    """Same for generateRandomLP({seed, numVariables: n, numConstraints: m, density: 1.0}) -- config 3b.
    Returns (matrix, varIndexByRow, varIndexByCol, opType)."""
    u = mulberry32_stream(seed, n + m * (2 * n + 2) + 1)
    obj = _rand_int(u[:n], 1, 100)
    per = u[n: n + m * (2 * n + 2)].reshape(m, 2 * n + 2)
    coef = _rand_int(per[:, 1:2 * n:2], 1, 100)
    rhs = _rand_int(per[:, 2 * n], 10, 1000)
    is_max_row = per[:, 2 * n + 1] < 0.5
    op = "max" if u[n + m * (2 * n + 2)] < 0.5 else "min"
    matrix = np.zeros((m + 1, n + 1), dtype=np.float64)
    matrix[0, 1:] = obj if op == "max" else -obj
    sign = np.where(is_max_row, 1.0, -1.0)             # ">=" rows are negated, RHS included (:371-378)
    matrix[1:, 0] = sign * rhs
    matrix[1:, 1:] = sign[:, None] * coef
    vibr = np.concatenate(([-1], np.arange(m))).astype(np.int32)
    vibc = np.concatenate(([-1], m + np.arange(n))).astype(np.int32)
    return matrix, vibr, vibc, op
