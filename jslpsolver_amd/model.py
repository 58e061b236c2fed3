"""Host-side model layer: JSON model -> dense simplex tableau.

Mirrors, for the hot path only, what the reference does between `Solve(json)` and the first pivot:
`Model.loadJson` (src/model.ts:278-419: constraint/variable/row/column ORDER and option parsing) and
`Tableau._resetMatrix` (src/tableau/tableau.ts:319-380: sign conventions).  The order matters for parity:
it fixes which row/column every first-index tie-break of the pivot rules sees.

Soft constraints (`weight` / `priority`) are modelled as the reference does (src/expressions.ts:73-94,186-202):
one relaxation variable per relaxed bound, whose cost lands in a priority-ordered optional objective row
(tableau.ts:278-290, 335-338).  Out of scope here (raises UnsupportedModel): multi-objective `optimize` objects
(polyopt), `external` solvers, MIR cuts and the presolve pre-pass.
"""
import math

import numpy as np


class UnsupportedModel(ValueError):
    pass


def js_keys(obj):
    """Object.keys order: canonical array-index keys ascending first, then insertion order."""
    ints, strs = [], []
    for k in obj.keys():
        ks = str(k)
        if ks.isdigit() and (ks == "0" or ks[0] != "0") and int(ks) < 4294967295:
            ints.append((int(ks), k))
        else:
            strs.append(k)
    return [k for _, k in sorted(ints)] + strs


def _truthy(x):
    return bool(x) and not (isinstance(x, float) and math.isnan(x))


class Model:
    """The subset of src/model.ts `Model` state the hot path reads."""

    def __init__(self, json_model, precision=None):
        self.precision = 1e-8 if precision is None else float(precision)  # tableau.ts:96
        self.json = json_model
        opt = json_model.get("optimize")
        if isinstance(opt, dict):
            if len(opt) > 1:
                raise UnsupportedModel("multi-objective models go through Polyopt (out of scope)")
        if json_model.get("external"):
            raise UnsupportedModel("external solver delegation is out of scope")
        self.isMinimization = json_model.get("opType") != "max"  # model.ts:279
        self._next_index = 0
        self.relaxationIndex = 1  # model.ts:69
        self.constraints = []  # dicts: index, isUpperBound, rhs, terms [(var_pos, coefficient)]
        self.variables = []    # dicts: id, cost, index, isInteger, priority
        self.integerVariables = []
        self.unrestricted = []
        cons_min, cons_max = {}, {}
        constraints = json_model.get("constraints") or {}
        for cid in js_keys(constraints):  # model.ts:291-332
            c = constraints[cid]
            if not isinstance(c, dict):
                continue
            weight, priority = c.get("weight"), c.get("priority")
            relaxed = weight is not None or priority is not None  # model.ts:301
            if c.get("equal") is None:
                if c.get("min") is not None:
                    cons_min[cid] = self._add_constraint(c["min"], False)
                    if relaxed:
                        self._relax([cons_min[cid]], weight, priority)
                if c.get("max") is not None:
                    cons_max[cid] = self._add_constraint(c["max"], True)
                    if relaxed:
                        self._relax([cons_max[cid]], weight, priority)
            else:
                cons_min[cid] = self._add_constraint(c["equal"], False)
                cons_max[cid] = self._add_constraint(c["equal"], True)
                if relaxed:  # Equality.relax: ONE variable shared by both bounds (expressions.ts:240-246)
                    self._relax([cons_min[cid], cons_max[cid]], weight, priority)

        # options (model.ts:338-374)
        self.tolerance = json_model.get("tolerance") or 0
        self.timeout = json_model.get("timeout") or None
        self.useMIRCuts = False
        self.checkForCycles = True
        self.keep_solutions = False
        self.usePresolve = True
        options = json_model.get("options")
        if options:
            if options.get("timeout"):
                self.timeout = options["timeout"]
            if self.tolerance == 0:
                self.tolerance = options.get("tolerance") or 0
            if options.get("useMIRCuts"):
                self.useMIRCuts = options["useMIRCuts"]
            if "exitOnCycles" in options and options["exitOnCycles"] is not None:
                self.checkForCycles = options["exitOnCycles"]
            self.keep_solutions = bool(options.get("keep_solutions"))
            if options.get("presolve") is not None:
                self.usePresolve = options["presolve"]
        self.options = options or {}

        ints = json_model.get("ints") or {}
        binaries = json_model.get("binaries") or {}
        unrestricted = json_model.get("unrestricted") or {}
        objective = opt if isinstance(opt, str) else (next(iter(opt)) if isinstance(opt, dict) and opt else None)
        variables = json_model.get("variables") or {}
        for vid in js_keys(variables):  # model.ts:382-416
            coeffs = variables[vid] or {}
            cost = coeffs.get(objective) if objective is not None else None
            cost = cost if _truthy(cost) else 0
            is_binary = _truthy(binaries.get(vid))
            is_integer = _truthy(ints.get(vid)) or is_binary
            pos = len(self.variables)
            var = {"id": vid, "cost": cost, "index": self._new_index(), "isInteger": is_integer, "priority": 0}
            self.variables.append(var)
            if is_integer:
                self.integerVariables.append(var)
            if _truthy(unrestricted.get(vid)):
                self.unrestricted.append(var["index"])
            if is_binary:  # model.ts:392-395: x <= 1
                self._add_constraint(1, True)["terms"].append((pos, 1))
            for name in js_keys(coeffs):
                if name == objective:
                    continue
                coefficient = coeffs[name]
                if name in cons_min:
                    cons_min[name]["terms"].append((pos, coefficient))
                if name in cons_max:
                    cons_max[name]["terms"].append((pos, coefficient))

        self.integer_index_set = frozenset(v["index"] for v in self.integerVariables)
        self.integer_index_array = np.array([v["index"] for v in self.integerVariables], dtype=np.int64)

    _PRIORITIES = {"required": 0, "strong": 1, "medium": 2, "weak": 3}

    def _relax(self, constraints, weight, priority):
        """createRelaxationVariable + Constraint._relax (expressions.ts:73-94, 186-202)"""
        if priority == 0 or priority == "required":
            return
        w = 1 if weight is None else weight
        p = 1 if priority is None else priority
        if isinstance(p, str):
            p = self._PRIORITIES.get(p, 0)  # model.addVariable's string mapping (model.ts:143-160)
        actual = -w if not self.isMinimization else w
        pos = len(self.variables)
        self.variables.append({"id": "r%d" % self.relaxationIndex, "cost": actual, "index": self._new_index(),
                               "isInteger": False, "priority": p})
        self.relaxationIndex += 1
        for c in constraints:
            c["terms"].append((pos, -1 if c["isUpperBound"] else 1))

    def _new_index(self):
        i = self._next_index
        self._next_index += 1
        return i

    def _add_constraint(self, rhs, is_upper):
        c = {"index": self._new_index(), "isUpperBound": is_upper, "rhs": rhs, "terms": []}
        self.constraints.append(c)
        return c

    # ---- Tableau.setModel -> initialize + _resetMatrix (tableau.ts:292-391) -----------------------------
    def build_tableau(self):
        n, m = len(self.variables), len(self.constraints)
        W, H = n + 1, m + 1
        matrix = np.zeros((H, W), dtype=np.float64)
        vibr = np.full(H, -1, dtype=np.int32)
        vibc = np.full(W, -1, dtype=np.int32)
        coeff = -1 if self.isMinimization else 1
        for v, var in enumerate(self.variables):
            if var["priority"] == 0:
                matrix[0, v + 1] = float(coeff) * float(var["cost"])
            vibc[v + 1] = var["index"]
        for r, c in enumerate(self.constraints, start=1):
            vibr[r] = c["index"]
            if c["isUpperBound"]:
                for pos, coefficient in c["terms"]:
                    matrix[r, pos + 1] = float(coefficient)
                matrix[r, 0] = float(c["rhs"])
            else:
                for pos, coefficient in c["terms"]:
                    matrix[r, pos + 1] = -float(coefficient)  # -0.0 for a zero coefficient, like JS
                matrix[r, 0] = -float(c["rhs"])
        return matrix, vibr, vibc

    def optional_objectives(self):
        """(priorities ascending, rows n x width): Tableau.setOptionalObjective via _resetMatrix (tableau.ts:278-290,335-338)"""
        coeff = -1 if self.isMinimization else 1
        by_priority = {}
        W = len(self.variables) + 1
        for v, var in enumerate(self.variables):
            if var["priority"] != 0:
                row = by_priority.setdefault(var["priority"], np.zeros(W, dtype=np.float64))
                row[v + 1] = float(coeff) * float(var["cost"])
        pr = sorted(by_priority)
        rows = np.stack([by_priority[p] for p in pr]) if pr else np.zeros((0, W), dtype=np.float64)
        return pr, rows
