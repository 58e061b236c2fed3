"""Loaders for the golden vectors in tests/golden (generated from the reference by gen_golden.js)."""
import glob
import gzip
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _num(x):
    if isinstance(x, str):
        return {"Infinity": float("inf"), "-Infinity": float("-inf"), "NaN": float("nan"), "-0": -0.0}[x]
    return float(x)


def load(path):
    with gzip.open(path, "rt") as fh:
        return json.load(fh)


def fixture_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "fixtures", "*.json.gz")))


def synthetic_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "synthetic", "*.json.gz")))


def ident(path):
    return os.path.basename(path)[:-len(".json.gz")]


def dense_tableau(tab):
    """(matrix HxW, varIndexByRow, varIndexByCol) from the sparse dump of Tableau.setModel's output"""
    H, W = tab["height"], tab["width"]
    m = np.zeros((H, W), dtype=np.float64)
    if tab["rows"] is not None:
        vals = np.array([_num(v) for v in tab["vals"]], dtype=np.float64)
        m[np.array(tab["rows"], dtype=np.int64), np.array(tab["cols"], dtype=np.int64)] = vals
    vibr = np.array([-1 if v is None else v for v in tab["varIndexByRow"]], dtype=np.int32)
    vibc = np.array([-1 if v is None else v for v in tab["varIndexByCol"]], dtype=np.int32)
    return m, vibr, vibc


def sha_matrix(m):
    return hashlib.sha256(np.ascontiguousarray(m, dtype=np.float64).tobytes()).hexdigest()


def sha_rhs(rhs, vibr):
    return hashlib.sha256(np.ascontiguousarray(rhs, dtype=np.float64).tobytes() +
                          np.ascontiguousarray(vibr, dtype=np.int32).tobytes()).hexdigest()


def num(x):
    return _num(x)
