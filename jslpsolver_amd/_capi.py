"""ctypes binding of the C ABI declared in include/jslp_engine.h.

The same binding class serves two libraries that export identical symbols:
  * jslpsolver_amd/csrc/libjslp_hip.so -- the product (hand-written HIP kernels, gfx950)
  * oracle/libjslp_oracle.so           -- TEST ONLY, loaded explicitly by tests / smoke / cpu_baseline
`load_hip()` is the only loader the product uses and it raises when the HIP library is missing:
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("JSLP_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libjslp_hip.so")  # env: debug builds only

JSLP_OK = 0
JSLP_ERR_ARG, JSLP_ERR_DEVICE, JSLP_ERR_NOMEM, JSLP_ERR_STATE, JSLP_ERR_CAPACITY, JSLP_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
JSLP_CUT_MIN = 0
JSLP_CUT_MAX = 1


class SimplexResult(C.Structure):
    """struct jslp_simplex_result (include/jslp_engine.h)"""

    _fields_ = [
        ("feasible", C.c_int32),
        ("bounded", C.c_int32),
        ("optimal", C.c_int32),
        ("unbounded_var_index", C.c_int32),
        ("pivots_phase1", C.c_int32),
        ("pivots_phase2", C.c_int32),
        ("cycle_phase", C.c_int32),
        ("cycle_start", C.c_int32),
        ("cycle_length", C.c_int32),
        ("height", C.c_int32),
        ("obj_cell", C.c_double),
        ("evaluation", C.c_double),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class WorkCounters(C.Structure):
    """struct jslp_work_counters (include/jslp_engine.h)"""

    _fields_ = [(name, C.c_int64) for name in ("relaxations", "simplex_calls", "pivots", "gated_cells", "gated_rows",
                                              "restored_rows", "cut_rows", "height_sum", "resident_aborts", "resident_handovers",
                                              "resident_launches", "resident_refusals", "node_queue_launches", "resident_fetch_retries")]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


_P = C.POINTER
_i32p = _P(C.c_int32)
_f64p = _P(C.c_double)
_i8p = _P(C.c_int8)

# name -> (restype, argtypes); must list EVERY symbol include/jslp_engine.h declares
SYMBOLS = {
    "jslp_backend_name": (C.c_char_p, []),
    "jslp_last_error": (C.c_char_p, []),
    "jslp_device_count": (C.c_int, []),
    "jslp_release_pooled_resources": (None, []),
    "jslp_engine_create": (C.c_int, [_P(C.c_void_p), C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    "jslp_engine_destroy": (None, [C.c_void_p]),
    "jslp_engine_upload": (C.c_int, [C.c_void_p, _f64p, _i32p, _i32p, _i32p, C.c_int32]),
    "jslp_engine_set_optional_objectives": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "jslp_engine_get_optional_objectives": (C.c_int, [C.c_void_p, _f64p, _i32p]),
    "jslp_engine_simplex": (C.c_int, [C.c_void_p, C.c_int, _P(SimplexResult)]),
    "jslp_engine_pivot": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "jslp_engine_save": (C.c_int, [C.c_void_p]),
    "jslp_engine_restore": (C.c_int, [C.c_void_p]),
    "jslp_engine_add_cuts": (C.c_int, [C.c_void_p, C.c_int32, _i8p, _i32p, _f64p]),
    "jslp_engine_relax": (C.c_int, [C.c_void_p, C.c_int32, _i8p, _i32p, _f64p, C.c_int, _P(SimplexResult), _f64p, _i32p]),
    "jslp_engine_relax_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                          _P(SimplexResult), _f64p, _i32p, C.c_int32]),
    "jslp_engine_relax_batch_pinned": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                                 _P(SimplexResult), _P(_f64p), _P(_i32p), _i32p]),
    "jslp_engine_state_record_bytes": (C.c_int32, []),
    "jslp_engine_relax_batch_device": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_int32]),
    "jslp_engine_results_from_states": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _P(SimplexResult)]),
    "jslp_engine_set_integer_variables": (C.c_int, [C.c_void_p, _i32p, C.c_int32]),
    "jslp_engine_apply_mir_cuts": (C.c_int, [C.c_void_p, _i32p]),
    "jslp_engine_mir_round": (C.c_int, [C.c_void_p, C.c_int, _i32p, _P(SimplexResult), _f64p, _i32p]),
    "jslp_engine_simplex_f32": (C.c_int, [C.c_void_p, C.c_double, C.c_int, _P(SimplexResult), _f64p, _i32p, _f64p]),
    "jslp_engine_checkpoint_create": (C.c_int, [C.c_void_p, _i32p]),
    "jslp_engine_checkpoint_restore": (C.c_int, [C.c_void_p, C.c_int32]),
    "jslp_engine_checkpoint_release": (C.c_int, [C.c_void_p, C.c_int32]),
    "jslp_engine_relax_from": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                         _P(SimplexResult), _f64p, _i32p, C.c_int32]),
    "jslp_engine_host_matrix": (C.c_int, [C.c_void_p, _P(_f64p), _P(C.c_int64)]),
    "jslp_engine_set_watched_variables": (C.c_int, [C.c_void_p, _i32p, C.c_int32]),
    "jslp_engine_relax_watched": (C.c_int, [C.c_void_p, C.c_int32, _i8p, _i32p, _f64p, C.c_int, _P(SimplexResult), _i32p, _f64p]),
    "jslp_engine_relax_batch_watched": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int, _P(SimplexResult), _i32p, _f64p]),
    "jslp_engine_relax_batch_watched_pinned": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int, _P(SimplexResult), _P(_i32p), _P(_f64p)]),
    "jslp_engine_relax_batch_watched_device": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jslp_engine_watched_count": (C.c_int32, [C.c_void_p]),
    "jslp_engine_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "jslp_engine_get_counters": (C.c_int, [C.c_void_p, _P(WorkCounters)]),
    "jslp_pool_create": (C.c_int, [_P(C.c_void_p), C.c_void_p, _i32p, C.c_int32]),
    "jslp_pool_destroy": (None, [C.c_void_p]),
    "jslp_pool_size": (C.c_int, [C.c_void_p]),
    "jslp_pool_sync_root": (C.c_int, [C.c_void_p]),
    "jslp_pool_relax_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                        _P(SimplexResult), _f64p, _i32p, C.c_int32]),
    "jslp_pool_relax_batch_pinned": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                               _P(SimplexResult), _P(_f64p), _P(_i32p), _i32p]),
    "jslp_pool_set_watched_variables": (C.c_int, [C.c_void_p, _i32p, C.c_int32]),
    "jslp_pool_relax_batch_watched": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                                _P(SimplexResult), _i32p, _f64p]),
    "jslp_pool_relax_batch_watched_pinned": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p, _i32p, _f64p, C.c_int,
                                                       _P(SimplexResult), _P(_i32p), _P(_f64p)]),
    "jslp_pool_watched_count": (C.c_int32, [C.c_void_p]),
    "jslp_pool_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "jslp_pool_get_counters": (C.c_int, [C.c_void_p, _P(WorkCounters)]),
    "jslp_engine_dims": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p]),
    "jslp_engine_read_rhs": (C.c_int, [C.c_void_p, _f64p, _i32p]),
    "jslp_engine_download": (C.c_int, [C.c_void_p, _f64p, _i32p, _i32p, _i32p, _i32p]),
    "jslp_engine_pivot_trace": (C.c_int, [C.c_void_p, _i32p, C.c_int64, _P(C.c_int64)]),
    "jslp_engine_last_path": (C.c_char_p, [C.c_void_p]),
    "jslp_engine_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "jslp_engine_get_timing": (C.c_int, [C.c_void_p, _f64p, _P(C.c_int64), _f64p]),
}


class EngineError(RuntimeError):
    pass


class Library:
    """A loaded engine library with typed entry points."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise EngineError("engine library not found: %s" % path)
        self.path = path
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(self.dll, name)  # AttributeError => the library does not export the ABI
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)

    @property
    def backend(self):
        return self.jslp_backend_name().decode()

    def check(self, rc, what):
        if rc != JSLP_OK:
            raise EngineError("%s failed (%d): %s" % (what, rc, self.jslp_last_error().decode()))


_hip = None


def load_hip():
    """The product's only loader.  Fails loudly: no HIP library, no engine."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise EngineError(
                "jslpsolver_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % HIP_LIB_PATH)
        _hip = Library(HIP_LIB_PATH)
    return _hip


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr_i32(a):
    return a.ctypes.data_as(_i32p) if a is not None else None


def ptr_f64(a):
    return a.ctypes.data_as(_f64p) if a is not None else None


def ptr_i8(a):
    return a.ctypes.data_as(_i8p) if a is not None else None
