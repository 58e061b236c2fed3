"""Host-side mirror of the reference `Tableau` hot-path methods (src/tableau/tableau.ts:103-229) on top of
the C ABI.  Method names follow the reference: simplex / pivot / save / restore / addCutConstraints /
applyCuts, so parity tests read like the reference's own tests.
"""
import numpy as np

from . import _capi
from ._capi import SimplexResult


class PackedCuts(tuple):
    """(n_nodes, cut_offsets, type, var_index, value) as Tableau.pack_cut_lists returns it -- a plain 5-tuple to every caller -- that also keeps the four
    ctypes pointers of its arrays: `a.ctypes.data_as(...)` costs ~2 us per array, four times per call, which is a fifth of a 40 us single-node batch."""

    def ptrs(self):
        p = self.__dict__.get("_ptrs")
        if p is None:
            _, offs, t, v, x = self
            p = self.__dict__["_ptrs"] = (_capi.ptr_i32(offs), _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x))
        return p


def _cut_ptrs(packed):
    f = getattr(packed, "ptrs", None)
    if f is not None:
        return f()
    _, offs, t, v, x = packed
    return _capi.ptr_i32(offs), _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x)


_VIEWS = {}


def _pinned_view(ptr, shape):
    """numpy view of an engine-owned pinned buffer (np.ctypeslib.as_array costs ~2 us a time): one view object per (address, shape, type) -- the
    engine hands out the same buffer call after call; a re-allocated buffer has another address and gets another view"""
    key = (_capi.C.cast(ptr, _capi.C.c_void_p).value, shape, ptr._type_)
    v = _VIEWS.get(key)
    if v is None:
        if len(_VIEWS) > 256:
            _VIEWS.clear()
        v = _VIEWS[key] = np.ctypeslib.as_array(ptr, shape=shape)
    return v


class Tableau:
    """Device-resident dense simplex tableau (one `jslp_engine`).

    matrix is the reference layout (tableau.ts:49-54): row-major height x width float64, row 0 = reduced
    costs, column 0 = RHS.
    """

    def __init__(self, matrix, var_index_by_row, var_index_by_col, unrestricted=(), precision=1e-8,
                 row_capacity=None, device=0, lib=None, optional_objectives=None, integer_variables=None):
        self.lib = lib if lib is not None else _capi.load_hip()
        matrix = _capi.as_f64(matrix)
        if matrix.ndim != 2:
            raise ValueError("matrix must be height x width")
        self.height0, self.width = matrix.shape
        self.precision = float(precision)
        self.row_capacity = int(row_capacity if row_capacity is not None else self.height0)
        self._h = _capi.C.c_void_p()
        self.lib.check(self.lib.jslp_engine_create(_capi.C.byref(self._h), int(device), self.height0, self.width,
                                                   self.row_capacity, self.precision), "jslp_engine_create")
        vibr = _capi.as_i32(var_index_by_row)
        vibc = _capi.as_i32(var_index_by_col)
        unr = _capi.as_i32(list(unrestricted))
        if vibr.shape[0] != self.height0 or vibc.shape[0] != self.width:
            raise ValueError("index maps do not match the matrix shape")
        self.lib.check(self.lib.jslp_engine_upload(self._h, _capi.ptr_f64(matrix), _capi.ptr_i32(vibr),
                                                   _capi.ptr_i32(vibc), _capi.ptr_i32(unr), int(unr.shape[0])),
                       "jslp_engine_upload")
        self.n_optional = 0
        if optional_objectives is not None and len(optional_objectives) > 0:
            oo = _capi.as_f64(optional_objectives)
            if oo.ndim != 2 or oo.shape[1] != self.width:
                raise ValueError("optional objectives must be n x width")
            self.lib.check(self.lib.jslp_engine_set_optional_objectives(self._h, int(oo.shape[0]), _capi.ptr_f64(oo)),
                           "jslp_engine_set_optional_objectives")
            self.n_optional = int(oo.shape[0])
        if integer_variables is not None and len(integer_variables) > 0:  # variable.isInteger for the MIR cuts
            iv = _capi.as_i32(list(integer_variables))
            self.lib.check(self.lib.jslp_engine_set_integer_variables(self._h, _capi.ptr_i32(iv), int(iv.shape[0])),
                           "jslp_engine_set_integer_variables")
        # tableau scalars the reference keeps on the Tableau object (tableau.ts:59-61,83-87)
        self.feasible = True
        self.bounded = True
        self.evaluation = 0.0
        self.simplexIters = 0
        self.bestPossibleEval = 0.0
        self.unboundedVarIndex = None
        self.last = None

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.jslp_engine_destroy(self._h)
            self._h = _capi.C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference Tableau surface ------------------------------------------------------------
    @property
    def height(self):
        h = _capi.C.c_int32()
        self.lib.check(self.lib.jslp_engine_dims(self._h, _capi.C.byref(h), None, None), "jslp_engine_dims")
        return h.value

    def _absorb(self, res):
        """fold a jslp_simplex_result into the Tableau scalars exactly as simplex.ts/tableau.ts do"""
        self.last = res
        self.feasible = bool(res.feasible)
        self.bounded = bool(res.bounded)
        if res.optimal:  # setEvaluation + simplexIters += 1 (simplex.ts:265-269, tableau.ts:420-430)
            self.evaluation = res.evaluation
            if self.simplexIters == 0:
                self.bestPossibleEval = res.evaluation
            self.simplexIters += 1
        elif not res.bounded:  # simplex.ts:298-303
            self.evaluation = float("-inf")
            self.unboundedVarIndex = res.unbounded_var_index
        return res

    def simplex(self, check_cycles=True):
        res = SimplexResult()
        self.lib.check(self.lib.jslp_engine_simplex(self._h, int(bool(check_cycles)), _capi.C.byref(res)),
                       "jslp_engine_simplex")
        return self._absorb(res)

    def pivot(self, row, col):
        self.lib.check(self.lib.jslp_engine_pivot(self._h, int(row), int(col)), "jslp_engine_pivot")

    def save(self):
        self.lib.check(self.lib.jslp_engine_save(self._h), "jslp_engine_save")

    def restore(self):
        self.lib.check(self.lib.jslp_engine_restore(self._h), "jslp_engine_restore")

    @staticmethod
    def _pack_cuts(cuts):
        n = len(cuts)
        t = np.array([_capi.JSLP_CUT_MIN if c["type"] == "min" else _capi.JSLP_CUT_MAX for c in cuts], dtype=np.int8)
        v = np.array([c["varIndex"] for c in cuts], dtype=np.int32)
        x = np.array([c["value"] for c in cuts], dtype=np.float64)
        return n, t, v, x

    def addCutConstraints(self, cuts):
        n, t, v, x = self._pack_cuts(cuts)
        self.lib.check(self.lib.jslp_engine_add_cuts(self._h, n, _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x)),
                       "jslp_engine_add_cuts")

    def applyCuts(self, cuts, check_cycles=True):
        """BranchAndCutService.applyCuts (branch-and-cut.ts:33-37) as ONE engine call; returns (result, rhs,
        varIndexByRow) -- the read-back isIntegral()/getMostFractionalVar() need."""
        n, t, v, x = self._pack_cuts(cuts)
        res = SimplexResult()
        rhs = np.empty(self.row_capacity, dtype=np.float64)
        vibr = np.empty(self.row_capacity, dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_relax(self._h, n, _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x),
                                                  int(bool(check_cycles)), _capi.C.byref(res), _capi.ptr_f64(rhs),
                                                  _capi.ptr_i32(vibr)), "jslp_engine_relax")
        self._absorb(res)
        return res, rhs[:res.height], vibr[:res.height]

    def pack_cut_lists(self, cut_lists):
        """cut lists -> the flat arrays jslp_engine_relax_batch takes (do this once when the same nodes are re-evaluated)"""
        n_nodes = len(cut_lists)
        offs = np.zeros(n_nodes + 1, dtype=np.int32)
        flat = []
        for i, cuts in enumerate(cut_lists):
            flat.extend(cuts)
            offs[i + 1] = len(flat)
        _, t, v, x = self._pack_cuts(flat)
        return PackedCuts((n_nodes, offs, t, v, x))

    def applyCutsBatch(self, cut_lists, check_cycles=True, packed=None, want_rows=True, copy=True):
        """Independent branch-and-bound nodes in one call.  copy=False returns views of the engine's pinned read-back
        buffer (jslp_engine_relax_batch_pinned): valid until the next call on this tableau."""
        pk = packed if packed is not None else self.pack_cut_lists(cut_lists)
        n_nodes, offs, t, v, x = pk
        out = (SimplexResult * max(n_nodes, 1))()
        stride = self.row_capacity
        if copy:
            rhs = np.empty((max(n_nodes, 1), stride), dtype=np.float64)
            vibr = np.empty((max(n_nodes, 1), stride), dtype=np.int32) if want_rows else None
            self.lib.check(self.lib.jslp_engine_relax_batch(self._h, n_nodes, _capi.ptr_i32(offs), _capi.ptr_i8(t),
                                                            _capi.ptr_i32(v), _capi.ptr_f64(x), int(bool(check_cycles)),
                                                            out, _capi.ptr_f64(rhs), _capi.ptr_i32(vibr), stride),
                           "jslp_engine_relax_batch")
            return [out[i] for i in range(n_nodes)], rhs, vibr
        p_rhs = _capi._f64p()
        p_rows = _capi._i32p()
        c_stride = _capi.C.c_int32()
        self.lib.check(self.lib.jslp_engine_relax_batch_pinned(
            self._h, n_nodes, *_cut_ptrs(pk),
            int(bool(check_cycles)), out, _capi.C.byref(p_rhs), _capi.C.byref(p_rows) if want_rows else None,
            _capi.C.byref(c_stride)), "jslp_engine_relax_batch_pinned")
        shape = (max(n_nodes, 1), c_stride.value)
        rhs = _pinned_view(p_rhs, shape)
        vibr = _pinned_view(p_rows, shape) if want_rows else None
        return out, rhs, vibr

    # ---- outcomes left in device memory (the one-process-per-GPU path: sharding.py) ----------------------------------
    def state_record_bytes(self):
        return int(self.lib.jslp_engine_state_record_bytes())

    def applyCutsBatchDevice(self, packed, check_cycles, states_ptr, rhs_ptr, rows_ptr, row_stride):
        """jslp_engine_relax_batch_device: raw addresses of memory on the engine's device (torch tensors' data_ptr(); plain host
        memory for the oracle library) receive the per-node state records, RHS columns and row maps; nothing is copied back"""
        n_nodes = packed[0]
        self.lib.check(self.lib.jslp_engine_relax_batch_device(
            self._h, n_nodes, *_cut_ptrs(packed), int(bool(check_cycles)),
            _capi.C.c_void_p(states_ptr), _capi.C.c_void_p(rhs_ptr), _capi.C.c_void_p(rows_ptr), int(row_stride)),
            "jslp_engine_relax_batch_device")

    def applyCutsBatchWatchedDevice(self, packed, check_cycles, states_ptr, rows_ptr, values_ptr):
        """jslp_engine_relax_batch_watched_device: the COMPACT outcome (state record + row / RHS cell of the watched variables per node)
        left in memory of the engine's device -- the exchange payload of the multi-process path (sharding.py)"""
        n_nodes = packed[0]
        self.lib.check(self.lib.jslp_engine_relax_batch_watched_device(
            self._h, n_nodes, *_cut_ptrs(packed), int(bool(check_cycles)),
            _capi.C.c_void_p(states_ptr), _capi.C.c_void_p(rows_ptr), _capi.C.c_void_p(values_ptr)),
            "jslp_engine_relax_batch_watched_device")

    def watched_count(self):
        """how many variables set_watched_variables registered on the engine (jslp_engine_watched_count)"""
        return int(self.lib.jslp_engine_watched_count(self._h))

    def results_from_states(self, states_u8, n_nodes):
        """raw state records (host bytes: this rank's or, after the exchange, another rank's) -> SimplexResult array"""
        out = (SimplexResult * max(n_nodes, 1))()
        buf = np.ascontiguousarray(states_u8, dtype=np.uint8)
        self.lib.check(self.lib.jslp_engine_results_from_states(self._h, _capi.C.c_void_p(buf.ctypes.data), int(n_nodes), out),
                       "jslp_engine_results_from_states")
        return out

    # ---- compact read-back, work counters, pinned host build --------------------------------------------------
    def set_watched_variables(self, var_indexes):
        """the variable indexes the branch-and-bound tree reads between relaxations (the integer variables)"""
        w = _capi.as_i32(list(var_indexes))
        self.lib.check(self.lib.jslp_engine_set_watched_variables(self._h, _capi.ptr_i32(w), int(w.shape[0])),
                       "jslp_engine_set_watched_variables")
        self.n_watched = int(w.shape[0])
        self.watched = [int(i) for i in w]  # (what is registered: a caller that borrows the registration puts it back)

    def applyCutsWatched(self, cuts, check_cycles=True):
        """applyCuts whose read-back is only rowByVarIndex / the RHS cell of the watched variables"""
        n, t, v, x = self._pack_cuts(cuts)
        res = SimplexResult()
        rows = np.empty(self.n_watched, dtype=np.int32)
        vals = np.empty(self.n_watched, dtype=np.float64)
        self.lib.check(self.lib.jslp_engine_relax_watched(self._h, n, _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x),
                                                          int(bool(check_cycles)), _capi.C.byref(res), _capi.ptr_i32(rows),
                                                          _capi.ptr_f64(vals)), "jslp_engine_relax_watched")
        self._absorb(res)
        return res, rows, vals

    def applyCutsBatchWatched(self, cut_lists, check_cycles=True, packed=None, copy=True):
        """applyCutsBatch whose read-back is, per node, rowByVarIndex / the RHS cell of the watched variables only: what a host
        that walks the tree itself reads per node, and ~10x fewer bytes over PCIe than the full RHS columns + row maps.
        copy=False returns views of the engine's pinned buffer (valid until the next call on this tableau)."""
        pk = packed if packed is not None else self.pack_cut_lists(cut_lists)
        n_nodes, offs, t, v, x = pk
        out = (SimplexResult * max(n_nodes, 1))()
        if not copy:
            p_rows = _capi._i32p()
            p_vals = _capi._f64p()
            self.lib.check(self.lib.jslp_engine_relax_batch_watched_pinned(
                self._h, n_nodes, *_cut_ptrs(pk), int(bool(check_cycles)), out,
                _capi.C.byref(p_rows), _capi.C.byref(p_vals)), "jslp_engine_relax_batch_watched_pinned")
            shape = (max(n_nodes, 1), self.n_watched)
            return out, _pinned_view(p_rows, shape), _pinned_view(p_vals, shape)
        rows = np.empty((max(n_nodes, 1), self.n_watched), dtype=np.int32)
        vals = np.empty((max(n_nodes, 1), self.n_watched), dtype=np.float64)
        self.lib.check(self.lib.jslp_engine_relax_batch_watched(self._h, n_nodes, _capi.ptr_i32(offs), _capi.ptr_i8(t), _capi.ptr_i32(v),
                                                                _capi.ptr_f64(x), int(bool(check_cycles)), out, _capi.ptr_i32(rows),
                                                                _capi.ptr_f64(vals)), "jslp_engine_relax_batch_watched")
        return out, rows, vals

    def set_counting(self, enabled):
        self.lib.check(self.lib.jslp_engine_set_counting(self._h, int(bool(enabled))), "jslp_engine_set_counting")

    def get_counters(self):
        c = _capi.WorkCounters()
        self.lib.check(self.lib.jslp_engine_get_counters(self._h, _capi.C.byref(c)), "jslp_engine_get_counters")
        return c.as_dict()

    def host_matrix(self):
        """height x width float64 view of the engine's pinned build buffer (zero-filled on every call); fill it and pass
        it to upload_host_matrix()"""
        p = _capi._f64p()
        n = _capi.C.c_int64()
        self.lib.check(self.lib.jslp_engine_host_matrix(self._h, _capi.C.byref(p), _capi.C.byref(n)), "jslp_engine_host_matrix")
        return np.ctypeslib.as_array(p, shape=(self.height0, self.width))

    def upload(self, matrix, var_index_by_row, var_index_by_col, unrestricted=()):
        """Tableau hand-over again (a new model of the same shape, or the buffer of host_matrix())"""
        vibr = _capi.as_i32(var_index_by_row)
        vibc = _capi.as_i32(var_index_by_col)
        unr = _capi.as_i32(list(unrestricted))
        m = np.ascontiguousarray(matrix, dtype=np.float64)
        self.lib.check(self.lib.jslp_engine_upload(self._h, _capi.ptr_f64(m), _capi.ptr_i32(vibr), _capi.ptr_i32(vibc),
                                                   _capi.ptr_i32(unr), int(unr.shape[0])), "jslp_engine_upload")

    # ---- fp32 experiment ------------------------------------------------------------------------------------
    def simplex_f32(self, precision, check_cycles=True):
        """simplex() on an fp32 copy of the live tableau with tolerance `precision` (SURVEY.md 8d, config 5's fp32-vs-fp64
        sweep).  The fp64 tableau and the Tableau scalars are left alone.  Returns (result, rhs, varIndexByRow, device ms)."""
        res = SimplexResult()
        rhs = np.empty(self.row_capacity, dtype=np.float64)
        vibr = np.empty(self.row_capacity, dtype=np.int32)
        ms = _capi.C.c_double(0.0)
        self.lib.check(self.lib.jslp_engine_simplex_f32(self._h, float(precision), int(bool(check_cycles)), _capi.C.byref(res),
                                                        _capi.ptr_f64(rhs), _capi.ptr_i32(vibr), _capi.C.byref(ms)),
                       "jslp_engine_simplex_f32")
        return res, rhs[:res.height], vibr[:res.height], ms.value

    # ---- MIR cuts (cutting-strategies.ts:74-212) --------------------------------------------------------
    def applyMIRCuts(self):
        """Tableau.applyMIRCuts (:199-212); returns the number of rows appended"""
        n = _capi.C.c_int32(0)
        self.lib.check(self.lib.jslp_engine_apply_mir_cuts(self._h, _capi.C.byref(n)), "jslp_engine_apply_mir_cuts")
        return n.value

    def mirRound(self, check_cycles=True):
        """one turn of the services' MIR loop (branch-and-cut.ts:41-43): applyMIRCuts + simplex + read-back.
        Returns (rows appended, result, rhs, varIndexByRow)."""
        n = _capi.C.c_int32(0)
        res = SimplexResult()
        rhs = np.empty(self.row_capacity, dtype=np.float64)
        vibr = np.empty(self.row_capacity, dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_mir_round(self._h, int(bool(check_cycles)), _capi.C.byref(n), _capi.C.byref(res),
                                                      _capi.ptr_f64(rhs), _capi.ptr_i32(vibr)), "jslp_engine_mir_round")
        self._absorb(res)
        return n.value, res, rhs[:res.height], vibr[:res.height]

    # ---- checkpoints (incremental-branch-and-cut.ts:31-107) -------------------------------------------
    def createCheckpoint(self):
        """createCheckpoint (:55-70): the matrix and the index maps stay in HBM; the host part of a StateCheckpoint
        (`evaluation`, `feasible`, :42-43) travels in the returned dict."""
        cid = _capi.C.c_int32(-1)
        self.lib.check(self.lib.jslp_engine_checkpoint_create(self._h, _capi.C.byref(cid)), "jslp_engine_checkpoint_create")
        return {"id": cid.value, "evaluation": self.evaluation, "feasible": self.feasible}

    def restoreCheckpoint(self, checkpoint):
        """restoreCheckpoint (:72-107)"""
        self.lib.check(self.lib.jslp_engine_checkpoint_restore(self._h, checkpoint["id"]), "jslp_engine_checkpoint_restore")
        self.evaluation = checkpoint["evaluation"]
        self.feasible = checkpoint["feasible"]

    def releaseCheckpoint(self, checkpoint):
        self.lib.check(self.lib.jslp_engine_checkpoint_release(self._h, checkpoint["id"]), "jslp_engine_checkpoint_release")

    def applyCutsFrom(self, checkpoint, cut_lists, check_cycles=True):
        """applyIncrementalCuts' fast path (:248-253) for one or more children of `checkpoint` in ONE engine call:
        restoreCheckpoint + addCutConstraints(cuts) + simplex + read-back per node.  Returns per node
        (result, rhs, varIndexByRow); the tableau scalars end as after the LAST node."""
        n_nodes, offs, t, v, x = self.pack_cut_lists(cut_lists)
        out = (SimplexResult * max(n_nodes, 1))()
        stride = self.row_capacity
        rhs = np.empty((max(n_nodes, 1), stride), dtype=np.float64)
        vibr = np.empty((max(n_nodes, 1), stride), dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_relax_from(self._h, checkpoint["id"], n_nodes, _capi.ptr_i32(offs),
                                                       _capi.ptr_i8(t), _capi.ptr_i32(v), _capi.ptr_f64(x),
                                                       int(bool(check_cycles)), out, _capi.ptr_f64(rhs),
                                                       _capi.ptr_i32(vibr), stride), "jslp_engine_relax_from")
        return [(out[i], rhs[i, :out[i].height], vibr[i, :out[i].height]) for i in range(n_nodes)]

    def absorb_from(self, checkpoint, res):
        """the tableau scalars after restoreCheckpoint(checkpoint) + simplex() with outcome `res`"""
        self.evaluation = checkpoint["evaluation"]
        self.feasible = checkpoint["feasible"]
        return self._absorb(res)

    # ---- read-back --------------------------------------------------------------------------------
    def read_rhs(self):
        h = self.height
        rhs = np.empty(h, dtype=np.float64)
        vibr = np.empty(h, dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_read_rhs(self._h, _capi.ptr_f64(rhs), _capi.ptr_i32(vibr)),
                       "jslp_engine_read_rhs")
        return rhs, vibr

    def download(self):
        h = _capi.C.c_int32()
        w = _capi.C.c_int32()
        n = _capi.C.c_int32()
        self.lib.check(self.lib.jslp_engine_dims(self._h, _capi.C.byref(h), _capi.C.byref(w), _capi.C.byref(n)),
                       "jslp_engine_dims")
        m = np.empty((h.value, w.value), dtype=np.float64)
        vibr = np.empty(h.value, dtype=np.int32)
        vibc = np.empty(w.value, dtype=np.int32)
        rbv = np.empty(n.value, dtype=np.int32)
        cbv = np.empty(n.value, dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_download(self._h, _capi.ptr_f64(m), _capi.ptr_i32(vibr), _capi.ptr_i32(vibc),
                                                     _capi.ptr_i32(rbv), _capi.ptr_i32(cbv)), "jslp_engine_download")
        return m, vibr, vibc, rbv, cbv

    def optional_objectives(self):
        """current optionalObjectives[o].reducedCosts rows (n x width)"""
        rows = np.zeros((max(self.n_optional, 1), self.width), dtype=np.float64)
        n = _capi.C.c_int32()
        self.lib.check(self.lib.jslp_engine_get_optional_objectives(self._h, _capi.ptr_f64(rows), _capi.C.byref(n)),
                       "jslp_engine_get_optional_objectives")
        return rows[:n.value]

    def pivot_trace(self):
        n = _capi.C.c_int64()
        self.lib.check(self.lib.jslp_engine_pivot_trace(self._h, None, 0, _capi.C.byref(n)), "jslp_engine_pivot_trace")
        buf = np.empty(2 * max(n.value, 1), dtype=np.int32)
        self.lib.check(self.lib.jslp_engine_pivot_trace(self._h, _capi.ptr_i32(buf), n.value, _capi.C.byref(n)),
                       "jslp_engine_pivot_trace")
        return buf[:2 * n.value].reshape(-1, 2)

    def last_path(self):
        return self.lib.jslp_engine_last_path(self._h).decode()

    def set_timing(self, enabled):
        self.lib.check(self.lib.jslp_engine_set_timing(self._h, int(bool(enabled))), "jslp_engine_set_timing")

    def get_timing(self):
        ms = _capi.C.c_double()
        n = _capi.C.c_int64()
        tot = _capi.C.c_double()
        self.lib.check(self.lib.jslp_engine_get_timing(self._h, _capi.C.byref(ms), _capi.C.byref(n), _capi.C.byref(tot)),
                       "jslp_engine_get_timing")
        return ms.value, n.value, tot.value


class DevicePool:
    """jslp_pool: `tableau` (the primary) plus one engine per further device ordinal, each holding the primary's saved
    root; batches of independent nodes are split over the members.  devices[0] must be the primary's."""

    def __init__(self, tableau, devices):
        self.t = tableau
        self.lib = tableau.lib
        d = _capi.as_i32(list(devices))
        self._p = _capi.C.c_void_p()
        self.lib.check(self.lib.jslp_pool_create(_capi.C.byref(self._p), tableau._h, _capi.ptr_i32(d), int(d.shape[0])),
                       "jslp_pool_create")
        self.n_watched = 0  # (informational: the outputs are sized with jslp_pool_watched_count)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.lib.jslp_pool_destroy(self._p)
            self._p = _capi.C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self.lib.jslp_pool_size(self._p)

    def sync_root(self):
        self.lib.check(self.lib.jslp_pool_sync_root(self._p), "jslp_pool_sync_root")

    def applyCutsBatch(self, cut_lists, check_cycles=True, packed=None, want_rows=True, copy=True):
        """Tableau.applyCutsBatch over every member of the pool"""
        t = self.t
        pk = packed if packed is not None else t.pack_cut_lists(cut_lists)
        n_nodes, offs, ty, v, x = pk
        out = (SimplexResult * max(n_nodes, 1))()
        stride = t.row_capacity
        if copy:
            rhs = np.empty((max(n_nodes, 1), stride), dtype=np.float64)
            vibr = np.empty((max(n_nodes, 1), stride), dtype=np.int32) if want_rows else None
            self.lib.check(self.lib.jslp_pool_relax_batch(self._p, n_nodes, _capi.ptr_i32(offs), _capi.ptr_i8(ty), _capi.ptr_i32(v),
                                                          _capi.ptr_f64(x), int(bool(check_cycles)), out, _capi.ptr_f64(rhs),
                                                          _capi.ptr_i32(vibr), stride), "jslp_pool_relax_batch")
            return [out[i] for i in range(n_nodes)], rhs, vibr
        p_rhs = _capi._f64p()
        p_rows = _capi._i32p()
        c_stride = _capi.C.c_int32()
        self.lib.check(self.lib.jslp_pool_relax_batch_pinned(
            self._p, n_nodes, *_cut_ptrs(pk), int(bool(check_cycles)),
            out, _capi.C.byref(p_rhs), _capi.C.byref(p_rows) if want_rows else None, _capi.C.byref(c_stride)),
            "jslp_pool_relax_batch_pinned")
        shape = (max(n_nodes, 1), c_stride.value)
        rhs = _pinned_view(p_rhs, shape)
        vibr = _pinned_view(p_rows, shape) if want_rows else None
        return out, rhs, vibr

    def set_watched_variables(self, var_indexes):
        """the variables whose row / RHS cell the compact read-back returns, on every member (jslp_pool_set_watched_variables)"""
        idx = _capi.as_i32(list(var_indexes))
        self.lib.check(self.lib.jslp_pool_set_watched_variables(self._p, _capi.ptr_i32(idx), int(idx.shape[0])),
                       "jslp_pool_set_watched_variables")
        self.n_watched = int(idx.shape[0])

    def applyCutsBatchWatched(self, cut_lists, check_cycles=True, packed=None, copy=True):
        """Tableau.applyCutsBatchWatched over every member of the pool: per node rowByVarIndex / the RHS cell of the watched
        variables (what mip-utils.ts:43-61, 100-126 read between relaxations); copy=False: views of the pool's pinned buffer"""
        t = self.t
        pk = packed if packed is not None else t.pack_cut_lists(cut_lists)
        n_nodes, offs, ty, v, x = pk
        out = (SimplexResult * max(n_nodes, 1))()
        # (the outputs are sized by what the LIBRARY will write -- jslp_pool_watched_count -- not by a Python-side shadow
        #  that a direct set_watched_variables on the primary engine would leave behind)
        n_watched = int(self.lib.jslp_pool_watched_count(self._p))
        if n_watched <= 0:
            raise _capi.EngineError("DevicePool.applyCutsBatchWatched: %s" % ("call set_watched_variables first" if n_watched == 0 else
                                    "the members' watched variables differ (set them through DevicePool.set_watched_variables)"))
        shape = (max(n_nodes, 1), n_watched)
        if not copy:
            p_rows = _capi._i32p()
            p_vals = _capi._f64p()
            self.lib.check(self.lib.jslp_pool_relax_batch_watched_pinned(
                self._p, n_nodes, *_cut_ptrs(pk), int(bool(check_cycles)), out,
                _capi.C.byref(p_rows), _capi.C.byref(p_vals)), "jslp_pool_relax_batch_watched_pinned")
            return out, _pinned_view(p_rows, shape), _pinned_view(p_vals, shape)
        rows = np.empty(shape, dtype=np.int32)
        vals = np.empty(shape, dtype=np.float64)
        self.lib.check(self.lib.jslp_pool_relax_batch_watched(self._p, n_nodes, _capi.ptr_i32(offs), _capi.ptr_i8(ty), _capi.ptr_i32(v),
                                                              _capi.ptr_f64(x), int(bool(check_cycles)), out, _capi.ptr_i32(rows),
                                                              _capi.ptr_f64(vals)), "jslp_pool_relax_batch_watched")
        return out, rows, vals

    def set_counting(self, enabled):
        self.lib.check(self.lib.jslp_pool_set_counting(self._p, int(bool(enabled))), "jslp_pool_set_counting")

    def get_counters(self):
        c = _capi.WorkCounters()
        self.lib.check(self.lib.jslp_pool_get_counters(self._p, _capi.C.byref(c)), "jslp_pool_get_counters")
        return c.as_dict()


def pivot_digest(pairs):
    """FNV-1a-32 over the (row, col) arguments of every pivot."""
    h = 2166136261
    for r, c in np.asarray(pairs, dtype=np.int64).reshape(-1, 2):
        h = ((h ^ int(r)) * 16777619) & 0xFFFFFFFF
        h = ((h ^ int(c)) * 16777619) & 0xFFFFFFFF
    return "%x" % h
