"""Host-side helpers of the Python mirror (no GPU): the packed cut lists that keep their ctypes pointers and the cached numpy views of engine-owned buffers
(engine.py: PackedCuts, _cut_ptrs, _pinned_view) -- what a batch call no longer rebuilds every time."""
import ctypes as C

import numpy as np

from jslpsolver_amd import _capi
from jslpsolver_amd.engine import PackedCuts, _cut_ptrs, _pinned_view


def _packed():
    offs = np.array([0, 2, 3], dtype=np.int32)
    t = np.array([0, 1, 0], dtype=np.int8)
    v = np.array([4, 5, 6], dtype=np.int32)
    x = np.array([1.5, 2.5, 3.5])
    return PackedCuts((2, offs, t, v, x)), (offs, t, v, x)


def test_packed_cuts_is_the_five_tuple_callers_unpack_and_keeps_its_pointers():
    pk, (offs, t, v, x) = _packed()
    n, o, tt, vv, xx = pk
    assert n == 2 and o is offs and tt is t and vv is v and xx is x and len(pk) == 5 and isinstance(pk, tuple)
    p1, p2 = _cut_ptrs(pk), _cut_ptrs(pk)
    assert p1 is p2  # made once
    assert C.cast(p1[0], C.c_void_p).value == offs.ctypes.data and C.cast(p1[1], C.c_void_p).value == t.ctypes.data
    assert C.cast(p1[2], C.c_void_p).value == v.ctypes.data and C.cast(p1[3], C.c_void_p).value == x.ctypes.data
    assert p1[0][2] == 3 and p1[3][1] == 2.5


def test_a_plain_tuple_still_works():
    pk, arrays = _packed()
    plain = (2,) + arrays
    p = _cut_ptrs(plain)
    assert C.cast(p[0], C.c_void_p).value == arrays[0].ctypes.data and p[2][0] == 4


def test_pinned_view_is_one_view_per_address_shape_and_type():
    buf = np.arange(24, dtype=np.int32)
    p = buf.ctypes.data_as(_capi._i32p)
    a, b = _pinned_view(p, (2, 12)), _pinned_view(buf.ctypes.data_as(_capi._i32p), (2, 12))
    assert a is b and a.shape == (2, 12) and a[1, 3] == 15
    c = _pinned_view(p, (4, 6))
    assert c is not a and c[3, 5] == 23
    buf[15] = -7  # a view, not a copy
    assert a[1, 3] == -7
    other = np.zeros(24, dtype=np.int32)
    assert _pinned_view(other.ctypes.data_as(_capi._i32p), (2, 12)) is not a
    fbuf = np.zeros(4)
    assert _pinned_view(fbuf.ctypes.data_as(_capi._f64p), (2, 2)).dtype == np.float64
