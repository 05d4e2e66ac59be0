"""CPU tests of the Python host side's key marshalling for ehx_set_batch (embeddinghub_amd/space.py::marshal_keys): the
pointer / length arrays must describe exactly the keys given (empty keys, embedded NUL bytes, non-ASCII text), and the
C entry point must accept them as its `const char* const*` / `const size_t*` arguments."""
import ctypes as C

import numpy as np

from embeddinghub_amd import _lib
from embeddinghub_amd.space import marshal_keys


def test_marshalled_keys_round_trip():
    keys = ["k%d" % i for i in range(3000)] + [b"", b"with\x00nul", "ünï", b"", "last"]
    n, arr, lens, keep = marshal_keys(keys)
    assert n == len(keys)
    ptrs = C.cast(arr, C.POINTER(C.c_void_p))
    for i, k in enumerate(keys):
        kb = k.encode() if isinstance(k, str) else k
        assert lens[i] == len(kb)
        assert (C.string_at(ptrs[i], lens[i]) if lens[i] else b"") == kb
    assert marshal_keys([])[0] == 0


def test_the_c_abi_accepts_the_marshalled_arrays():
    L = _lib.load()
    n, arr, lens, keep = marshal_keys(["a", "b"])
    rows = np.zeros((2, 4), dtype=np.float32)
    rc = L.ehx_set_batch(None, n, arr, lens, rows.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == _lib.EINVAL          # "space is NULL": the call got as far as the argument check, no GPU needed
