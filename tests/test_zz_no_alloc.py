"""The allocation-free block decode (``rdr_decode_bc_noalloc``) against ``rdr_decode_bc``, through
the C ABI: bit-exact, on two calls in a row (the second smaller, so that it runs over the staging
memory the first one left behind), plus the size limit."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import abi
from renderer_amd.renderer import load_library

pytestmark = pytest.mark.gpu


def _chain_bytes(fmt, w, h, mips):
    bb = 8 if fmt in (abi.FORMAT_BC1, abi.FORMAT_BC4) else 16
    total = 0
    for l in range(mips):
        total += ((max(w >> l, 1) + 3) // 4) * ((max(h >> l, 1) + 3) // 4) * bb
    return total


def _texels(w, h, mips):
    return sum(max(w >> l, 1) * max(h >> l, 1) for l in range(mips))


def _decode(fn, fmt, w, h, mips, blocks):
    out = np.full(_texels(w, h, mips), 0xDEADBEEF, np.uint32)
    rc = fn(0, fmt, w, h, mips, blocks.ctypes.data_as(C.c_void_p), blocks.nbytes, out.ctypes.data_as(C.c_void_p))
    return rc, out


def test_noalloc_decode_matches_allocating_decode():
    L = load_library()
    rng = np.random.default_rng(7)
    # every call is smaller than the one before it, and in another format
    for fmt, w, h, mips in ((abi.FORMAT_BC3, 256, 128, 9), (abi.FORMAT_BC7, 96, 40, 4), (abi.FORMAT_BC1, 52, 36, 3),
                            (abi.FORMAT_BC4, 8, 8, 4)):
        blocks = rng.integers(0, 256, _chain_bytes(fmt, w, h, mips), dtype=np.uint8)
        rc, want = _decode(L.rdr_decode_bc, fmt, w, h, mips, blocks)
        assert rc == 0, L.rdr_last_error().decode()
        rc, got = _decode(L.rdr_decode_bc_noalloc, fmt, w, h, mips, blocks)
        assert rc == 0, L.rdr_last_error().decode()
        assert np.array_equal(got, want), (fmt, w, h, mips)


def test_noalloc_decode_largest_chain_and_limit():
    L = load_library()
    rng = np.random.default_rng(11)
    fmt, w, h, mips = abi.FORMAT_BC2, 1024, 1024, 11  # the stated maximum
    blocks = rng.integers(0, 256, _chain_bytes(fmt, w, h, mips), dtype=np.uint8)
    rc, want = _decode(L.rdr_decode_bc, fmt, w, h, mips, blocks)
    assert rc == 0, L.rdr_last_error().decode()
    rc, got = _decode(L.rdr_decode_bc_noalloc, fmt, w, h, mips, blocks)
    assert rc == 0, L.rdr_last_error().decode()
    assert np.array_equal(got, want)
    # one step beyond: refused with a message, nothing written
    fmt, w, h, mips = abi.FORMAT_BC1, 2048, 1024, 1
    blocks = np.zeros(_chain_bytes(fmt, w, h, mips), np.uint8)
    rc, got = _decode(L.rdr_decode_bc_noalloc, fmt, w, h, mips, blocks)
    assert rc != 0 and "staging" in L.rdr_last_error().decode()
    assert (got == 0xDEADBEEF).all()
