"""The allocation-free block decode (``rdr_decode_bc_noalloc``) against ``rdr_decode_bc``, through
the C ABI: bit-exact, on two calls in a row (the second smaller, so that it runs over the staging
memory the first one left behind), plus the size limit, and the property the entry point exists for: a call leaves
the device's free memory as it found it."""
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


def _free_device_bytes():
    """hipMemGetInfo's free figure for the current device (the HIP runtime the library itself is linked against)."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_decode_allocates_nothing():
    """Free device memory is the same before and after a call (the first call on a device may still load the code
    object and look the staging array's address up, so it is made beforehand); the allocating entry point is the
    control: its staging buffers are freed again, but memory the runtime's pool took for them may stay taken, so only
    the no-alloc side is held to equality."""
    L = load_library()
    rng = np.random.default_rng(3)
    fmt, w, h, mips = abi.FORMAT_BC7_SRGB, 512, 512, 10
    blocks = rng.integers(0, 256, _chain_bytes(fmt, w, h, mips), dtype=np.uint8)
    rc, want = _decode(L.rdr_decode_bc, fmt, w, h, mips, blocks)
    assert rc == 0, L.rdr_last_error().decode()
    rc, got = _decode(L.rdr_decode_bc_noalloc, fmt, w, h, mips, blocks)  # warm: code object, symbol address
    assert rc == 0 and np.array_equal(got, want)
    before = _free_device_bytes()
    for _ in range(3):
        rc, got = _decode(L.rdr_decode_bc_noalloc, fmt, w, h, mips, blocks)
        assert rc == 0 and np.array_equal(got, want)
        assert _free_device_bytes() == before


def test_noalloc_decode_matches_allocating_decode():
    L = load_library()
    rng = np.random.default_rng(7)
    # every call is smaller than the one before it, and in another format
    for fmt, w, h, mips in ((abi.FORMAT_BC3, 256, 128, 9), (abi.FORMAT_BC7, 96, 40, 4), (abi.FORMAT_BC7_SRGB, 64, 64, 7),
                            (abi.FORMAT_BC1, 52, 36, 3), (abi.FORMAT_BC4, 8, 8, 4)):
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
