#!/usr/bin/env python3
"""The rasteriser's own roofline: rates of no-return 64-bit atomicMin (agent scope) into a 66 MB visibility
buffer, for the address patterns of profiles/atomic_peak.hip.  Prints one JSON object; G atomics per second."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    path = os.path.join(HERE, "libatomic_peak.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.atomic_peak_rate.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.atomic_peak_rate.restype = C.c_double
    return L


def rates(device=0, repeats=5, full=True):
    """{'random', 'box_2x2', ...: G atomics/s}.  `box_2x2` (64 boxes of 2 x 2 pixels scattered over 32 x 32 pixels
    per wave, the shape of a meshlet of the bench field) is the figure the bench prices the rasteriser against."""
    L = load()
    if L is None:
        raise RuntimeError("profiles/libatomic_peak.so is missing: make -C profiles")
    out = {"box_2x2_region32": L.atomic_peak_rate(device, 1, 32, 2, 2, repeats)}
    if full:
        out["random"] = L.atomic_peak_rate(device, 0, 0, 0, 0, repeats)
        out["box_1x1_region32"] = L.atomic_peak_rate(device, 1, 32, 1, 1, repeats)
        out["box_4x4_region32"] = L.atomic_peak_rate(device, 1, 32, 4, 4, repeats)
        out["box_2x2_region128"] = L.atomic_peak_rate(device, 1, 128, 2, 2, repeats)
        out["box_4x4_region8"] = L.atomic_peak_rate(device, 1, 8, 4, 4, repeats)
        out["same_address"] = L.atomic_peak_rate(device, 2, 0, 0, 0, repeats)
    return out


if __name__ == "__main__":
    r = rates()
    r["unit"] = "G atomics/s (8 bytes each)"
    json.dump(r, sys.stdout, indent=1)
    print()
