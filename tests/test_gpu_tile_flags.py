"""The sort-last tile flags (one word per 32 x 32-pixel tile behind the visibility buffer, non-zero when a rasteriser may
have written a key into the tile): the exchange between ranks fetches keys only from ranks whose flag is set, so a key in
an unflagged tile would be lost.  Checked directly here, on one GPU with the flags forced on: every pixel that holds a key
lies in a flagged tile, for small and large triangles, alpha-tested draws, meshlets that reach the near plane, and both
phases of a frame with occlusion culling."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, abi, synth
from renderer_amd.scene import Camera

pytestmark = pytest.mark.gpu
TILE = 32


def tile_flags(g):
    """The flag words, copied out of device memory with the HIP runtime the library itself is linked against."""
    tx, ty = (g.width + TILE - 1) // TILE, (g.height + TILE - 1) // TILE
    ptr = g._L.rdr_visbuffer_device_ptr(g._h) + g.width * g.height * 8
    g.sync()
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(ty * tx, np.uint32)
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2)) == 0  # device to host
    return out.reshape(ty, tx)


def check(g):
    vis = g.read_visbuffer_u64()
    flags = tile_flags(g)
    has_key = vis != np.uint64(abi.VIS_CLEAR)
    ys, xs = np.nonzero(has_key)
    unflagged = flags[ys // TILE, xs // TILE] == 0
    assert not unflagged.any(), f"{int(unflagged.sum())} keys of {len(ys)} lie in tiles without a flag"
    return int(has_key.sum()), int((flags != 0).sum()), flags.size


@pytest.mark.parametrize("occlusion", [False, True])
@pytest.mark.parametrize("scene_name", ["field", "alphafield", "atrium"])
def test_every_key_lies_in_a_flagged_tile(monkeypatch, scene_name, occlusion):
    monkeypatch.setenv("RDR_TILE_FLAGS", "1")
    if scene_name == "field":
        sc = synth.make_field(300, 4, lods=3, with_env=True, prepared=True)
    elif scene_name == "alphafield":
        sc = synth.make_field(120, 3, lods=1, alpha_fraction=0.5, with_env=True, prepared=True)
    else:
        sc = synth.make_atrium(columns=6, spheres=20, sphere_subdiv=3)
    g = Renderer(1000, 563, RendererConfig(lod_enabled=scene_name == "field", cull_enabled=occlusion))
    g.set_scene(sc)
    base = sc.camera
    rng = np.random.default_rng(7)
    seen_sparse = False
    for k in range(6):
        # from the scene's own camera inwards: later views stand among the instances, with meshlets across the near plane
        p = np.asarray(base.position, np.float64) * (1.0 - 0.18 * k) + rng.normal(size=3) * 0.3 * k
        d = np.asarray(base.direction, np.float64) + rng.normal(size=3) * 0.15 * k
        g.render_visbuffer(Camera(tuple(p), tuple(d / np.linalg.norm(d))), (0.3, 0.8, -0.4))
        assert g.frame_status() == 0
        keys, flagged, tiles = check(g)
        seen_sparse = seen_sparse or (keys > 0 and flagged < tiles)
    if scene_name != "atrium":
        assert seen_sparse  # the flags say something: some frame left tiles unflagged
    g.close()
