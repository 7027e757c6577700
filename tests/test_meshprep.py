"""Host-side mesh preparation (the counterpart of Geometry::append_mesh, scenery/src/mesh.rs:197-374):
structural invariants of the meshlets, of the LOD chain and of the simplifier.  The reference
holds no fixtures for this step and meshoptimizer is not in the tree, so these are properties the
reference's own rules imply, not meshoptimizer's exact output."""
import numpy as np
import pytest

from renderer_amd import abi, synth
from renderer_amd.meshprep import build_mesh
from renderer_amd.renderer import RendererError
from renderer_amd.scene import Scene, mat4_trs


def _tri_keys(vertices, tris):
    """Triangles as order-independent sets of position triples."""
    p = vertices["p"][:, :3][np.asarray(tris, np.int64).reshape(-1, 3)]  # (n, 3, 3)
    rows = [tuple(sorted(map(tuple, t.tolist()))) for t in p]
    return sorted(rows)


def _edge_uses(tris):
    e = np.sort(np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]).astype(np.int64), axis=1)
    _, counts = np.unique(e[:, 0] * (1 << 32) + e[:, 1], return_counts=True)
    return counts


@pytest.fixture(scope="module")
def sphere():
    v, lods, _ = synth.icosphere(4, lods=1)
    return v, lods[0], build_mesh(v, lods[0])


def test_level0_keeps_every_triangle_and_respects_the_meshlet_limits(sphere):
    v, idx, b = sphere
    assert _tri_keys(b.vertices, b.lod_triangles(0)) == _tri_keys(v, idx)
    assert _tri_keys(b.vertices, b.indices) == _tri_keys(v, idx)
    ml = b.meshlets
    assert ml["vertex_count"].max() <= 64 and ml["triangle_count"].max() <= 124 and ml["triangle_count"].min() >= 1
    assert (ml["triangle_offset"] % 4 == 0).all()  # each meshlet's triangle bytes start on a 4-byte boundary
    m = b.mesh
    assert int(m["vertex_count"]) == len(b.vertices) and int(m["lods"][0]["meshlet_offset"]) == 0 and float(m["lods"][0]["error"]) == 0.0
    # mesh bounds: mean of the vertices, farthest vertex (mesh.rs:234-243)
    p = b.vertices["p"][:, :3]
    assert np.allclose(m["center"], p.mean(0), atol=1e-5)
    assert np.isclose(float(m["radius"]), np.linalg.norm(p - m["center"], axis=1).max(), rtol=1e-5)


def test_meshlet_bounds_hold_their_vertices_and_normals(sphere):
    _, _, b = sphere
    p = b.vertices["p"][:, :3].astype(np.float64)
    for m in b.meshlets:
        vo, to, nv, nt = int(m["vertex_offset"]), int(m["triangle_offset"]), int(m["vertex_count"]), int(m["triangle_count"])
        ids = b.meshlet_vertices[vo:vo + nv]
        assert (np.linalg.norm(p[ids] - m["center"], axis=1) <= float(m["radius"]) * (1 + 1e-6)).all()
        local = b.meshlet_triangles[to:to + nt * 3].reshape(-1, 3)
        assert local.max() < nv
        t = p[ids[local.astype(np.int64)]]
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        cutoff = int(m["cone_cutoff"])
        if cutoff < 127:  # a usable cone: every normal within it, also after the s8 quantisation
            axis = m["cone_axis"].astype(np.float64) / 127.0
            mindp = (n @ axis).min() / max(np.linalg.norm(axis), 1e-12)
            assert np.sqrt(max(0.0, 1.0 - mindp * mindp)) <= cutoff / 127.0 + 1e-6


def test_lod_chain_follows_the_reference_rules(sphere):
    v, idx, b = sphere
    m = b.mesh
    n_lods = int(m["lod_count"])
    assert 2 <= n_lods <= 8
    prev_err, n0 = 0.0, len(idx)
    target = n0 * 0.65
    for li in range(1, n_lods):
        tris = b.lod_triangles(li)
        assert 0 < tris.size < n0 * 0.95          # the chain stops before a level this large (mesh.rs:301)
        assert tris.size <= target + 3 * 64       # reaches the x0.65 target up to one pass's granularity
        assert (_edge_uses(tris) == 2).all()      # a closed surface stays closed
        err = float(m["lods"][li]["error"])
        assert err >= prev_err > -1.0 and err > 0.0
        # no new vertices: a level only uses the mesh's own vertex buffer, and it stays near the unit sphere
        c = b.vertices["p"][:, :3][tris.astype(np.int64)].mean(axis=1)
        assert (1.0 - np.linalg.norm(c, axis=1)).max() <= 0.1 * 2.0  # the relative limit 0.1 of the extent 2
        prev_err, target = err, int(target) * 0.65
    # consecutive levels occupy consecutive meshlet ranges
    ends = [int(m["lods"][i]["meshlet_offset"]) + int(m["lods"][i]["meshlet_count"]) for i in range(n_lods)]
    assert ends[-1] == len(b.meshlets) and all(int(m["lods"][i + 1]["meshlet_offset"]) == ends[i] for i in range(n_lods - 1))


def test_duplicate_vertices_are_merged_and_unused_ones_dropped():
    v, lods, _ = synth.icosphere(2, lods=1)
    idx = lods[0]
    soup = v[idx]                                    # three private vertices per triangle
    extra = np.concatenate([soup, v[:5]])            # plus five that nothing references
    b = build_mesh(extra, np.arange(len(soup), dtype=np.uint32), max_lods=1)
    assert len(b.vertices) == len(v) and int(b.mesh["lod_count"]) == 1
    assert _tri_keys(b.vertices, b.indices) == _tri_keys(v, idx)


def test_open_surface_keeps_its_border():
    corners = np.array([[0, 0, 0], [4, 0, 0], [4, 0, 2], [0, 0, 2]], np.float32)
    v, idx = synth.grid_mesh(corners, 32, 16)
    b = build_mesh(v, idx)
    assert int(b.mesh["lod_count"]) >= 3
    p = b.vertices["p"][:, :3].astype(np.float64)
    for li in range(int(b.mesh["lod_count"])):
        t = p[b.lod_triangles(li).astype(np.int64)]
        area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum()
        assert np.isclose(area, 8.0, rtol=1e-5)      # a flat sheet simplifies without losing area: the border did not move in
        assert (np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])[:, 1] != 0).all()  # no degenerate triangles
    # flat: the reported errors are (near) zero
    assert max(float(b.mesh["lods"][i]["error"]) for i in range(int(b.mesh["lod_count"]))) < 1e-4


def test_errors():
    v, lods, _ = synth.icosphere(1, lods=1)
    with pytest.raises(RendererError, match="multiple of 3"):
        build_mesh(v, lods[0][:-1])
    with pytest.raises(RendererError, match="out of range"):
        build_mesh(v, np.array([0, 1, len(v)], np.uint32))
    with pytest.raises(RendererError, match="max_lods"):
        build_mesh(v, lods[0], max_lods=9)


def test_scene_append_mesh_auto_offsets_compose(oracle):
    """Two prepared meshes in one scene: offsets are rebased, and the CPU checker renders LOD 0 of
    the prepared mesh exactly like the plain greedy meshlets of the same triangles (same triangles,
    different meshlets: every pixel shows the same draw at the same depth)."""
    from renderer_amd import RendererConfig
    v, lods, _ = synth.icosphere(3, lods=1)
    a, b = Scene(), Scene()
    for sc, auto in ((a, True), (b, False)):
        for k in range(2):
            mi = sc.append_mesh_auto(v, lods[0]) if auto else sc.append_mesh(v, [lods[0]])
            sc.add_draw(mi, mat4_trs((2.5 * k - 1.2, 0.0, 0.0)), 0xFFFF)
        sc.env = synth.default_env()
        sc.camera.position, sc.camera.direction = (0.0, 0.5, 5.0), (0.0, 0.1, 1.0)
    assert int(a.meshes[1]["vertex_offset"]) == int(a.meshes[0]["vertex_count"])
    assert int(a.meshes[1]["lods"][0]["meshlet_offset"]) == sum(int(L["meshlet_count"]) for L in a.meshes[0]["lods"][:int(a.meshes[0]["lod_count"])])
    out = []
    for sc in (a, b):
        o = oracle.OracleRenderer(160, 90, RendererConfig(lod_enabled=False))
        o.set_scene(sc)
        out.append(o.render_visbuffer().copy() >> np.uint64(32))  # depth half of the key
    assert np.array_equal(out[0], out[1])


def _fifo_misses(indices, cache=16):
    """Vertex transforms a FIFO post-transform cache of ``cache`` entries would do for this index order."""
    fifo, misses = [], 0
    for v in np.asarray(indices).reshape(-1).tolist():
        if v not in fifo:
            misses += 1
            fifo.append(v)
            if len(fifo) > cache:
                fifo.pop(0)
    return misses


def test_fast_mode_orders_for_a_fifo_cache_and_keeps_the_mesh(sphere):
    """RendererConfig::fast_mode (scenery/src/lib.rs:11, mesh.rs:206-210,305-309): the second ordering pass
    targets a 16-entry FIFO cache.  Same triangles, same limits, every level still a valid chain, and the
    order does what it is for: far fewer FIFO misses than the unordered list."""
    v, idx, normal = sphere
    rng = np.random.default_rng(0)
    shuffled = np.asarray(idx).reshape(-1, 3)[rng.permutation(len(idx) // 3)].reshape(-1)
    fast = build_mesh(v, shuffled, fast_mode=True)
    assert _tri_keys(fast.vertices, fast.indices) == _tri_keys(v, idx)
    assert _tri_keys(fast.vertices, fast.lod_triangles(0)) == _tri_keys(v, idx)
    assert fast.meshlets["vertex_count"].max() <= 64 and fast.meshlets["triangle_count"].max() <= 124
    assert int(fast.mesh["lod_count"]) == int(normal.mesh["lod_count"])
    assert not np.array_equal(fast.indices, normal.indices)
    n_tri = len(idx) // 3
    assert _fifo_misses(fast.indices) < 0.8 * n_tri  # a good order needs about 0.6 - 0.7 transforms per triangle
    assert _fifo_misses(fast.indices) < 0.5 * _fifo_misses(shuffled)
    for li in range(1, int(fast.mesh["lod_count"])):
        t = fast.lod_triangles(li)
        assert len(t) and t.max() < len(fast.vertices)
