"""Tangent generation (``rdr_generate_tangents``, the stand-in for the mikktspace crate the reference calls at
scenery/src/scene.rs:207): hand-derived cases.  Host code in the library, no GPU needed."""
import math

import numpy as np
import pytest

from renderer_amd import abi
from renderer_amd.meshprep import generate_tangents


def mesh(points, uvs, tris, normal=(0.0, 0.0, 1.0)):
    v = np.zeros(len(points), abi.VERTEX)
    for k, (p, uv) in enumerate(zip(points, uvs)):
        v[k]["p"] = (*p, 0.0)
        v[k]["n"] = (*normal, 0.0)
        v[k]["uv"] = (*uv, 0.0, 0.0)
        v[k]["t"] = (9.0, 9.0, 9.0, 9.0)
    return v, np.asarray(tris, np.uint32).reshape(-1)


def test_quad_with_plain_uv_gets_plus_x_and_positive_handedness():
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], [(0, 0), (1, 0), (1, 1), (0, 1)], [0, 1, 2, 0, 2, 3])
    generate_tangents(v, idx)
    assert np.allclose(v["t"], [[1, 0, 0, 1]] * 4, atol=1e-6)


def test_mirrored_uv_flips_tangent_and_handedness():
    # u runs against x: dP/du = -x, and u x v is opposite to the surface orientation
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], [(1, 0), (0, 0), (0, 1), (1, 1)], [0, 1, 2, 0, 2, 3])
    generate_tangents(v, idx)
    assert np.allclose(v["t"], [[-1, 0, 0, -1]] * 4, atol=1e-6)


def test_rotated_uv_and_scaled_uv_only_change_direction():
    # u = 3 y, v = -2 x: the tangent is +y whatever the scale; orientation preserved
    pts = [(0, 0, 0), (2, 0, 0), (2, 1, 0), (0, 1, 0)]
    v, idx = mesh(pts, [(3 * y, -2 * x) for x, y, _ in pts], [0, 1, 2, 0, 2, 3])
    generate_tangents(v, idx)
    assert np.allclose(v["t"], [[0, 1, 0, 1]] * 4, atol=1e-6)


def test_tangent_is_projected_into_the_plane_of_the_vertex_normal():
    n = np.array([0.0, 0.6, 0.8])
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], [(0, 0), (0, 1), (-1, 1), (-1, 0)], [0, 1, 2, 0, 2, 3], normal=tuple(n))
    generate_tangents(v, idx)  # u = -y: dP/du = (0, -1, 0), minus its component along n, normalised
    t = np.array([0.0, -1.0, 0.0]) - n * (n @ np.array([0.0, -1.0, 0.0]))
    t /= np.linalg.norm(t)
    assert np.allclose(v["t"][:, :3], [t] * 4, atol=1e-6) and np.allclose(v["t"][:, 3], 1.0)


def test_angle_weighting_at_a_vertex_shared_by_two_triangles():
    """A = origin is shared, through the edge A-C, by T1 (90 degrees at A, dP/du = +x) and T2 (45 degrees at A,
    dP/du = (1, 1, 0) / sqrt 2): the tangent at A is normalize(pi/2 * (1,0,0) + pi/4 * (1,1,0)/sqrt 2)."""
    pts = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (-1, 1, 0)]
    uvs = [(0, 0), (1, 0), (0, 1), (-1, 2)]
    v, idx = mesh(pts, uvs, [0, 1, 2, 0, 2, 3])
    generate_tangents(v, idx)
    want = (math.pi / 2) * np.array([1.0, 0, 0]) + (math.pi / 4) * np.array([1.0, 1.0, 0]) / math.sqrt(2)
    want /= np.linalg.norm(want)
    assert np.allclose(v["t"][0], [*want, 1.0], atol=1e-6)
    assert np.allclose(v["t"][1], [1, 0, 0, 1], atol=1e-6)          # B: only T1
    d = np.array([1.0, 1.0, 0]) / math.sqrt(2)
    assert np.allclose(v["t"][3], [*d, 1.0], atol=1e-6)             # D: only T2
    # C: T1 has 45 degrees there, T2 has 90 degrees
    want_c = (math.pi / 4) * np.array([1.0, 0, 0]) + (math.pi / 2) * d
    want_c /= np.linalg.norm(want_c)
    assert np.allclose(v["t"][2], [*want_c, 1.0], atol=1e-6)


def test_triangles_of_opposite_uv_orientation_do_not_share_a_group():
    """Two quads side by side, the right one with mirrored u: the shared edge's vertices belong to one group per
    side; the record is shared, so the side written last (triangle order) wins, as in the reference."""
    pts = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (2, 0, 0), (2, 1, 0)]
    uvs = [(0, 0), (1, 0), (1, 1), (0, 1), (0, 0), (0, 1)]  # right quad: u decreases with x
    v, idx = mesh(pts, uvs, [0, 1, 2, 0, 2, 3, 1, 4, 5, 1, 5, 2])
    generate_tangents(v, idx)
    assert np.allclose(v["t"][[0, 3]], [[1, 0, 0, 1]] * 2, atol=1e-6)
    assert np.allclose(v["t"][[4, 5]], [[-1, 0, 0, -1]] * 2, atol=1e-6)
    assert np.allclose(v["t"][[1, 2]], [[-1, 0, 0, -1]] * 2, atol=1e-6)  # shared records: the right quad came last


def test_degenerate_uv_and_degenerate_triangles():
    # constant uv: no direction can be derived; such a triangle alone keeps the default (1, 0, 0) with w = -1
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0.5, 0.5)] * 3, [0, 1, 2])
    generate_tangents(v, idx)
    assert np.allclose(v["t"], [[1, 0, 0, -1]] * 3)
    # a triangle with two equal positions borrows from the good triangle that shares its vertices
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 0)], [(0, 0), (0, 1), (-1, 0), (0, 0)], [0, 1, 2, 0, 3, 1])
    generate_tangents(v, idx)  # u = -y
    assert np.allclose(v["t"][:3], [[0, -1, 0, 1]] * 3, atol=1e-6)
    assert np.allclose(v["t"][3], [0, -1, 0, 1], atol=1e-6)


def test_errors():
    from renderer_amd.renderer import RendererError
    v, idx = mesh([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0, 0), (1, 0), (0, 1)], [0, 1, 7])
    with pytest.raises(RendererError, match="out of range"):
        generate_tangents(v, idx)


def test_sphere_tangents_are_unit_and_orthogonal_to_the_normal():
    from renderer_amd import synth
    v, lods, _ = synth.icosphere(3, lods=1)
    v = np.ascontiguousarray(v)
    generate_tangents(v, lods[0])
    t, n = v["t"][:, :3].astype(np.float64), v["n"][:, :3].astype(np.float64)
    good = np.abs(np.linalg.norm(t, axis=1) - 1.0) < 1e-4
    assert good.mean() > 0.98  # the poles of a latitude / longitude parametrisation may be degenerate
    assert np.abs((t[good] * n[good]).sum(1)).max() < 1e-4
    assert set(np.unique(v["t"][:, 3])) <= {-1.0, 1.0}
