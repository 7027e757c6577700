"""Shared scene builders for the tests."""
import math

import numpy as np

from renderer_amd import abi, synth
from renderer_amd.scene import Camera, Scene, mat4_trs

H_45 = 1.0 / math.tan(math.radians(22.5))  # proj[1][1] of the reference's 45 degree projection


def world_from_fb(x_fb, y_fb, z, width, height):
    """World position (camera at the origin looking down +z, i.e. Camera.direction = (0,0,-1))
    that projects to framebuffer position (x_fb, y_fb), y down, at view depth z.
    clip = (-w x, h y, near, z) with w = h / aspect (renderer/src/renderer.rs:1162-1167)."""
    w = H_45 / (width / height)
    ndc_x = x_fb / (width / 2) - 1.0
    ndc_y = 1.0 - y_fb / (height / 2)
    return (-ndc_x * z / w, ndc_y * z / H_45, z)


def fb_triangle_scene(tris_fb, width, height, alpha_mode=0, material=None, image=None):
    """One mesh, identity model, made of triangles given as ((x,y,z),)*3 in framebuffer coordinates."""
    pts = [[world_from_fb(x, y, z, width, height) for (x, y, z) in tri] for tri in tris_fb]
    sc = synth.make_triangle_scene(pts[0], Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0)), alpha_mode, pts[1:])
    if image is not None:
        sc.add_image(image)
    if material is not None:
        sc.materials = sc.materials[:0]
        sc.add_material(**material)
    return sc


def covered(vis):
    return np.asarray(vis) != np.uint64(abi.VIS_CLEAR)


def key_id(vis):
    return (np.asarray(vis) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def psnr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = ((a - b) ** 2).mean()
    return 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)
