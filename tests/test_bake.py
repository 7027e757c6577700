"""IBL bake (bake/src/brdf.rs, brdf.comp.glsl, filter_envmap.frag.glsl): known properties of the
CPU restatement, and the HIP kernels against it."""
import numpy as np
import pytest

from renderer_amd import synth
from renderer_amd.scene import Cubemap


def constant_sky(rgb, n=8, mips=4):
    levels = []
    for l in range(mips):
        s = max(1, n >> l)
        f = np.zeros((6, s, s, 4), np.float32)
        f[..., :3] = rgb
        f[..., 3] = 1.0
        levels.append(f)
    return Cubemap(levels)


def test_oracle_brdf_lut_properties(oracle):
    size, samples = 16, 512
    lut = oracle.bake_brdf_lut(size, samples)
    assert lut.shape == (size, size, 4) and np.isfinite(lut).all()
    assert (lut[..., 3] == 0).all()                       # the shader never writes alpha
    a, b, c = lut[..., 0], lut[..., 1], lut[..., 2]
    assert (a >= 0).all() and (b >= 0).all() and (c >= 0).all()
    # split sum: scale + bias is the directional albedo of a white GGX lobe, at most ~1
    assert (a + b).max() <= 1.05
    # smooth surfaces (last rows: roughness -> 0) seen head-on (last columns) reflect almost everything: A -> 1, B -> 0
    assert a[-1, -1] > 0.95 and b[-1, -1] < 0.02
    # the Fresnel bias grows towards grazing angles (first columns) on smooth surfaces
    assert b[-1, 0] > b[-1, -1] + 0.2
    # more samples move the table only a little (the estimator has converged to ~1e-2)
    ref = oracle.bake_brdf_lut(size, 4 * samples)
    assert np.abs(ref - lut).max() < 0.03


@pytest.mark.parametrize("distribution", [0, 1, 2])
def test_oracle_prefilter_of_a_constant_sky_is_that_constant(oracle, distribution):
    """Every sample of a uniform environment has the same colour, and the estimator is a weighted
    mean (GGX / Charlie: sum c n.l / sum n.l, Lambertian: mean), so each texel returns it."""
    rgb = np.array([0.25, 0.5, 2.0], np.float32)
    levels = oracle.bake_env_map(constant_sky(rgb), distribution, 8, 3, 64)
    assert [l.shape for l in levels] == [(6, 8, 8, 4), (6, 4, 4, 4), (6, 2, 2, 4)]
    for k, l in enumerate(levels):
        assert (l[..., 3] == 1.0).all()
        if distribution == 2 and k == 0:
            # Charlie at roughness 0: sin_theta = xi^0 = 1, every half vector is perpendicular to n, the reflected
            # ray points along -n, n.l < 0 for all samples and the shader's fallback divides zero by the sample count
            assert (l[..., :3] == 0).all()
        else:
            assert np.allclose(l[..., :3], rgb, rtol=2e-5, atol=0)


def test_oracle_prefilter_blurs_with_roughness(oracle):
    """GGX level 0 (roughness 0) is the sky itself seen along the normal; higher levels average
    more of it, so their contrast across texels drops."""
    sky = synth.default_env().skybox
    levels = oracle.bake_env_map(sky, 1, 16, 4, 128)
    contrast = [float(l[..., :3].std()) for l in levels]
    assert contrast[0] > contrast[1] > contrast[2] > contrast[3] > 0
    # roughness 0: every sample reflects straight back along n -> the source's LOD-0 value in that direction
    from renderer_amd.bake import cubemap_desc  # noqa: F401  (desc helper shared with the GPU wrapper)
    n = 16
    j, i, face = 5, 9, 4
    u, v = (i + 0.5) / n * 2 - 1, (1 - (j + 0.5) / n) * 2 - 1
    d = np.array([u, v, 1.0]); d /= np.linalg.norm(d)
    src = sky.mips[0]
    sn = src.shape[1]
    sc, tc, ma = d[0], -d[1], d[2]                       # +Z face
    x, y = 0.5 * (sc / ma + 1) * sn - 0.5, 0.5 * (tc / ma + 1) * sn - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    ax, ay = x - x0, y - y0
    t = src[4]
    want = (t[y0, x0] * (1 - ax) + t[y0, x0 + 1] * ax) * (1 - ay) + (t[y0 + 1, x0] * (1 - ax) + t[y0 + 1, x0 + 1] * ax) * ay
    assert np.allclose(levels[0][face, j, i, :3], want[:3], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_brdf_lut_matches_oracle(oracle):
    from renderer_amd.bake import bake_brdf_lut
    size, samples = 32, 1000
    ref = oracle.bake_brdf_lut(size, samples)
    got = bake_brdf_lut(size, samples).mips[0]
    assert got.shape == ref.shape
    # same sample set, same order of accumulation; sin/cos/pow differ by ulps between libms
    assert np.abs(got - ref).max() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("distribution", [0, 1, 2])
def test_hip_env_map_matches_oracle(oracle, distribution):
    from renderer_amd.bake import bake_env_map
    sky = synth.default_env().skybox
    size, mips, samples = 16, 4, 96
    ref = oracle.bake_env_map(sky, distribution, size, mips, samples)
    got = bake_env_map(sky, distribution, size, mips, samples).mips
    assert len(got) == mips
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 1e-3 * scale


@pytest.mark.gpu
def test_baked_environment_renders(oracle):
    """The baked maps plug into the scene description and the frame still matches the oracle."""
    from renderer_amd import Renderer, RendererConfig
    from renderer_amd.bake import bake_env
    sc = synth.make_atrium()
    sc.env = bake_env(sc.env.skybox, size=32, samples=64, lut_size=32, lut_samples=256)
    w, h = 160, 90
    conf = RendererConfig(skybox=True)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    o.render_visbuffer()
    ref = o.shade()
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    g.render()
    assert np.abs(g.read_color().astype(np.int32) - ref.astype(np.int32)).max() <= 2


def _write_hdr(path, img, rle):
    """Radiance RGBE writer for the reader's test (shared exponent per pixel)."""
    h, w = img.shape[:2]
    m = img[..., :3].max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0).astype(np.int32)
    mant = np.where(m[..., None] > 1e-32, img[..., :3] / np.ldexp(1.0, e)[..., None] * 256.0, 0.0)
    rgbe = np.concatenate([np.clip(mant, 0, 255).astype(np.uint8), np.where(m > 1e-32, e + 128, 0).astype(np.uint8)[..., None]], -1)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            if not rle:
                f.write(rgbe[y].tobytes())
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row, x = rgbe[y, :, c], 0
                while x < w:  # runs of equal bytes as runs, the rest as literals of up to 128
                    n = 1
                    while x + n < w and n < 127 and row[x + n] == row[x]:
                        n += 1
                    if n >= 3:
                        f.write(bytes([128 + n, row[x]]))
                    else:
                        n = 1
                        while x + n < w and n < 128 and not (x + n + 2 < w and row[x + n] == row[x + n + 1] == row[x + n + 2]):
                            n += 1
                        f.write(bytes([n]) + row[x:x + n].tobytes())
                    x += n
    return rgbe


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr_reader(tmp_path, rle):
    from renderer_amd.bake import read_radiance_hdr
    rng = np.random.default_rng(3)
    img = (rng.random((6, 40, 3)) * np.array([4.0, 1.0, 0.01])).astype(np.float32)
    img[2, 5:30] = img[2, 5]          # a long run
    img[4] = 0.0                       # black: exponent byte 0
    rgbe = _write_hdr(tmp_path / "s.hdr", img, rle)
    got = read_radiance_hdr(tmp_path / "s.hdr")
    want = rgbe[..., :3].astype(np.float32) * np.where(rgbe[..., 3:] > 0, np.ldexp(1.0, rgbe[..., 3:].astype(np.int32) - 136), 0.0)
    assert got.shape == (6, 40, 4) and np.array_equal(got[..., :3], want.astype(np.float32)) and (got[..., 3] == 1).all()
    assert np.abs(got[..., :3] - img).max() <= np.abs(img).max() / 128


def test_equirect_to_cubemap_follows_the_reference_mapping():
    """Spot values against the reference's formulas written out by hand (bake/src/cubemaps.rs:9-118):
    cubemap face 4 (+Z) is the centre of the vertical cross, i.e. cross face 2 = direction
    (1, a - 1, 1 - b): its centre looks along +x, which the equirect mapping puts at u = width / 2
    (theta = 0), v = height / 2 (phi = 0)."""
    from renderer_amd.bake import equirect_to_cubemap
    w, h = 64, 32
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    img = np.stack([u / w, v / h, np.full_like(u, 0.5, dtype=float), np.ones_like(u, dtype=float)], -1).astype(np.float32)
    cm = equirect_to_cubemap(img)
    fs = w // 4
    assert len(cm.mips) == int(np.log2(fs)) and cm.mips[0].shape == (6, fs, fs, 4) and cm.mips[0].dtype == np.uint8
    assert [m.shape[1] for m in cm.mips] == [16, 8, 4, 2]
    centre = cm.mips[0][4][fs // 2, fs // 2].astype(int)
    assert abs(centre[0] - int(0.5 * 255)) <= 2 and abs(centre[1] - int(0.5 * 255)) <= 2 and centre[2] == 127 and centre[3] == 255
    # cubemap face 2 (+Y) is the top of the cross = cross face 4 = (b - 1, a - 1, 1): straight along +z = the picture's top row
    top = cm.mips[0][2][fs // 2, fs // 2].astype(int)
    assert top[1] <= 4
    # cubemap face 3 (-Y) = cross face 5 = (1 - b, a - 1, -1): the bottom row
    assert cm.mips[0][3][fs // 2, fs // 2, 1] >= 240
    # a constant picture gives a constant cubemap on every level, values truncated not rounded
    flat = equirect_to_cubemap(np.full((32, 64, 4), 0.999 * 200 / 255, np.float32))
    assert all((m[..., :3] == 199).all() for m in flat.mips)


@pytest.mark.gpu
def test_bake_command_line_feeds_the_renderer(oracle, tmp_path):
    """sky picture -> `python -m renderer_amd.bake` -> EnvMap JSON + KTX2 files -> loaded back -> a frame
    rendered with it matches the oracle rendering with the same files."""
    from renderer_amd import Renderer, RendererConfig
    from renderer_amd.bake import main as bake_main
    from renderer_amd.ktx2 import load_env_json
    w, h = 128, 64
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    sky = np.stack([0.2 + 0.6 * u / w, 0.3 + 0.5 * (1 - v / h), 0.9 - 0.5 * v / h, np.ones_like(u, dtype=float)], -1).astype(np.float32)
    np.save(tmp_path / "sky.npy", sky)
    assert bake_main([str(tmp_path / "sky.npy"), str(tmp_path / "env.json"), "--size", "16", "--samples", "32",
                      "--lut-size", "16", "--lut-samples", "64"]) == 0
    env = load_env_json(tmp_path / "env.json")
    assert env.skybox.mips[0].shape == (6, 32, 32, 4) and env.ggx.mips[0].shape == (6, 16, 16, 4) and len(env.ggx.mips) == 4
    sc = synth.make_atrium()
    sc.env = env
    conf = RendererConfig(skybox=True)
    o = oracle.OracleRenderer(160, 90, conf)
    o.set_scene(sc)
    o.render_visbuffer()
    ref = o.shade()
    g = Renderer(160, 90, conf)
    g.set_scene(sc)
    g.render()
    assert np.abs(g.read_color().astype(np.int32) - ref.astype(np.int32)).max() <= 2
    g.close()
