"""The fp32 cube map (rt_upload_skybox_f32): an extension of the build, the reference's sky being an 8-bit sRGB image like
every other texture. The oracle's restatement of the cube fetch (main.comp:82-84) reads the same float texels, so the fetch
is compared bit for bit and rendered images as for the 8-bit sky. Parity of the extension itself is unpinned by
construction: the reference has nothing to compare it with."""
import numpy as np
import pytest

from util import rel_close


def test_radiance_hdr_round_trip_and_panorama_resampling(tmp_path):
    from ray_tracer_amd import loaders
    rng = np.random.default_rng(5)
    img = rng.uniform(0.0, 80.0, size=(24, 48, 3)).astype(np.float32)
    img[3, 4] = 0.0
    img[5, 6] = (1e-3, 2e-3, 5e-4)
    loaders.save_hdr(tmp_path / "a.hdr", img)
    back = loaders.load_hdr(tmp_path / "a.hdr")
    assert back.shape == img.shape and back.dtype == np.float32
    # shared-exponent format: every channel within 1/128 of the pixel's largest channel
    assert (np.abs(back - img) <= img.max(axis=-1, keepdims=True) / 128 + 1e-30).all()
    assert (back[3, 4] == 0).all()
    # run-length encoded scanlines, written by hand: a run of 40 equal bytes + 8 literal ones per channel
    w, h = 48, 2
    rows = b""
    for _ in range(h):
        row = bytes([2, 2, 0, w])
        for ch, v in enumerate((100, 50, 25, 130)):
            row += bytes([128 + 40, v]) + bytes([8]) + bytes([v + 1] * 8)
        rows += row
    (tmp_path / "rle.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + rows)
    rle = loaders.load_hdr(tmp_path / "rle.hdr")
    assert np.allclose(rle[0, 0], np.array([100, 50, 25]) * 2.0 ** (130 - 136)) and np.allclose(rle[1, 47], np.array([101, 51, 26]) * 2.0 ** (131 - 136))
    with pytest.raises(loaders.SceneFormatError):
        (tmp_path / "bad.hdr").write_bytes(b"P6\n1 1\n255\n...")
        loaders.load_hdr(tmp_path / "bad.hdr")
    # a constant panorama gives a constant cube; a panorama that is bright above the horizon lights +Y and not -Y
    cube = loaders.equirect_to_cube(np.full((16, 32, 3), 2.5, np.float32), 8)
    assert cube.shape == (6, 8, 8, 4) and np.allclose(cube[..., :3], 2.5) and (cube[..., 3] == 1).all()
    pano = np.zeros((32, 64, 3), np.float32)
    pano[:8] = 10.0                                           # the top quarter of the latitudes: above 45 degrees
    cube = loaders.equirect_to_cube(pano, 16)
    assert cube[2, 4:12, 4:12, :3].min() == 10.0 and cube[3, ..., :3].max() == 0.0
    assert 0.0 < cube[4, ..., 0].mean() < 10.0                # a side face sees both
    # centre of the panorama = -z
    pano = np.zeros((32, 64, 3), np.float32)
    pano[12:20, 28:36] = 7.0
    cube = loaders.equirect_to_cube(pano, 16)
    assert cube[5, 6:10, 6:10, 0].min() > 0 and cube[4, ..., 0].max() == 0


def test_yaml_scene_with_hdr_panorama(tmp_path):
    from ray_tracer_amd import loaders
    pano = np.zeros((16, 32, 3), np.float32)
    pano[:8] = (3.0, 4.0, 30.0)
    loaders.save_hdr(tmp_path / "sky.hdr", pano)
    (tmp_path / "s.yaml").write_text(
        "skybox_hdr: sky.hdr\nskybox_hdr_size: 8\nmaterials: {m: {base_colour: [200, 200, 200]}}\n"
        "surfaces:\n  - {type: mesh, material: m, vertices: [[-1, 0, -1], [1, 0, -1], [0, 0, 1]], triangles: [[0, 1, 2]]}\n"
        "camera: {resolution: [32, 18], location: [0, 2, 4], look_at: [0, 0, 0], focal_length: 1.0, focus_distance: 1.0, aperture_radius: 0.0}\n")
    sc = loaders.load_scene_yaml(tmp_path / "s.yaml")
    assert sc.skybox_hdr is not None and sc.skybox_hdr.shape == (6, 8, 8, 4) and sc.skybox_hdr.dtype == np.float32
    assert np.allclose(sc.skybox_hdr[2, 2:6, 2:6, :3], (3.0, 4.0, 30.0), rtol=1e-2)


@pytest.mark.gpu
def test_hdr_cube_fetch_bit_exact_and_replaced_by_the_next_texture_upload(rt, oracle, scene_cache):
    sc = scene_cache("car", scale=0.03, width=96, height=54, hdr_sky=True)
    assert sc.skybox_hdr.max() > 30.0                      # a sun no 8-bit texel holds
    r = rt.HipRenderer.create(sc, max_bounces=4)
    o = oracle.OracleScene(sc)
    rng = np.random.default_rng(23)
    d = np.concatenate([rng.normal(size=(6000, 3)), [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 1], [-1, 1, -1],
                                                     [0, 0, 0], [np.nan, 1, 0]]]).astype(np.float32)
    g, h = r.sample_texture(0xFFFFFFFF, d), o.sample_texture(0xFFFFFFFF, d)
    assert np.array_equal(g, h, equal_nan=True)
    assert g[:6000, :3].max() > 5.0
    # towards the sun: the texel value itself
    up = r.sample_texture(0xFFFFFFFF, np.array([[0.1, 1.0, -0.1]], np.float32))
    assert up[0, 0] > 1.0
    # the 2-D textures are untouched by the fp32 sky
    uv = rng.uniform(-2, 2, size=(2000, 3)).astype(np.float32)
    assert np.array_equal(r.sample_texture(0, uv), o.sample_texture(0, uv))
    # rt_upload_textures installs an 8-bit sky again
    plain = scene_cache("car", scale=0.03, width=96, height=54)
    r.upload_scene(plain)
    o8 = oracle.OracleScene(plain)
    assert np.array_equal(r.sample_texture(0xFFFFFFFF, d[:6008]), o8.sample_texture(0xFFFFFFFF, d[:6008]))
    assert r.sample_texture(0xFFFFFFFF, d[:6000])[:, :3].max() <= 1.0
    r.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact_arith", "default"])
def test_render_under_an_hdr_sky_matches_the_oracle(rt, oracle, scene_cache, mode):
    """The car scene has no lights: everything it shows is sky, now with radiance up to 60. Per-pixel agreement of single
    samples, and a converged pair within the RMSE the 8-bit sky is held to."""
    sc = scene_cache("car", spatial=True, scale=0.05, width=192, height=108, hdr_sky=True)
    r = rt.HipRenderer.create(sc, max_bounces=8, flags=rt.api.RT_EXACT_ARITH if mode == "exact_arith" else 0)
    o = oracle.OracleScene(sc)
    g = r.render_radiance(1)
    h, _ = o.render_radiance(1, 8)
    close = rel_close(g, h, 1e-3, 1e-4).all(axis=-1)
    assert close.mean() > 0.97, close.mean()
    assert h.max() > 5.0                                   # the sun is seen (before the firefly clamp)
    spp = 128
    r.render(spp)
    gm = r.read_accum()[..., :3]
    want, _, _ = o.render(spp, 8)
    wm = want[..., :3] / want[..., 3:4]
    rmse = float(np.sqrt(np.mean((gm - wm) ** 2)))
    mean = float(wm.mean())
    print(f"{sc.name} under the fp32 sky, 192x108 {spp} spp ({mode}): RMSE vs oracle {rmse:.2e} (mean radiance {mean:.4f}, relative {rmse / mean:.2e})")
    assert rmse < 2.0e-2 * mean
    assert abs(float(gm.mean()) - mean) < 2e-3 * mean
    r.destroy()
