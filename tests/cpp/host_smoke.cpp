// Builds a two-triangle scene through the C++ facade mirror and, when a GPU is present, renders one sample.
// Exit code 0 = behaved as specified (including the "no GPU: create fails with a message" case).
#include <cstdio>
#include <cstring>

#include "hip_renderer.hpp"

int main() {
    rt::Scene sc;
    rt_material m{};
    m.base_colour[0] = m.base_colour[1] = m.base_colour[2] = 0.8f;
    m.roughness = 0.5f; m.sheen_tint = 0.5f; m.clearcoat_roughness = 0.03f; m.ior = 1.45f; m.shade_smooth = 1;
    m.base_color_tex = m.metallic_roughness_tex = m.normal_tex = m.emission_tex = RT_NO_TEXTURE;
    sc.materials.push_back(m);
    const float quad[2][3][3] = {{{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}}, {{-1, -1, 0}, {1, 1, 0}, {-1, 1, 0}}};
    for (auto& t : quad) {
        rt_triangle tr{};
        for (int v = 0; v < 3; ++v) {
            for (int k = 0; k < 3; ++k) tr.v[v][k] = t[v][k];
            tr.n[v][2] = 1.0f;
        }
        sc.triangles.push_back(tr);
    }
    const float loc[3] = {0, 0, 3}, at[3] = {0, 0, 0};
    rt::check(rt_camera_look_at(&sc.camera, 32, 32, loc, at, 1.0f, 1.0f, 0.0f));
    sc.build_bvh();
    sc.build_lights();
    if (sc.bvh.size() != 1 || sc.bvh[0].amt != 2 || !sc.emissive_tris.empty()) { std::puts("unexpected scene prep result"); return 1; }
    uint64_t sz[6];
    sc.get_buffer_sizes(sz);
    if (sz[0] != 32 || sz[1] != 96 || sz[2] != 288 || sc.write_buffers().size() != 416) { std::puts("unexpected blob sizes"); return 1; }
    try {
        rt::HipRenderer r = rt::HipRenderer::create(sc, 4);
        r.dispatch_compute(2);
        r.frame_complete();
        std::vector<float> img(32 * 32 * 4);
        r.read_accum(img);
        if (r.sample_count() != 2 || !(img[(16 * 32 + 16) * 4] > 0.0f)) { std::puts("render produced nothing"); return 1; }
        const rt_tune_report rep = r.tune_resident_set();            // the root is a leaf: nothing to move, and nothing changes
        std::vector<float> again(32 * 32 * 4);
        r.reset_accumulator();
        r.dispatch_compute(2);
        r.frame_complete();
        r.read_accum(again);
        if (rep.records_moved[0] != 0 || again != img) { std::puts("tuning changed the picture"); return 1; }
        // the two tile slices of the image: each leaves the other's pixels alone, together they are the picture above
        std::vector<float> half0(32 * 32 * 4), half1(32 * 32 * 4);
        r.set_tile_slice(0, 2); r.dispatch_compute(2); r.frame_complete(); r.read_accum(half0);
        r.set_tile_slice(1, 2); r.dispatch_compute(2); r.frame_complete(); r.read_accum(half1);
        r.set_tile_slice(0, 1);
        for (size_t k = 0; k < img.size(); ++k) {
            if ((half0[k] != 0.0f && half1[k] != 0.0f) || half0[k] + half1[k] != img[k]) { std::puts("tile slices do not add up to the picture"); return 1; }
        }
        std::puts("gpu render ok");
    } catch (const rt::Error& e) {
        if (e.code != RT_ERR_DEVICE) { std::printf("unexpected failure: %s\n", e.what()); return 1; }
        std::printf("no gpu: %s\n", e.what());
    }
    return 0;
}
