// harness.cpp — command-line host for librdr_hip.so in C++, the counterpart of the reference's
// `potato <config.json> <envmap.json> <scene>` (app/src/main.rs:23-72) for this path:
//
//   rdr_harness <scene.bin> <width> <height> [--lod] [--cull] [--skybox] [--frames N]
//               [--dump-vis vis.u32x2] [--dump-color color.rgba8]
//
// It reads a flat scene dump (every array SceneResources::load_scene uploads, renderer/src/renderer.rs:426-586;
// layout in renderer_amd/scene_bin.py), hands the arrays to rdr::Renderer exactly as the reference hands its
// Scene to Renderer::set_scene, renders, and writes the buffers in the reference's texel formats.  Everything
// goes through renderer.hpp, i.e. through the C ABI; a failure prints the library's message and exits 1.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "renderer.hpp"

namespace {

struct Blob {
    std::vector<uint8_t> bytes;
    size_t at = 0;
    const uint8_t* take(size_t n) {
        if (at + n > bytes.size()) throw std::runtime_error("scene.bin: truncated");
        const uint8_t* p = bytes.data() + at;
        at += n;
        return p;
    }
    uint32_t u32() {
        uint32_t v;
        std::memcpy(&v, take(4), 4);
        return v;
    }
};

uint32_t mip_dim(uint32_t d, uint32_t l) { return (d >> l) ? (d >> l) : 1u; }

size_t image_bytes(uint32_t w, uint32_t h, uint32_t mips, uint32_t format) {
    size_t total = 0;
    for (uint32_t l = 0; l < mips; ++l) {
        const size_t mw = mip_dim(w, l), mh = mip_dim(h, l);
        if (format >= RDR_FORMAT_BC1) total += ((mw + 3) / 4) * ((mh + 3) / 4) * ((format == RDR_FORMAT_BC1 || format == RDR_FORMAT_BC4) ? 8 : 16);
        else total += mw * mh * (format == RDR_FORMAT_RGBA32F ? 16 : 4);
    }
    return total;
}

rdr_image_desc read_image(Blob& b) {
    rdr_image_desc d{};
    d.width = b.u32(); d.height = b.u32(); d.mip_count = b.u32(); d.format = b.u32();
    if (d.width) d.data = b.take(image_bytes(d.width, d.height, d.mip_count, d.format));
    return d;
}

rdr_cubemap_desc read_cube(Blob& b) {
    rdr_cubemap_desc d{};
    d.size = b.u32(); d.mip_count = b.u32(); d.format = b.u32();
    if (d.size) {
        size_t total = 0;
        for (uint32_t l = 0; l < d.mip_count; ++l) total += 6 * (size_t)mip_dim(d.size, l) * mip_dim(d.size, l) * (d.format == RDR_FORMAT_RGBA32F ? 16 : 4);
        d.data = b.take(total);
    }
    return d;
}

void write_file(const char* path, const void* data, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(data), (std::streamsize)n);
    if (!f) throw std::runtime_error(std::string("cannot write ") + path);
}

} // namespace

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <scene.bin> <width> <height> [--lod] [--cull] [--skybox] [--frames N] [--dump-vis f] [--dump-color f]\n", argv[0]);
        return 2;
    }
    try {
        rdr::RendererConfig cfg;
        cfg.cull_enabled = false;
        cfg.collect_stats = true;
        int frames = 1;
        const char* dump_vis = nullptr;
        const char* dump_color = nullptr;
        for (int i = 4; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--lod") cfg.lod_enabled = true;
            else if (a == "--cull") cfg.cull_enabled = true;
            else if (a == "--skybox") cfg.skybox = true;
            else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
            else if (a == "--dump-vis" && i + 1 < argc) dump_vis = argv[++i];
            else if (a == "--dump-color" && i + 1 < argc) dump_color = argv[++i];
            else throw std::runtime_error("unknown argument " + a);
        }
        Blob b;
        {
            std::ifstream f(argv[1], std::ios::binary);
            if (!f) throw std::runtime_error(std::string("cannot open ") + argv[1]);
            b.bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        }
        if (std::memcmp(b.take(8), "RDRSCN01", 8) != 0) throw std::runtime_error("scene.bin: bad magic");
        uint32_t hdr[16];
        for (uint32_t& h : hdr) h = b.u32();
        float cam[9];
        std::memcpy(cam, b.take(36), 36);

        rdr_scene_desc s{};
        s.vertex_count = hdr[0]; s.index_count = hdr[1]; s.meshlet_vertex_count = hdr[2]; s.meshlet_triangle_bytes = hdr[3];
        s.meshlet_count = hdr[4]; s.mesh_count = hdr[5]; s.draw_count = hdr[6]; s.material_count = hdr[7];
        s.transform_count = hdr[8]; s.meshlet_visibility_count = hdr[9]; s.image_count = hdr[10];
        s.vertices = reinterpret_cast<const rdr_vertex*>(b.take((size_t)hdr[0] * sizeof(rdr_vertex)));
        s.indices = reinterpret_cast<const uint32_t*>(b.take((size_t)hdr[1] * 4));
        s.meshlet_vertices = reinterpret_cast<const uint32_t*>(b.take((size_t)hdr[2] * 4));
        s.meshlet_triangles = b.take(hdr[3]);
        s.meshlets = reinterpret_cast<const rdr_meshlet*>(b.take((size_t)hdr[4] * sizeof(rdr_meshlet)));
        s.meshes = reinterpret_cast<const rdr_mesh*>(b.take((size_t)hdr[5] * sizeof(rdr_mesh)));
        s.mesh_draws = reinterpret_cast<const rdr_mesh_draw*>(b.take((size_t)hdr[6] * sizeof(rdr_mesh_draw)));
        s.materials = reinterpret_cast<const rdr_material*>(b.take((size_t)hdr[7] * sizeof(rdr_material)));
        s.global_transforms = reinterpret_cast<const float*>(b.take((size_t)hdr[8] * 64));
        std::vector<rdr_image_desc> images(hdr[10]);
        for (auto& im : images) im = read_image(b);
        s.images = images.data();
        rdr_env_desc env{};
        if (hdr[11]) {
            env.brdf_lut = read_image(b);
            env.skybox = read_cube(b);
            env.lambertian = read_cube(b);
            env.ggx = read_cube(b);
            env.charlie = read_cube(b);
            s.env = &env;
        }

        const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]);
        rdr::Renderer renderer(w, h, cfg);
        renderer.set_scene(s);
        rdr::Camera camera;
        camera.position = {cam[0], cam[1], cam[2]};
        camera.direction = {cam[3], cam[4], cam[5]};
        for (int f = 0; f < frames; ++f) renderer.render(camera, {cam[6], cam[7], cam[8]});
        const std::vector<uint32_t> vis = renderer.read_visbuffer();
        const std::vector<uint8_t> color = renderer.read_color();
        const rdr_stats st = renderer.stats();
        if (dump_vis) write_file(dump_vis, vis.data(), vis.size() * 4);
        if (dump_color) write_file(dump_color, color.data(), color.size());
        std::printf("rendered %ux%u: %llu triangles submitted, %llu rasterised, %llu pixels covered\n", w, h,
                    (unsigned long long)st.triangles_submitted, (unsigned long long)st.triangles_rasterised, (unsigned long long)st.pixels_covered);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rdr_harness: %s\n", e.what());
        return 1;
    }
}
