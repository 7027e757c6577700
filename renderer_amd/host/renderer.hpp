// renderer.hpp — header-only C++ mirror of the reference's Renderer over the C ABI.
//
// Same call shape as renderer/src/renderer.rs:780-1060 (Renderer::new / set_scene /
// render) and scenery/src/camera.rs (Camera); every fallible call throws
// std::runtime_error carrying rdr_last_error(), the counterpart of anyhow::Result.
#pragma once

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdr_abi.h"

namespace rdr {

struct Camera { // scenery/src/camera.rs:6-9; the view looks along -direction
    std::array<float, 3> position{0, 0, 0};
    std::array<float, 3> direction{0, 0, 1};
};

struct RendererConfig { // scenery/src/lib.rs:7-14, the fields that reach the GPU
    bool lod_enabled = false;
    bool cull_enabled = true;
    bool collect_stats = false;
    bool time_kernels = false;
    bool skybox = false; // run the skybox pass on pixels without geometry (renderer/src/renderer/pass.rs:397-445)
};

class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, const RendererConfig& cfg = {}, int hip_device = 0)
        : width_(width), height_(height) {
        rdr_config c{cfg.lod_enabled, cfg.cull_enabled, cfg.collect_stats, cfg.time_kernels, cfg.skybox};
        check(rdr_create(&c, width, height, &hip_device, 1, &handle_));
    }
    ~Renderer() { rdr_destroy(handle_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void set_scene(const rdr_scene_desc& scene) { check(rdr_set_scene(handle_, &scene)); }
    void update_transforms(const float* mat4s, uint32_t count) { check(rdr_update_transforms(handle_, mat4s, count)); }
    void render(const Camera& cam, const std::array<float, 3>& sun) {
        check(rdr_render(handle_, cam.position.data(), cam.direction.data(), sun.data()));
    }
    std::vector<uint8_t> read_color() {
        std::vector<uint8_t> out((size_t)width_ * height_ * 4);
        check(rdr_read_color(handle_, out.data()));
        return out;
    }
    std::vector<uint32_t> read_visbuffer() {
        std::vector<uint32_t> out((size_t)width_ * height_ * 2);
        check(rdr_read_visbuffer(handle_, out.data()));
        return out;
    }
    rdr_stats stats() {
        rdr_stats s{};
        check(rdr_get_stats(handle_, &s));
        return s;
    }
    rdr_renderer* handle() { return handle_; }

private:
    static void check(int rc) {
        if (rc) throw std::runtime_error(rdr_last_error());
    }
    rdr_renderer* handle_ = nullptr;
    uint32_t width_, height_;
};

} // namespace rdr
