// pth_renderer.h -- headless C++ mirror of the reference's Renderer (renderer.rs:134-177,183,
// 583,744,867-904,906-1083) on top of the C ABI.  No window, swapchain or Vulkan objects: the
// ptx context owns the images; record_commands() renders one frame.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "pth_scene.h"

namespace pth {

struct Texture { uint32_t bindless_indice = 0; };   // graphics.rs:354-381

class Renderer {
    ptx_context *ctx_ = nullptr;      // declared first: `scene` is constructed from it

public:
    // Renderer::new(render_resolution) (renderer.rs:183); throws std::runtime_error on failure
    Renderer(uint32_t width, uint32_t height, int device_ordinal, const ptx_limits *limits = nullptr);
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    // renderer.rs:583-666
    void resize_render_buffer(uint32_t width, uint32_t height);
    // renderer.rs:867-904
    const Texture *lookup_texture(const std::string &ident) const;
    void add_texture_to_lookup_map(const std::string &ident, const Texture &t) { loaded_textures_map_[ident] = t; }
    uint32_t add_texture(const uint8_t *rgba8, uint32_t w, uint32_t h, bool srgb, bool repeat = false, bool linear = false);

    // Build extension (the --no-ref-quirks switch; default true = the reference's behaviour).
    // false: normals go to world space, texels are multiplied by baseColorFactor, import_gltf descends into
    // child nodes and honours the glTF sampler's wrap mode.  Call before importing / the first frame.
    void set_ref_quirks(bool on);
    bool ref_quirks() const { return ref_quirks_; }

    // renderer.rs:906-1083: one frame = one sample per pixel.  The reference draws a fresh
    // rand::random() seed per frame (renderer.rs:1005,1021); set fixed_seed for reproducible
    // output (every frame then uses `seed`), otherwise a SplitMix64 stream seeded by `seed`.
    void record_commands();
    // n frames without touching camera or scene, in one ptx_render_batch (fixed_seed only)
    void record_commands_batch(uint32_t n_frames);

    // renderer.rs:744-817 (without the OIDN branch): RGBA32F -> linear clamp x255 -> RGB8 PNG.
    // Refuses to overwrite an existing file.
    void save_render_texture_to_file(const std::string &path);
    std::vector<float> read_image(int kind);     // renderer.rs:668-742

    ptx_context *context() { return ctx_; }
    uint32_t width() const { return render_resolution_[0]; }
    uint32_t height() const { return render_resolution_[1]; }

    // pub fields of the reference (renderer.rs:148-176)
    Scene scene;
    Camera camera;
    uint32_t samples_per_pixel = 1;
    float z_near = 0.1f, z_far = 100.0f;            // renderer.rs:565-566
    uint32_t accumulated_count = 0;
    uint32_t accumulation_limit = 0xFFFFFFFFu;
    Camera prev_cam_state;
    bool resized_dirty = false;

    bool fixed_seed = true;
    uint32_t seed = 0;
    ptx_render_params params;

private:
    void sync_instances();
    bool settle_state();
    void frame_uniforms(ptx_uniforms &u, ptx_camera &cam);
    uint32_t next_seed();
    uint32_t render_resolution_[2] = {0, 0};
    std::map<std::string, Texture> loaded_textures_map_;
    uint64_t rng_state_ = 0;
    bool rng_started_ = false;
    bool ref_quirks_ = true;
};

}  // namespace pth
