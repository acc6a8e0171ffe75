// pth_renderer.cpp -- Renderer mirror: frame loop, accumulation bookkeeping, readback, save.
#include "pth_renderer.h"

#include <sys/stat.h>

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "pth_image.h"

namespace pth {

namespace {

ptx_context *make_context(int device, const ptx_limits *limits)
{
    ptx_context *c = nullptr;
    int rc = limits ? ptx_create_with_limits(device, limits, &c) : ptx_create(device, &c);
    if (rc) throw std::runtime_error(std::string("Renderer::new: ") + ptx_last_error(nullptr));
    return c;
}

[[noreturn]] void ptx_throw(ptx_context *ctx, const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + ptx_last_error(ctx));
}

}  // namespace

Renderer::Renderer(uint32_t width, uint32_t height, int device, const ptx_limits *limits)
    : ctx_(make_context(device, limits)), scene(ctx_)
{
    ptx_default_params(&params);
    resize_render_buffer(width, height);     // renderer.rs:574
    resized_dirty = false;                   // a fresh renderer starts clean apart from the scene flag
}

Renderer::~Renderer() { ptx_destroy(ctx_); }

void Renderer::resize_render_buffer(uint32_t width, uint32_t height)
{
    if (ptx_resize(ctx_, width, height)) ptx_throw(ctx_, "resize_render_buffer");
    render_resolution_[0] = width; render_resolution_[1] = height;
    resized_dirty = true;                    // renderer.rs:665
}

const Texture *Renderer::lookup_texture(const std::string &ident) const
{
    auto it = loaded_textures_map_.find(ident);
    return it == loaded_textures_map_.end() ? nullptr : &it->second;
}

uint32_t Renderer::add_texture(const uint8_t *rgba8, uint32_t w, uint32_t h, bool srgb, bool repeat, bool linear)
{
    uint32_t idx = 0;
    int flags = (srgb ? PTX_TEXTURE_SRGB : 0) | (repeat ? PTX_TEXTURE_REPEAT : 0) | (linear ? PTX_TEXTURE_LINEAR : 0);
    if (ptx_add_texture(ctx_, rgba8, w, h, flags, &idx)) ptx_throw(ctx_, "Bindless texture pool size exceeded");
    return idx;
}

void Renderer::set_ref_quirks(bool on)
{
    ref_quirks_ = on;
    if (ptx_set_scene_options(ctx_, on ? 0u : (uint32_t)(PTX_SCENE_WORLD_NORMALS | PTX_SCENE_TEXTURE_TIMES_FACTOR)))
        ptx_throw(ctx_, "set_ref_quirks");
    scene.world_mut();      // the device scene has to be rebuilt with the new options
}

void Renderer::sync_instances()
{
    // renderer.rs:912-961 builds the Offsets rows + TLAS every frame; the acceleration structure
    // here is only rebuilt when the entity list changed (todo.md:3 asks for exactly that).
    const std::vector<Entity> &ents = scene.world();
    std::vector<ptx_instance> inst(ents.size());
    for (size_t i = 0; i < ents.size(); ++i) {
        const Entity &e = ents[i];
        Mat4 m = e.transform.get_matrix();
        for (int r = 0; r < 3; ++r)          // mat4_to_as_instance_array, renderer.rs:1416-1423
            for (int c = 0; c < 4; ++c) inst[i].xform[r][c] = m.m[c * 4 + r];
        inst[i].geometry = e.geometry.handle;
        inst[i].off.vertex_offset = (uint32_t)e.geometry.shared_vertex_buffer_offset;
        inst[i].off.vertex_count = (uint32_t)e.geometry.shared_vertex_buffer_count;
        inst[i].off.material_offset = (uint32_t)e.material.get_shared_buffer_offset();
        inst[i].off.padding = 0;
        inst[i].off.index_offset = (uint32_t)e.geometry.shared_index_buffer_offset;
        inst[i].off.index_count = (uint32_t)e.geometry.shared_index_buffer_count;
    }
    // ptx_update_instances refits when the list only moved and rebuilds otherwise
    if (ptx_update_instances(ctx_, inst.data(), (uint32_t)inst.size())) ptx_throw(ctx_, "build_top_level_acceleration_structure");
}

uint32_t Renderer::next_seed()
{
    if (fixed_seed) return seed;
    if (!rng_started_) { rng_state_ = seed; rng_started_ = true; }
    uint64_t z = (rng_state_ += 0x9E3779B97F4A7C15ull);     // SplitMix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

void Renderer::frame_uniforms(ptx_uniforms &u, ptx_camera &cam)
{
    // renderer.rs:970-980: aspect is the swapchain's, which the app pins to the render aspect
    float aspect = (float)render_resolution_[0] / (float)render_resolution_[1];
    Mat4 proj = Mat4::perspective_rh(camera.fov_y, aspect, camera.near, camera.far);
    Mat4 view = Mat4::look_to_rh(camera.transform.position, camera.transform.forward(), camera.transform.up());
    Mat4 vp = proj * view, vi = view.inverse(), pi = proj.inverse();   // renderer.rs:1011-1015
    memcpy(cam.view_proj, vp.m, 64); memcpy(cam.view_inverse, vi.m, 64); memcpy(cam.proj_inverse, pi.m, 64);
    memset(&u, 0, sizeof u);
    u.samples_per_pixel = samples_per_pixel;
    u.accumulated_count = accumulated_count;
    u.z_near = z_near; u.z_far = z_far;
}

bool Renderer::settle_state()
{
    // renderer.rs:988-1000, short-circuit order kept: a moving camera leaves the scene flag set,
    // so the frame after the camera stops is dirty once more
    bool cam_dirty = camera != prev_cam_state;
    bool was_resized = resized_dirty;
    resized_dirty = false;
    prev_cam_state = camera;
    bool state_dirty = cam_dirty || scene.check_dirty_and_reset() || was_resized;
    if (scene.take_rebuild_flag()) sync_instances();
    if (state_dirty) accumulated_count = 0;
    return state_dirty;
}

void Renderer::record_commands()
{
    bool state_dirty = settle_state();
    ptx_uniforms u; ptx_camera cam;
    frame_uniforms(u, cam);
    u.seed = next_seed();                                   // renderer.rs:1005,1021
    if (!state_dirty && accumulated_count < accumulation_limit) accumulated_count += 1;   // renderer.rs:1029-1031
    if (ptx_render(ctx_, &u, &cam, &params)) ptx_throw(ctx_, "trace_rays");
}

void Renderer::record_commands_batch(uint32_t n_frames)
{
    if (!fixed_seed) throw std::runtime_error("record_commands_batch needs fixed_seed");
    // Equivalent to n_frames record_commands() calls with camera and scene untouched in between:
    // a dirty first frame renders with accumulated_count 0 and does not advance the counter; clean
    // frames advance it by one each until accumulation_limit, after which every further frame
    // re-renders at the limit.  Runs of clean, advancing frames go through one ptx_render_batch.
    while (n_frames > 0) {
        bool state_dirty = settle_state();
        ptx_uniforms u; ptx_camera cam;
        frame_uniforms(u, cam);
        u.seed = seed;
        if (state_dirty || accumulated_count >= accumulation_limit) {
            if (ptx_render(ctx_, &u, &cam, &params)) ptx_throw(ctx_, "trace_rays");
            n_frames -= 1;
            continue;
        }
        uint64_t room = (uint64_t)accumulation_limit - accumulated_count;
        uint32_t batched = (uint32_t)std::min<uint64_t>(n_frames, room);
        if (ptx_render_batch(ctx_, &u, &cam, &params, batched)) ptx_throw(ctx_, "trace_rays");
        accumulated_count += batched;
        n_frames -= batched;
    }
}

std::vector<float> Renderer::read_image(int kind)
{
    uint64_t np = (uint64_t)render_resolution_[0] * render_resolution_[1];
    uint64_t floats = kind == PTX_IMAGE_DEPTH ? np : (kind == PTX_IMAGE_PRIMARY_ID ? np * 2 : np * 4);
    std::vector<float> out(floats);
    if (ptx_read_image(ctx_, kind, out.data(), floats * 4)) ptx_throw(ctx_, "copy_image_to_host_vec_f32");
    return out;
}

void Renderer::save_render_texture_to_file(const std::string &path)
{
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0) throw std::runtime_error("File already exists");   // renderer.rs:752-754
    std::vector<float> albedo = read_image(PTX_IMAGE_ALBEDO);
    uint64_t np = (uint64_t)render_resolution_[0] * render_resolution_[1];
    std::vector<uint8_t> rgb(np * 3);
    for (uint64_t i = 0; i < np; ++i)
        for (int c = 0; c < 3; ++c) {
            // DynamicImage::into_rgb8: clamp to [0,1], scale by 255, round (image 0.25.6, not vendored)
            float v = albedo[i * 4 + c];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            if (!(v == v)) v = 0.0f;
            rgb[i * 3 + c] = (uint8_t)std::lround(v * 255.0f);
        }
    std::string err;
    if (!encode_png_rgb8(path, render_resolution_[0], render_resolution_[1], rgb.data(), err)) throw std::runtime_error(err);
}

}  // namespace pth
