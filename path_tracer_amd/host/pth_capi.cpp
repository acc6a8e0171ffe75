// pth_capi.cpp -- C shim over the C++ host mirror (Renderer / Scene) so that Python (tests,
// bench.py, the headless driver) can drive it with ctypes.  Every function returns 0 on success;
// pth_last_error() holds the message of the last failure on the calling thread.
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "pth_image.h"
#include "pth_renderer.h"

using namespace pth;

namespace {
thread_local std::string g_err;
template <class F> int guard(F &&f)
{
    try { f(); return 0; }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
    catch (...) { g_err = "unknown error"; return 1; }
}
}  // namespace

extern "C" {

const char *pth_last_error() { return g_err.c_str(); }

int pth_renderer_create(uint32_t width, uint32_t height, int device, const ptx_limits *limits, void **out)
{
    return guard([&] { *out = new Renderer(width, height, device, limits); });
}
void pth_renderer_destroy(void *r) { delete (Renderer *)r; }
void *pth_renderer_context(void *r) { return ((Renderer *)r)->context(); }

int pth_import_gltf(void *r, const char *path, float ox, float oy, float oz)
{
    return guard([&] { Scene::import_gltf(*(Renderer *)r, path, Vec3(ox, oy, oz)); });
}

int pth_set_ref_quirks(void *r, int on) { return guard([&] { ((Renderer *)r)->set_ref_quirks(on != 0); }); }

// moves entity `index` (the UI's transform editor, config_window.rs); the next frame is scene-dirty
int pth_set_entity_transform(void *r, uint64_t index, const float pos[3], const float quat_xyzw[4], const float scale[3])
{
    return guard([&] {
        std::vector<Entity> &w = ((Renderer *)r)->scene.world_mut();
        if (index >= w.size()) throw std::runtime_error("entity index out of range");
        w[index].transform.position = Vec3(pos[0], pos[1], pos[2]);
        w[index].transform.rotation = Quat{quat_xyzw[0], quat_xyzw[1], quat_xyzw[2], quat_xyzw[3]};
        w[index].transform.scale = Vec3(scale[0], scale[1], scale[2]);
    });
}

int pth_set_camera(void *r, const float pos[3], const float quat_xyzw[4], float fov_y, float z_near, float z_far)
{
    Renderer *R = (Renderer *)r;
    R->camera.transform.position = Vec3(pos[0], pos[1], pos[2]);
    R->camera.transform.rotation = Quat{quat_xyzw[0], quat_xyzw[1], quat_xyzw[2], quat_xyzw[3]};
    R->camera.fov_y = fov_y; R->camera.near = z_near; R->camera.far = z_far;
    return 0;
}
int pth_set_seed(void *r, uint32_t seed, int fixed) { ((Renderer *)r)->seed = seed; ((Renderer *)r)->fixed_seed = fixed != 0; return 0; }
int pth_set_params(void *r, const ptx_render_params *p) { ((Renderer *)r)->params = *p; return 0; }
int pth_set_accumulation_limit(void *r, uint32_t limit) { ((Renderer *)r)->accumulation_limit = limit; return 0; }
uint32_t pth_accumulated_count(void *r) { return ((Renderer *)r)->accumulated_count; }

int pth_resize(void *r, uint32_t w, uint32_t h) { return guard([&] { ((Renderer *)r)->resize_render_buffer(w, h); }); }
int pth_record_commands(void *r) { return guard([&] { ((Renderer *)r)->record_commands(); }); }
int pth_record_commands_batch(void *r, uint32_t n) { return guard([&] { ((Renderer *)r)->record_commands_batch(n); }); }
int pth_save_image(void *r, const char *path) { return guard([&] { ((Renderer *)r)->save_render_texture_to_file(path); }); }

// scene inspection (tests compare these with the oracle-side ingest)
uint64_t pth_scene_counts(void *r, uint64_t *n_vertices, uint64_t *n_indices, uint64_t *n_materials, uint64_t *n_entities)
{
    Renderer *R = (Renderer *)r;
    *n_vertices = R->scene.vertices.size(); *n_indices = R->scene.indices.size();
    *n_materials = R->scene.materials.size(); *n_entities = R->scene.world().size();
    return 0;
}
int pth_scene_copy(void *r, ptx_vertex *v, uint32_t *idx, ptx_material *m, ptx_instance *inst)
{
    Renderer *R = (Renderer *)r;
    if (v) memcpy(v, R->scene.vertices.data(), R->scene.vertices.size() * sizeof(ptx_vertex));
    if (idx) memcpy(idx, R->scene.indices.data(), R->scene.indices.size() * 4);
    if (m) memcpy(m, R->scene.materials.data(), R->scene.materials.size() * sizeof(ptx_material));
    if (inst) {
        const auto &ents = R->scene.world();
        for (size_t i = 0; i < ents.size(); ++i) {
            Mat4 mm = ents[i].transform.get_matrix();
            for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 4; ++c) inst[i].xform[rr][c] = mm.m[c * 4 + rr];
            inst[i].geometry = ents[i].geometry.handle;
            inst[i].off = ptx_offsets{(uint32_t)ents[i].geometry.shared_vertex_buffer_offset, (uint32_t)ents[i].geometry.shared_vertex_buffer_count,
                                      (uint32_t)ents[i].material.get_shared_buffer_offset(), 0,
                                      (uint32_t)ents[i].geometry.shared_index_buffer_offset, (uint32_t)ents[i].geometry.shared_index_buffer_count};
        }
    }
    return 0;
}

// camera matrices exactly as record_commands would send them (renderer.rs:970-980,1011-1015)
int pth_camera_matrices(const float pos[3], const float q[4], float fov_y, float aspect, float z_near, float z_far, ptx_camera *out)
{
    Transform t; t.position = Vec3(pos[0], pos[1], pos[2]); t.rotation = Quat{q[0], q[1], q[2], q[3]};
    Mat4 proj = Mat4::perspective_rh(fov_y, aspect, z_near, z_far);
    Mat4 view = Mat4::look_to_rh(t.position, t.forward(), t.up());
    Mat4 vp = proj * view, vi = view.inverse(), pi = proj.inverse();
    memcpy(out->view_proj, vp.m, 64); memcpy(out->view_inverse, vi.m, 64); memcpy(out->proj_inverse, pi.m, 64);
    return 0;
}

// image codecs (no GPU needed)
int pth_decode_image(const char *path, uint32_t *w, uint32_t *h, int *channels, uint8_t *dst, uint64_t dst_bytes)
{
    Image img; std::string err;
    if (!decode_image_file(path, img, err)) { g_err = err; return 1; }
    *w = img.width; *h = img.height; *channels = img.channels;
    if (dst) {
        if (dst_bytes < img.pixels.size()) { g_err = "destination too small"; return 1; }
        memcpy(dst, img.pixels.data(), img.pixels.size());
    }
    return 0;
}
int pth_encode_png_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb)
{
    std::string err;
    if (!encode_png_rgb8(path, w, h, rgb, err)) { g_err = err; return 1; }
    return 0;
}

}  // extern "C"
