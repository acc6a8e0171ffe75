// pth_scene.h -- headless C++ mirror of the reference's scene-side host types, driving the C ABI
// of include/ptx.h.  Same names, argument meaning and error behaviour as the Rust originals:
//   Transform        scene/mod.rs:523-580
//   Camera           camera.rs:4-24
//   Material         material.rs:13-95
//   Geometry         scene/geometry/mod.rs:24-134
//   Scene            scene/mod.rs:56-520 (shared buffers, geometry map, entity list)
//   Scene::import_gltf  scene/gltf.rs:14-152
// The legion ECS is replaced by a plain entity vector (iteration order = push order, which is
// what renderer.rs:918-955 relies on).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "pth_math.h"

namespace pth {

struct Transform {
    Vec3 position{0, 0, 0};
    Quat rotation{};
    Vec3 scale{1, 1, 1};
    Mat4 get_matrix() const { return Mat4::from_scale_rotation_translation(scale, rotation.normalize(), position); }
    Vec3 forward() const { return rotation.mul_vec3(Vec3(0, 0, -1)); }
    Vec3 right() const { return rotation.mul_vec3(Vec3(1, 0, 0)); }
    Vec3 up() const { return rotation.mul_vec3(Vec3(0, 1, 0)); }
    bool operator==(const Transform &o) const { return position == o.position && rotation == o.rotation && scale == o.scale; }
};

struct Camera {
    Transform transform;
    float fov_y = 70.0f * 3.14159265358979323846f / 180.0f;   // 70.0_f32.to_radians()
    float near = 0.1f;
    float far = 1000.0f;
    bool operator==(const Camera &o) const { return transform == o.transform && fov_y == o.fov_y && near == o.near && far == o.far; }
    bool operator!=(const Camera &o) const { return !(*this == o); }
};

struct Material {
    uint64_t shared_buffer_index = 0;
    std::string name;
    float base_color[4] = {1, 1, 1, 1};
    bool has_base_texture = false;
    uint32_t base_texture_indice = 0;
    float metallic = 1.0f, roughness = 1.0f;
    // build extension: what glTF carries beyond material.rs:13-24; sent with ptx_set_material_ext
    float emissive[3] = {0, 0, 0};      // emissiveFactor x KHR_materials_emissive_strength
    float transmission = 0.0f;          // KHR_materials_transmission
    float ior = 1.5f;                   // KHR_materials_ior
    uint32_t mr_texture = 0;            // metallicRoughnessTexture: bindless index + 1, 0 = none (not under ref quirks)
    uint32_t normal_texture = 0;        // normalTexture (linear), emissiveTexture (sRGB): the same convention
    uint32_t emissive_texture = 0;
    float normal_scale = 1.0f;          // normalTexture.scale
    uint64_t get_shared_buffer_offset() const { return shared_buffer_index; }
};

struct Geometry {
    uint32_t handle = 0;                 // ptx geometry (the reference holds a BLAS here)
    uint32_t vertex_count = 0, index_count = 0;
    uint64_t shared_vertex_buffer_offset = 0, shared_vertex_buffer_count = 0;
    uint64_t shared_index_buffer_offset = 0, shared_index_buffer_count = 0;
};

struct Entity {
    Geometry geometry;
    Transform transform;
    Material material;
};

struct CustomVertex { float position[3], normal[3], tex_coords[2]; };   // renderer.rs:114-127

class Renderer;

class Scene {
public:
    explicit Scene(ptx_context *ctx) : ctx_(ctx) {}

    // scene/mod.rs:270-337,405-467,340-403: bump-append; returns (start, end) / index.  Throws
    // std::runtime_error where the reference panics on overflow.
    std::pair<uint64_t, uint64_t> add_vertices(const std::vector<ptx_vertex> &v);
    std::pair<uint64_t, uint64_t> add_indices(const std::vector<uint32_t> &i);
    uint64_t add_material(const ptx_material &m);
    void set_material_ext(uint64_t index, const ptx_material_ext &e);   // build extension
    void set_material_normal_scale(uint64_t index, float scale);        // build extension: glTF normalTexture.scale

    const Geometry *get_geometry(const std::string &ident) const;
    void add_geometry(const std::string &ident, const Geometry &g) { geometries_[ident] = g; }
    // geometry/mod.rs:54-103
    Geometry create_geometry(const std::string &ident, const std::vector<CustomVertex> &vertices,
                             const std::vector<uint32_t> &indices);

    // scene/gltf.rs:14-129
    static void import_gltf(Renderer &renderer, const std::string &path, Vec3 position_offset);

    void push(const Entity &e) { entities_.push_back(e); dirty_ = true; rebuild_ = true; }
    std::vector<Entity> &world_mut() { dirty_ = true; rebuild_ = true; return entities_; }
    const std::vector<Entity> &world() const { return entities_; }
    bool check_dirty_and_reset() { bool d = dirty_; dirty_ = false; return d; }
    // build-side flag: the entity list changed since the acceleration structure was last built
    bool take_rebuild_flag() { bool r = rebuild_; rebuild_ = false; return r; }

    // host copies kept for inspection / tests
    std::vector<ptx_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<ptx_material> materials;
    std::vector<ptx_material_ext> material_ext;

private:
    ptx_context *ctx_;
    std::map<std::string, Geometry> geometries_;
    std::vector<Entity> entities_;
    bool dirty_ = true;
    bool rebuild_ = true;
};

// gltf.rs:132-152
std::vector<Vec3> calculate_normals(const std::vector<Vec3> &positions, const std::vector<uint32_t> &indices);

}  // namespace pth
