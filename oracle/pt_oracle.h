/*
 * pt_oracle.h -- CPU restatement of AHL00/path_tracer's render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under path_tracer_amd/ may include, link
 * or call this.  It is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported CPU baseline.
 *
 * PARITY STATUS: the reference has no tests, fixtures or golden images, cannot
 * be built or run here (Rust + Vulkan KHR ray tracing + Windows, SURVEY.md
 * 8c), and keeps its BVH build/traversal/triangle test inside the GPU driver.
 * So: the integrator arithmetic below follows the reference's GLSL line by
 * line (cited per function); the pieces the reference delegates to third
 * parties (driver ray/triangle test + tie-break, glam 0.30.2 matrices,
 * GLSL sin/cos precision, vulkano 0.35.1 sampler defaults, image decoders)
 * are restated from their published behaviour and are "parity unpinned".
 *
 * Data layouts are those of /root/reference/src/shaders/shared.glsl:1-60.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shared.glsl:1-6 (std430: vec3 @0, vec3 @16, vec2 @32, vec2 @40) = 48 B */
typedef struct {
    float position[3]; float _pad0;
    float normal[3];   float _pad1;
    float uv[2];
    float padding[2];
} orc_vertex;

/* shared.glsl:8-15, renderer.rs:932-939 = 24 B */
typedef struct {
    uint32_t vertex_offset, vertex_count, material_offset, padding, index_offset, index_count;
} orc_offsets;

/* shared.glsl:27-36, material.rs:70-76 = 32 B */
typedef struct {
    float base_color[4];
    uint32_t has_base_texture;
    uint32_t base_texture_indice;
    float metallic;
    float roughness;
} orc_material;

/* shared.glsl:17-21, column-major mat4 x3 = 192 B (renderer.rs:1011-1015) */
typedef struct {
    float view_proj[16];
    float view_inverse[16];
    float proj_inverse[16];
} orc_camera;

/* shared.glsl:51-60 = 32 B (renderer.rs:1003-1010) */
typedef struct {
    uint32_t samples_per_pixel, seed, accumulated_count;
    float z_near, z_far;
    uint32_t padding[3];
} orc_uniforms;

/* one TLAS instance: renderer.rs:944-954 (3x4 row-major transform from
 * mat4_to_as_instance_array, renderer.rs:1416-1423) + its Offsets row */
typedef struct {
    float xform[3][4];
    uint32_t geometry;      /* unused by the oracle (triangles come from off) */
    orc_offsets off;
} orc_instance;

typedef struct {
    const uint8_t *rgba8;   /* W*H*4, row 0 first */
    uint32_t width, height;
    uint32_t srgb;          /* bit0: R8G8B8A8_SRGB (graphics.rs:494-495); bit1 (build extension):
                               wrap = repeat instead of the reference's clamp-to-edge; bit2 (same): bilinear filter
                               instead of nearest */
    uint32_t _pad;
} orc_texture;

/* Build extension (todo.md:4 "Proper lighting"): material properties the
 * reference's Material (material.rs:70-76) does not carry.  One row per material, same index. */
typedef struct {
    float emissive[3];      /* glTF emissiveFactor x KHR_materials_emissive_strength (linear radiance) */
    float transmission;     /* KHR_materials_transmission: probability of the smooth dielectric lobe */
    float ior;              /* KHR_materials_ior (1.5) */
    uint32_t mr_texture;    /* glTF metallicRoughnessTexture: texture index + 1, 0 = none; G scales the roughness factor,
                               B the metallic factor (lit integrator) */
    uint32_t normal_texture;   /* glTF normalTexture: texture index + 1, 0 = none; a LINEAR texture holding a tangent-space normal
                                  (x, y, z) = texel * 2 - 1, +Y up; the tangent frame is derived per triangle from its world-space
                                  positions and uvs; x and y are multiplied by orc_scene_desc.normal_scale (glTF
                                  normalTexture.scale) before the normal is put together (lit integrator) */
    uint32_t emissive_texture; /* glTF emissiveTexture: texture index + 1, 0 = none; emitted radiance = emissive x texel (lit
                                  integrator); the light table still picks triangles by area x luminance(emissive) */
} orc_material_ext;

/* scene options (orc_scene_desc.options); 0 = the reference's behaviour, quirks included */
#define ORC_SCENE_WORLD_NORMALS 1u        /* normals taken to world space by the inverse transpose (rchit.glsl:175-179 does not) */
#define ORC_SCENE_TEXTURE_TIMES_FACTOR 2u /* albedo = texel * baseColorFactor (rchit.glsl:193-195 drops the factor) */

/* orc_params.integrator */
#define ORC_INTEGRATOR_REFERENCE 0u
#define ORC_INTEGRATOR_LIT 1u             /* emissive triangles + next-event estimation + metallic-roughness / glass BSDFs */

/* constants hard-coded in the reference, exposed as parameters */
typedef struct {
    uint32_t max_depth;         /* rgen.glsl:18  (16) */
    float t_min;                /* rgen.glsl:37  (0.001) */
    float t_max;                /* rgen.glsl:38  (10000) */
    float normal_eps;           /* rchit.glsl:242 (0.001) */
    uint32_t rng_sample_offset; /* added to accumulated_count for the RNG only (spp partition) */
    uint32_t flags;             /* bit0: force brute-force closest hit; bit1 (checker of the checker): lit integrator without
                                   multiple importance sampling -- light samples unweighted, emitters reached by a sampled
                                   diffuse / glossy bounce dropped; same expectation, used by the CPU tests only */
    uint32_t integrator;        /* ORC_INTEGRATOR_* */
    uint32_t _pad;
} orc_params;

typedef struct {
    const orc_vertex *vertices;   uint64_t n_vertices;
    const uint32_t *indices;      uint64_t n_indices;
    const orc_material *materials; uint64_t n_materials;
    const orc_instance *instances; uint64_t n_instances;
    const orc_texture *textures;   uint64_t n_textures;
    const orc_material_ext *material_ext;   /* n_materials rows, or NULL: no emission, no transmission */
    uint64_t options;                       /* ORC_SCENE_* */
    const float *normal_scale;              /* glTF normalTexture.scale per material (n_materials), or NULL: 1 */
} orc_scene_desc;

typedef struct orc_scene orc_scene;

typedef struct {
    uint64_t rays;      /* path segments traced (= traceRayEXT calls) */
    uint64_t hits;      /* segments that hit geometry */
    uint64_t samples;   /* pixel-samples rendered */
    uint64_t textured_hits;
    uint64_t nodes_visited, tris_tested;
    uint64_t shadow_rays;   /* lit integrator: occlusion queries (not counted in rays) */
} orc_stats;

void orc_default_params(orc_params *p);

/* Flattens instances to world-space triangles and builds the oracle's own
 * BVH2 (used unless flags bit0).  Keeps pointers to vertices / indices /
 * materials / textures: they must outlive the scene. */
orc_scene *orc_scene_create(const orc_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
uint64_t orc_scene_triangle_count(const orc_scene *s);

/*
 * Renders n_samples frames.  Frame i uses accumulated_count = u->accumulated_count + i
 * for the running mean / G-buffer rule and (that + rng_sample_offset) for the RNG.
 * albedo: W*H*4 float (read-modify-write running mean, rgen.glsl:124-141)
 * normal: W*H*4 float, depth: W*H float (written only by accumulated_count==0 frames)
 * primary_ids: W*H*2 u32 (instance, primitive) of the first hit, 0xFFFFFFFF on miss; may be NULL
 * Row 0 = top of the view (gl_LaunchIDEXT.y = 0).
 */
void orc_render(const orc_scene *s, const orc_camera *cam, const orc_uniforms *u,
                const orc_params *p, uint32_t width, uint32_t height, uint32_t n_samples,
                float *albedo, float *normal, float *depth, uint32_t *primary_ids,
                orc_stats *stats, int n_threads);

/* Rows [row_begin, row_end) of the same frames (image pointers are those of the full W x H images;
 * the other rows are left untouched): a pixel depends only on its own launch id. */
void orc_render_rows(const orc_scene *s, const orc_camera *cam, const orc_uniforms *u,
                     const orc_params *p, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                     uint32_t n_samples, float *albedo, float *normal, float *depth, uint32_t *primary_ids,
                     orc_stats *stats, int n_threads);

/* lit integrator: number of emissive triangles in the light table, and row i of it
 * (9 floats of world-space vertices, 3 of radiance, cdf, pdf per unit area) for the tests */
uint64_t orc_scene_light_count(const orc_scene *s);
void orc_scene_light(const orc_scene *s, uint64_t i, float verts[9], float le[3], float *cdf, float *pdf_area);

/* closest hit for one ray (brute force if flags bit0). Returns 1 on hit. */
int orc_trace(const orc_scene *s, const float org[3], const float dir[3], float t_min, float t_max,
              uint32_t flags, float *t, float *u, float *v, uint32_t *instance, uint32_t *prim);

/* known-answer helpers (shared.glsl:156-168,194-207,236-271) */
uint32_t orc_hash(uint32_t s);
uint32_t orc_hash2(uint32_t x, uint32_t y);
float orc_rand(float uvx, float uvy, uint32_t depth, uint32_t accumulated_count, uint32_t seed);
void orc_create_basis(const float n[3], float t[3], float b[3]);
void orc_cosine_hemisphere_sample(float uvx, float uvy, uint32_t depth, uint32_t acc, uint32_t seed,
                                  const float n[3], float out[3]);
void orc_sincos(float phi, float *s, float *c);
void orc_sky_color(const float dir[3], float out[3]);
float orc_srgb_to_linear(uint32_t c);

/* host-side restatements: scene/mod.rs:557-566, renderer.rs:970-980,1011-1015,1416-1423 */
void orc_transform_matrix(const float pos[3], const float quat_xyzw[4], const float scale[3],
                          float xform[3][4]);
void orc_camera_matrices(const float pos[3], const float quat_xyzw[4], float fov_y, float aspect,
                         float z_near, float z_far, orc_camera *out);

#ifdef __cplusplus
}
#endif
#endif
