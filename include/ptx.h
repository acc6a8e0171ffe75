/*
 * ptx.h -- C ABI of the MI355X (gfx950) wavefront path-tracing backend that
 * replaces AHL00/path_tracer's Vulkan ray-tracing render loop.
 *
 * The reference has no FFI for this path (its only FFI is the unused nrd-sys
 * crate); each entry point below carries exactly the payload the reference's
 * Rust host already produces at the cited call site, so a Rust `extern "C"`
 * block binds it unchanged.  All structs are byte-identical
 * to /root/reference/src/shaders/shared.glsl:1-60.
 *
 * Conventions: every call returns a ptx_status code (0 = ok) -- no exception or
 * panic crosses the ABI; caller-owned inputs are copied before the call
 * returns; a context is not thread-safe, distinct contexts may be driven from
 * distinct threads; there is no CPU fallback (ptx_create fails without a HIP
 * device).
 */
#ifndef PTX_H
#define PTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PTX_OK = 0,
    PTX_ERR_INVALID = 1,    /* bad argument / out-of-range offset */
    PTX_ERR_CAPACITY = 2,   /* a fixed-capacity scene buffer is full (the reference panics: scene/mod.rs:281-283,347-349,412-414; renderer.rs:899-901) */
    PTX_ERR_HIP = 3,        /* HIP runtime error, text in ptx_last_error */
    PTX_ERR_STATE = 4       /* call order violated (e.g. render before resize/set_instances) */
} ptx_status;

/* shared.glsl:1-6, renderer.rs:91-111 -- 48 B */
typedef struct {
    float position[3]; float _pad0;
    float normal[3];   float _pad1;
    float uv[2];
    float padding[2];
} ptx_vertex;

/* shared.glsl:8-15, renderer.rs:932-939 -- 24 B */
typedef struct {
    uint32_t vertex_offset, vertex_count, material_offset, padding, index_offset, index_count;
} ptx_offsets;

/* shared.glsl:27-36, material.rs:70-76 -- 32 B */
typedef struct {
    float base_color[4];
    uint32_t has_base_texture;
    uint32_t base_texture_indice;
    float metallic;
    float roughness;
} ptx_material;

/* shared.glsl:17-21, renderer.rs:1011-1015 -- three column-major mat4, 192 B */
typedef struct {
    float view_proj[16];
    float view_inverse[16];
    float proj_inverse[16];
} ptx_camera;

/* shared.glsl:51-60, renderer.rs:1003-1010 -- 32 B */
typedef struct {
    uint32_t samples_per_pixel;   /* sent by the reference, unused by its shaders */
    uint32_t seed;
    uint32_t accumulated_count;
    float z_near, z_far;
    uint32_t padding[3];
} ptx_uniforms;

/* One TLAS instance + its Offsets row (renderer.rs:927-954): xform is the 3x4
 * row-major matrix of mat4_to_as_instance_array (renderer.rs:1416-1423); the
 * array index is gl_InstanceCustomIndexEXT.  76 B. */
typedef struct {
    float xform[3][4];
    uint32_t geometry;            /* handle from ptx_create_geometry */
    ptx_offsets off;
} ptx_instance;

/* Constants the reference hard-codes, with its values as defaults
 * (ptx_default_params): rgen.glsl:18,37-38; rchit.glsl:242. */
typedef struct {
    uint32_t max_depth;           /* 16 */
    float t_min;                  /* 0.001 */
    float t_max;                  /* 10000 */
    float normal_eps;             /* 0.001 */
    uint32_t rng_sample_offset;   /* added to accumulated_count for the RNG only (spp partition across GPUs); 0 */
    uint32_t flags;               /* reserved, 0 */
    uint32_t integrator;          /* PTX_INTEGRATOR_REFERENCE (0, default): the reference's rchit/rmiss shading */
    uint32_t _pad;
} ptx_render_params;

/* Build extensions past the reference (todo.md:4 "Proper lighting", todo.md:6
 * "GLTF textures stretched").  None of them changes the default, reference-exact behaviour. */
typedef enum {
    PTX_INTEGRATOR_REFERENCE = 0, /* rchit.glsl:158-254 + rmiss.glsl:18-34, quirks included */
    PTX_INTEGRATOR_LIT = 1        /* emissive triangles, next-event estimation with occlusion rays, glTF
                                     metallic-roughness (Lambert + GGX) and smooth-dielectric BSDFs, Russian
                                     roulette; specified by oracle/pt_oracle.c raygen_pixel_lit */
} ptx_integrator;

/* What a glTF material carries beyond material.rs:70-76; one row per material index. 32 B. */
typedef struct {
    float emissive[3];            /* emissiveFactor x KHR_materials_emissive_strength, linear radiance; 0 */
    float transmission;           /* KHR_materials_transmission, 0..1; 0 */
    float ior;                    /* KHR_materials_ior; 1.5 */
    uint32_t mr_texture;          /* glTF metallicRoughnessTexture as bindless index + 1 (0: none), added with
                                     ptx_add_texture WITHOUT the sRGB bit: roughness = roughnessFactor x G,
                                     metallic = metallicFactor x B at the hit's uv (lit integrator only) */
    uint32_t normal_texture;      /* glTF normalTexture as bindless index + 1 (0: none), a LINEAR texture: tangent-space
                                     normal = texel x 2 - 1, +Y up; the tangent frame is derived per triangle from its
                                     world-space positions and uvs (the vertex format has no tangent); glTF's
                                     normalTexture.scale: ptx_set_material_normal_scale */
    uint32_t emissive_texture;    /* glTF emissiveTexture as bindless index + 1 (0: none): emitted radiance = emissive x texel
                                     at the point's uv -- at a hit and at a light sample; the light table still picks
                                     triangles by area x luminance(emissive) (lit integrator only).  The texel is decoded the
                                     way the texture was registered (its PTX_TEXTURE_SRGB bit; the oracle does the same):
                                     glTF defines emissive maps as sRGB, so hosts add them with the bit set, as both
                                     loaders here do -- a map added without it is read as linear data, not rejected */
} ptx_material_ext;

typedef enum {
    PTX_SCENE_WORLD_NORMALS = 1,        /* vertex normals go to world space by the inverse transpose of the instance
                                           transform (rchit.glsl:175-179 leaves them in object space) */
    PTX_SCENE_TEXTURE_TIMES_FACTOR = 2  /* albedo = texel x baseColorFactor (rchit.glsl:193-195 drops the factor) */
} ptx_scene_option;

/* ptx_add_texture `srgb` argument as a flag word: bit0 sRGB decode, bit1 wrap = repeat instead of the
 * reference's clamp-to-edge, bit2 bilinear instead of the reference's nearest filter (graphics.rs:446-459
 * reads the glTF sampler's wrap and filter modes and then ignores them).  Bilinear: LOD 0, texel centres at
 * i + 0.5, the four texels decoded (sRGB -> linear) before the blend, neighbours wrapped / clamped per bit1. */
#define PTX_TEXTURE_SRGB 1
#define PTX_TEXTURE_REPEAT 2
#define PTX_TEXTURE_LINEAR 4

/* Fixed buffer capacities; defaults are the reference's (scene/mod.rs:48-54, renderer.rs:180). */
typedef struct {
    uint64_t max_vertices;        /* 2 500 000 */
    uint64_t max_indices;         /* 5 000 000 */
    uint32_t max_materials;       /* 10 000 */
    uint32_t max_offsets;         /* 10 000 (instances) */
    uint32_t max_textures;        /* 5 000 */
    uint32_t _pad;
} ptx_limits;

typedef enum {
    PTX_IMAGE_ALBEDO = 0,         /* RGBA32F running mean, A = 1   (renderer.rs:476-495) */
    PTX_IMAGE_NORMAL = 1,         /* RGBA32F first-hit normal, A=1 (renderer.rs:497-508) */
    PTX_IMAGE_DEPTH = 2,          /* R32F normalised depth         (renderer.rs:510-531) */
    PTX_IMAGE_PRIMARY_ID = 3      /* 2 x u32 (instance, primitive) of the first hit, 0xFFFFFFFF on miss -- build extension for the parity check */
} ptx_image_kind;

typedef struct {
    uint64_t rays;                /* path segments traced since the last ptx_reset_stats */
    uint64_t hits;
    uint64_t samples;             /* pixel-samples accumulated */
    uint64_t extend_launches;     /* launches of the traversal kernel */
    double extend_ms;             /* their summed GPU time (HIP events on the context stream; only while profiling is on) */
    double shade_ms;
    double total_ms;              /* GPU time of all render calls while profiling is on */
    uint64_t bvh_nodes, bvh_triangles, bvh_bytes;
    uint32_t bvh_depth, _pad;
    uint64_t nodes_fetched, tris_tested;   /* BVH nodes fetched / triangles tested; counted only at profiling level 2 */
    uint64_t wave_sched[6];                /* level 2: wave iterations, node blocks, lanes in them, triangle blocks, lanes in them, refills */
    uint64_t shadow_rays;                  /* lit integrator: occlusion queries (not counted in rays) */
    uint64_t lights;                       /* emissive triangles in the light table */
    uint64_t bvh_builds, bvh_refits;       /* acceleration-structure builds / refits since ptx_create */
    uint64_t bvh_device_refits;            /* of the refits, those done entirely on the GPU */
    uint64_t tail_launches;                /* chunks whose last bounces ran as one fused launch (since ptx_reset_stats) */
    uint64_t extend_phase_cycles[8];       /* level 2: shader cycles of wave lifetime per phase of the traversal kernel, summed over waves: ray
                                              fetch, node-load wait, node test + pick, triangle-queue append, triangle fetch wait, triangle
                                              test, hand-back of the hits, retire + loop control */
    uint64_t extend_append[3];             /* level 2: wave iterations that queued leaf triangles, passes of the per-lane queue-write loop in them,
                                              ray-fetch passes of the refills -- the weights tools/isa_budget.py prices the kernel's blocks with */
} ptx_stats;

typedef struct ptx_context ptx_context;

/* Renderer::new (renderer.rs:183).  One context per GPU. */
int ptx_create(int device_ordinal, ptx_context **ctx);
int ptx_create_with_limits(int device_ordinal, const ptx_limits *limits, ptx_context **ctx);
int ptx_destroy(ptx_context *ctx);
void ptx_default_limits(ptx_limits *limits);
void ptx_default_params(ptx_render_params *params);

/* Scene::add_vertices / add_indices / add_material (scene/mod.rs:270,405,340):
 * bump-append into the shared buffers, returning the element offset. */
int ptx_add_vertices(ptx_context *ctx, const ptx_vertex *vertices, uint64_t n, uint64_t *first);
int ptx_add_indices(ptx_context *ctx, const uint32_t *indices, uint64_t n, uint64_t *first);
int ptx_add_material(ptx_context *ctx, const ptx_material *material, uint32_t *index);

/* Build extension: extended properties of material `index` (see ptx_material_ext).  Takes effect at the next
 * ptx_set_instances.  No counterpart in the reference. */
int ptx_set_material_ext(ptx_context *ctx, uint32_t index, const ptx_material_ext *ext);
/* Build extension: glTF normalTexture.scale of material `index` (default 1): x and y of the tangent-space normal are
 * multiplied by it before the normal is put together (lit integrator).  Finite values only. */
int ptx_set_material_normal_scale(ptx_context *ctx, uint32_t index, float scale);
/* Build extension: bitwise OR of ptx_scene_option; takes effect at the next ptx_set_instances. */
int ptx_set_scene_options(ptx_context *ctx, uint32_t options);

/* Renderer::add_texture after Texture::from_gltf (renderer.rs:878, graphics.rs:383):
 * decoded 8-bit RGBA texels, row 0 first; srgb=1 is R8G8B8A8_SRGB.  Sampled as the
 * reference's default sampler does: nearest, clamp-to-edge, LOD 0.  The caller keeps
 * the ident -> index cache (renderer.rs:867-873). */
int ptx_add_texture(ptx_context *ctx, const uint8_t *rgba8, uint32_t width, uint32_t height, int srgb,
                    uint32_t *bindless_index);

/* Geometry::create's BLAS build (geometry/mod.rs:54-103): registers a vertex/index
 * range of the shared buffers as a geometry and validates its indices. */
int ptx_create_geometry(ptx_context *ctx, uint64_t first_vertex, uint64_t n_vertices, uint64_t first_index,
                        uint64_t n_indices, uint32_t *geometry);

/* The per-frame instance list + TLAS build of record_commands (renderer.rs:912-961).
 * Rebuilds the acceleration structure (synchronises the stream).  The caller restarts the
 * accumulation (accumulated_count = 0), as renderer.rs:995-1000 does when the scene is dirty. */
int ptx_set_instances(ptx_context *ctx, const ptx_instance *instances, uint32_t n);

/* Build extension (todo.md:3 "Stop rebuilding entire AS on every frame"; the reference's unused TLAS update,
 * renderer.rs:1086-1192): the same instance list as the last ptx_set_instances / ptx_update_instances -- same
 * geometry handles and Offsets rows -- with new transforms.  Keeps the tree and refits its boxes; falls back to a
 * full rebuild when the list differs or the moved scene has loosened the tree too far.  Same results as
 * ptx_set_instances either way (the closest hit does not depend on the tree). */
int ptx_update_instances(ptx_context *ctx, const ptx_instance *instances, uint32_t n);

/* Renderer::resize_render_buffer (renderer.rs:583): reallocates the three images, resets accumulation. */
int ptx_resize(ptx_context *ctx, uint32_t width, uint32_t height);

/* One trace_rays dispatch (renderer.rs:1002-1067) = 1 sample per pixel with the given
 * seed / accumulated_count.  Asynchronous on the context's stream. */
int ptx_render(ptx_context *ctx, const ptx_uniforms *uniforms, const ptx_camera *camera,
               const ptx_render_params *params);

/* Build extension: n_samples consecutive frames in one call, frame i using
 * accumulated_count = uniforms->accumulated_count + i.  Bit-identical to n_samples
 * ptx_render calls. */
int ptx_render_batch(ptx_context *ctx, const ptx_uniforms *uniforms, const ptx_camera *camera,
                     const ptx_render_params *params, uint32_t n_samples);

/* No counterpart in the reference.  For callers that cannot allocate device memory while rendering (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): one frame exactly like ptx_render -- same
 * images, bit for bit -- whose wavefront state (ray queues, hit records, per-path results, queue counters) lives in
 * a file-scope device array of the library instead of buffers allocated on demand.  The call makes no hipMalloc /
 * hipFree / stream or event creation and no host allocation.  Limits: width x height <= 262 144 pixels (512 x 512),
 * PTX_INTEGRATOR_REFERENCE only; beyond that it returns PTX_ERR_CAPACITY / PTX_ERR_INVALID with a message.  Every GPU
 * the library runs on holds its copy of the array (31.59 MB) until the process ends.  All contexts of one GPU share
 * that GPU's array.  Calls issued one after the other -- from one host thread, or from several with the calls
 * themselves serialised -- are ordered on the device too, across contexts (each static frame waits for the device's
 * previous one through a per-device event), so no ptx_synchronize is needed between them; calls ISSUED concurrently
 * from two threads on one GPU race.  Ordered with the context's other render calls; asynchronous on the context's main
 * stream.  Note for callers under graph capture: the first launch of a kernel on a stream lets the HIP runtime size
 * its private-segment (scratch) memory -- issue one warm-up frame before capturing. */
int ptx_render_static(ptx_context *ctx, const ptx_uniforms *uniforms, const ptx_camera *camera,
                      const ptx_render_params *params);

/* Image readback (renderer.rs:668-742).  Synchronises the stream.  Row 0 = top of the view.
 * bytes must equal width*height*{16,16,4,8}. */
int ptx_read_image(ptx_context *ctx, int kind, void *dst, uint64_t bytes);

/* Uploads an RGBA32F image into the accumulation buffer (resume from a dump, or the
 * combined result of a multi-GPU reduce). */
int ptx_write_albedo(ptx_context *ctx, const void *src, uint64_t bytes);

/* Multi-GPU (build extension).  One-process-per-GPU hosts (torch.distributed
 * over RCCL) use the device-pointer form: the albedo buffer scaled by `scale` is written to
 * / read from caller-owned DEVICE memory on the context's device, ordered on the context's
 * stream, so the caller can all-reduce it in place. */
int ptx_export_albedo_device(ptx_context *ctx, void *device_dst, uint64_t bytes, float scale);
int ptx_import_albedo_device(ptx_context *ctx, const void *device_src, uint64_t bytes, float scale);
/* Single-process hosts: sum the accumulation buffers of n contexts (one per GPU) into
 * ctxs[root] with an RCCL reduce over xGMI, then scale by 1/n. */
int ptx_reduce_accumulation(ptx_context **ctxs, int n, int root);

/* Build extension, no counterpart in the reference (its constants are compiled in, SURVEY.md 5 "config / flags"):
 * per-context scheduling knobs.  None of them changes a result bit.  Each has an environment variable that only sets
 * the DEFAULT of contexts created afterwards (named beside it), so two contexts of one process can differ and a host
 * never has to touch the environment.  Values out of range return PTX_ERR_INVALID. */
typedef enum {
    PTX_OPTION_FRAMES_IN_FLIGHT = 1,   /* 1 .. 4 chunks (frames, or chunks of a batch) overlapping on as many streams; 2   [PTX_FRAMES_IN_FLIGHT] */
    PTX_OPTION_TAIL_RAYS = 2,          /* expected queue length at or below which the remaining bounces of a chunk run as one
                                          fused launch; 0: never; 65536                                                    [PTX_TAIL_RAYS] */
    PTX_OPTION_TAIL_SHIFT = 3,         /* ... or at or below chunk paths >> shift; 7                                       [PTX_TAIL_SHIFT] */
    PTX_OPTION_TARGET_PATHS = 4,       /* paths per chunk of a batch (48 + 24 B of queue state each, per frame in flight);
                                          67 108 864                                                                       [PTX_TARGET_PATHS] */
    PTX_OPTION_MATERIAL_SORT = 5,      /* lit integrator: group each bounce's queue by material class before shading
                                          (measured 30 - 49 % slower); 0                              [PTX_MATERIAL_SORT] */
    PTX_OPTION_EXTEND_WAVES = 6,       /* waves of the persistent traversal grid; 0 = what the device holds (default)      [PTX_EXTEND_WAVES] */
    PTX_OPTION_SHADE_GRID = 7,         /* most blocks a shading launch starts; 0 = library default, 8192                   [PTX_SHADE_GRID] */
    PTX_OPTION_HOST_REFIT = 8,         /* 1: ptx_update_instances never refits on the device; 0                            [PTX_HOST_REFIT] */
    PTX_OPTION_REFIT_VERIFY = 9,       /* 1: every device refit is compared with a host refit (tests); 0                   [PTX_REFIT_VERIFY] */
    PTX_OPTION_TEXEL_SETS = 10,        /* lit integrator: 1 (default) interleaves the base colour / normal / metallic-roughness
                                          textures of a material (same size, same wrap and filter) into one array with one entry
                                          per fetch position -- one or two 16-byte loads per hit instead of up to twelve; costs up
                                          to 32 B per texel of device memory per such material, built on the first lit frame;
                                          0: the textures are fetched one by one                                              [PTX_TEXEL_SETS] */
    PTX_OPTION_TEXEL_SET_BYTES = 11    /* read-only (ptx_get_option): device memory the context's texel sets hold, in bytes */
} ptx_option;
int ptx_set_option(ptx_context *ctx, int key, uint64_t value);
int ptx_get_option(ptx_context *ctx, int key, uint64_t *value);

int ptx_synchronize(ptx_context *ctx);
int ptx_set_profiling(ptx_context *ctx, int level);     /* 0 off; 1 HIP-event timing of each stage launch (chunks still overlap on two streams, so a stage's
                                                           time includes waiting behind the other chunk's kernels); 2 also the instrumented traversal kernel
                                                           (slower) with every chunk on one stream, so that shade_ms is the shading kernels' own time */
int ptx_get_stats(ptx_context *ctx, ptx_stats *stats);  /* synchronises */
int ptx_reset_stats(ptx_context *ctx);
const char *ptx_last_error(ptx_context *ctx);           /* ctx may be NULL: last error of ptx_create on this thread */

#ifdef __cplusplus
}
#endif
#endif
