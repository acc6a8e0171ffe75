// ptx-render -- headless driver over the C++ host mirror: loads glTF scenes the way
// the reference's render_window.rs:412-426 does, renders a sample budget on one or more GPUs and
// writes what the reference's save path and its OIDN/NRD hand-off read back (renderer.rs:668-817):
//   <out>.png          albedo, clamp to [0,1] x255 -> RGB8 (renderer.rs:805)
//   <out>_albedo.pfm   RGB32F running mean
//   <out>_normal.pfm   RGB32F first-hit normal
//   <out>_depth.pfm    R32F normalised depth
// Multi-GPU (--gpus N, single process): every GPU holds the whole scene and renders its slice of the
// sample indices (rng_sample_offset); one RCCL sum-reduce of the accumulation buffers at the end
// (ptx_reduce_accumulation).  No window, no interaction: this is not the reference's UI.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "pth_renderer.h"

namespace {

struct SceneArg { std::string path; float off[3]; };

void usage()
{
    fprintf(stderr,
            "usage: ptx-render scene.gltf[@x,y,z] [more.gltf[@x,y,z] ...] [options]\n"
            "  --res WxH        render resolution (1280x720)\n"
            "  --spp N          samples per pixel (16)\n"
            "  --seed S         per-frame seed, fixed for every frame (0x1234567)\n"
            "  --depth D        max path depth (16, the reference's MAX_DEPTH)\n"
            "  --camera x,y,z   camera position (-1.5,1,5: render_window.rs:436)\n"
            "  --yaw DEG        rotation about +Y (0)\n"
            "  --fov DEG        vertical field of view (70)\n"
            "  --gpus N         GPUs, samples split evenly (1)\n"
            "  --out PREFIX     output prefix (render); existing files are not overwritten\n"
            "  --no-pfm         write only the PNG\n"
            "  --no-ref-quirks  world-space normals, texel x baseColorFactor, child nodes, sampler wrap + filter\n"
            "  --lit            lit integrator (emissive triangles + next-event estimation + GGX / glass); implies --no-ref-quirks\n");
}

bool parse3(const char *s, float v[3]) { return sscanf(s, "%f,%f,%f", &v[0], &v[1], &v[2]) == 3; }

// PFM: "PF" (3 channels) or "Pf" (1), width height, negative scale = little endian, rows bottom to top
void write_pfm(const std::string &path, const float *data, uint32_t w, uint32_t h, int src_stride, int channels)
{
    FILE *f = fopen(path.c_str(), "wbx");
    if (!f) throw std::runtime_error("cannot create " + path + " (exists?)");
    fprintf(f, "%s\n%u %u\n-1.0\n", channels == 3 ? "PF" : "Pf", w, h);
    std::vector<float> row((size_t)w * channels);
    for (uint32_t y = 0; y < h; ++y) {
        const float *src = data + (size_t)(h - 1 - y) * w * src_stride;
        for (uint32_t x = 0; x < w; ++x)
            for (int c = 0; c < channels; ++c) row[(size_t)x * channels + c] = src[(size_t)x * src_stride + c];
        if (fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { fclose(f); throw std::runtime_error("short write on " + path); }
    }
    fclose(f);
}

}  // namespace

int main(int argc, char **argv)
{
    std::vector<SceneArg> scenes;
    uint32_t W = 1280, H = 720, spp = 16, seed = 0x1234567u, depth = 16;
    float cam[3] = {-1.5f, 1.0f, 5.0f}, yaw = 0.0f, fov = 70.0f;
    int gpus = 1;
    bool pfm = true, ref_quirks = true, lit = false;
    std::string out = "render";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "--res") { if (sscanf(need("--res"), "%ux%u", &W, &H) != 2) { usage(); return 2; } }
        else if (a == "--spp") spp = (uint32_t)strtoul(need("--spp"), nullptr, 0);
        else if (a == "--seed") seed = (uint32_t)strtoul(need("--seed"), nullptr, 0);
        else if (a == "--depth") depth = (uint32_t)strtoul(need("--depth"), nullptr, 0);
        else if (a == "--camera") { if (!parse3(need("--camera"), cam)) { usage(); return 2; } }
        else if (a == "--yaw") yaw = (float)atof(need("--yaw"));
        else if (a == "--fov") fov = (float)atof(need("--fov"));
        else if (a == "--gpus") gpus = atoi(need("--gpus"));
        else if (a == "--out") out = need("--out");
        else if (a == "--no-pfm") pfm = false;
        else if (a == "--no-ref-quirks") ref_quirks = false;
        else if (a == "--lit") { lit = true; ref_quirks = false; }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a.size() && a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 2; }
        else {
            SceneArg s; s.off[0] = s.off[1] = s.off[2] = 0.0f;
            size_t at = a.rfind('@');
            if (at != std::string::npos && parse3(a.c_str() + at + 1, s.off)) s.path = a.substr(0, at); else s.path = a;
            scenes.push_back(s);
        }
    }
    if (scenes.empty() || !W || !H || !spp || gpus < 1 || (uint32_t)gpus > spp) { usage(); return 2; }

    try {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        std::vector<std::unique_ptr<pth::Renderer>> rs;
        for (int g = 0; g < gpus; ++g) {
            rs.emplace_back(new pth::Renderer(W, H, g));
            pth::Renderer &r = *rs.back();
            if (!ref_quirks) r.set_ref_quirks(false);
            if (lit) r.params.integrator = PTX_INTEGRATOR_LIT;
            for (const SceneArg &s : scenes) pth::Scene::import_gltf(r, s.path, pth::Vec3(s.off[0], s.off[1], s.off[2]));
            r.camera.transform.position = pth::Vec3(cam[0], cam[1], cam[2]);
            float h = yaw * (3.14159265358979323846f / 180.0f) * 0.5f;
            r.camera.transform.rotation = pth::Quat{0.0f, std::sin(h), 0.0f, std::cos(h)};
            r.camera.fov_y = fov * (3.14159265358979323846f / 180.0f);
            r.seed = seed; r.fixed_seed = true;
            r.params.max_depth = depth;
        }
        auto t1 = clk::now();
        // sample slice of GPU g: [first, first + n); the slices run concurrently (the render call is asynchronous)
        std::vector<uint32_t> count(gpus);
        uint32_t first = 0;
        for (int g = 0; g < gpus; ++g) {
            count[g] = spp / gpus + ((uint32_t)g < spp % gpus ? 1u : 0u);
            rs[g]->params.rng_sample_offset = first;
            first += count[g];
            // the reference's frames after a change are overwritten: the first is camera-dirty, the second scene-dirty
            // (the short-circuit at renderer.rs:995 leaves the scene flag set) -- settle both, then accumulate the slice,
            // so that the image holds exactly count[g] samples
            rs[g]->record_commands();
            rs[g]->record_commands();
            rs[g]->record_commands_batch(count[g]);
        }
        if (gpus > 1) {
            // equal-weight combine needs equal slice sizes up to the remainder; weight by sample count
            std::vector<ptx_context *> ctxs;
            for (auto &r : rs) ctxs.push_back(r->context());
            for (int g = 0; g < gpus; ++g) {
                // scale each mean to a sum share before the reduce: mean_g * count_g * gpus / spp, so that
                // ptx_reduce_accumulation's 1/gpus leaves sum_g(mean_g * count_g) / spp
                std::vector<float> img = rs[g]->read_image(PTX_IMAGE_ALBEDO);
                float k = (float)count[g] * (float)gpus / (float)spp;
                if (k != 1.0f) {
                    for (size_t i = 0; i < img.size(); ++i) img[i] *= k;
                    if (ptx_write_albedo(ctxs[g], img.data(), img.size() * 4)) throw std::runtime_error(ptx_last_error(ctxs[g]));
                }
            }
            if (ptx_reduce_accumulation(ctxs.data(), gpus, 0)) throw std::runtime_error(ptx_last_error(ctxs[0]));
        }
        for (auto &r : rs) if (ptx_synchronize(r->context())) throw std::runtime_error(ptx_last_error(r->context()));
        auto t2 = clk::now();

        pth::Renderer &root = *rs[0];
        root.save_render_texture_to_file(out + ".png");
        if (pfm) {
            std::vector<float> a = root.read_image(PTX_IMAGE_ALBEDO), n = root.read_image(PTX_IMAGE_NORMAL), d = root.read_image(PTX_IMAGE_DEPTH);
            write_pfm(out + "_albedo.pfm", a.data(), W, H, 4, 3);
            write_pfm(out + "_normal.pfm", n.data(), W, H, 4, 3);
            write_pfm(out + "_depth.pfm", d.data(), W, H, 1, 1);
        }
        uint64_t rays = 0, samples = 0;
        for (auto &r : rs) { ptx_stats st; ptx_get_stats(r->context(), &st); rays += st.rays; samples += st.samples; }
        double load_s = std::chrono::duration<double>(t1 - t0).count(), render_s = std::chrono::duration<double>(t2 - t1).count();
        printf("{\"width\": %u, \"height\": %u, \"spp\": %u, \"gpus\": %d, \"load_s\": %.3f, \"render_s\": %.4f, \"samples\": %llu, "
               "\"rays\": %llu, \"msamples_per_s\": %.1f, \"mrays_per_s\": %.1f, \"out\": \"%s.png\"}\n",
               W, H, spp, gpus, load_s, render_s, (unsigned long long)samples, (unsigned long long)rays,
               samples / render_s / 1e6, rays / render_s / 1e6, out.c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "ptx-render: %s\n", e.what());
        return 1;
    }
    return 0;
}
