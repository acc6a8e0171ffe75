// pth_scene.cpp -- Scene buffers, Geometry::create, Material::from_gltf and Scene::import_gltf.
#include "pth_scene.h"

#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "pth_image.h"
#include "pth_json.h"
#include "pth_renderer.h"

namespace pth {

namespace {

[[noreturn]] void ptx_throw(ptx_context *ctx, const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + ptx_last_error(ctx));
}

std::string dir_of(const std::string &path)
{
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

std::vector<uint8_t> slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    std::string s = ss.str();
    return std::vector<uint8_t>(s.begin(), s.end());
}

struct AccessorView {
    const uint8_t *base = nullptr;
    size_t count = 0, stride = 0;
    int component = 0, ncomp = 0;
    bool normalized = false;
};

int comp_size(int c)
{
    switch (c) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; }
    throw std::runtime_error("glTF: unknown componentType");
}

AccessorView accessor(const Json &doc, const std::vector<std::vector<uint8_t>> &buffers, size_t idx)
{
    const Json &a = doc.at("accessors").at(idx);
    if (!a.has("bufferView")) throw std::runtime_error("glTF: sparse / empty accessors are not supported");
    const Json &bv = doc.at("bufferViews").at((size_t)a.get_int("bufferView", 0));
    AccessorView v;
    v.component = (int)a.get_int("componentType", 0);
    std::string type = a.get_string("type", "");
    v.ncomp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    if (!v.ncomp) throw std::runtime_error("glTF: unsupported accessor type " + type);
    v.count = (size_t)a.get_int("count", 0);
    size_t elem = (size_t)comp_size(v.component) * v.ncomp;
    v.stride = (size_t)bv.get_int("byteStride", 0);
    if (!v.stride) v.stride = elem;
    const Json *nj = a.find("normalized");
    v.normalized = nj && nj->type == Json::Bool && nj->b;
    size_t buf = (size_t)bv.get_int("buffer", 0);
    if (buf >= buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
    size_t start = (size_t)bv.get_int("byteOffset", 0) + (size_t)a.get_int("byteOffset", 0);
    if (v.count && start + (v.count - 1) * v.stride + elem > buffers[buf].size())
        throw std::runtime_error("glTF: accessor runs past the end of its buffer");
    v.base = buffers[buf].data() + start;
    return v;
}

float read_float(const AccessorView &v, size_t i, int c)
{
    const uint8_t *p = v.base + i * v.stride + (size_t)c * comp_size(v.component);
    switch (v.component) {
    case 5126: { float f; memcpy(&f, p, 4); return f; }
    case 5121: return v.normalized ? (float)p[0] / 255.0f : (float)p[0];
    case 5123: { uint16_t u; memcpy(&u, p, 2); return v.normalized ? (float)u / 65535.0f : (float)u; }
    default: throw std::runtime_error("glTF: unsupported attribute component type");
    }
}

uint32_t read_index(const AccessorView &v, size_t i)
{
    const uint8_t *p = v.base + i * v.stride;
    switch (v.component) {
    case 5121: return p[0];
    case 5123: { uint16_t u; memcpy(&u, p, 2); return u; }
    case 5125: { uint32_t u; memcpy(&u, p, 4); return u; }
    default: throw std::runtime_error("glTF: unsupported index component type");
    }
}

// Texture::from_gltf + Renderer::add_texture (graphics.rs:383-516, renderer.rs:867-904), cached by ident
Texture texture_from_gltf(const Json &doc, const Json &info, Renderer &renderer, const std::string &gltf_dir, bool srgb)
{
    const Json &tex = doc.at("textures").at((size_t)info.get_int("index", 0));
    const Json &image = doc.at("images").at((size_t)tex.get_int("source", 0));
    if (!image.has("uri")) throw std::runtime_error("glTF: buffer-view images are not supported");
    std::string uri = image.get_string("uri", "");
    std::string ident = "gltf_" + uri;                       // material.rs:59-61
    if (!srgb) ident += "#unorm";
    // graphics.rs:446-451 reads the glTF sampler's wrap / filter modes and then builds a default
    // sampler (nearest, clamp-to-edge, todo.md:6).  Without the reference's quirks the wrap mode is honoured
    // (glTF default: repeat, 10497), and so is the magnification filter (LINEAR = 9729, also when the
    // sampler leaves it open).
    bool repeat = false, linear = false;
    if (!renderer.ref_quirks()) {
        int64_t ws = 10497, wt = 10497, mag = 9729;
        if (tex.has("sampler")) {
            const Json &smp = doc.at("samplers").at((size_t)tex.get_int("sampler", 0));
            ws = smp.get_int("wrapS", 10497); wt = smp.get_int("wrapT", 10497);
            mag = smp.get_int("magFilter", 9729);
        }
        repeat = ws == 10497 && wt == 10497;
        if (repeat) ident += "#repeat";
        linear = mag == 9729;
        if (linear) ident += "#linear";
    }
    // gltf::image::Data::from_source decodes on every call (material.rs:64-65); the
    // renderer's ident cache then skips the upload (graphics.rs:396-398).  Decoding is
    // skipped here when the ident is cached -- same result, less work.
    if (const Texture *cached = renderer.lookup_texture(ident)) return *cached;
    Texture t;
    Image img; std::string err;
    if (!decode_image_file(gltf_dir + "/" + uri, img, err)) throw std::runtime_error(err);
    std::vector<uint8_t> rgba;
    const uint8_t *px = img.pixels.data();
    if (img.channels == 3) {                              // graphics.rs:507-516
        rgba.resize((size_t)img.width * img.height * 4);
        for (size_t i = 0; i < (size_t)img.width * img.height; ++i) {
            rgba[i * 4] = px[i * 3]; rgba[i * 4 + 1] = px[i * 3 + 1]; rgba[i * 4 + 2] = px[i * 3 + 2]; rgba[i * 4 + 3] = 255;
        }
        px = rgba.data();
    }
    t.bindless_indice = renderer.add_texture(px, img.width, img.height, srgb, repeat, linear);   // R8G8B8A8_SRGB / _UNORM
    renderer.add_texture_to_lookup_map(ident, t);
    return t;
}

// material.rs:34-90
Material material_from_gltf(const Json &doc, const Json *gltf_mat, Renderer &renderer, const std::string &gltf_dir)
{
    Material m;
    m.name = gltf_mat ? gltf_mat->get_string("name", "Unnamed") : "Unnamed";
    const Json *pbr = gltf_mat ? gltf_mat->find("pbrMetallicRoughness") : nullptr;
    if (pbr) {
        if (const Json *bc = pbr->find("baseColorFactor"))
            for (int i = 0; i < 4 && i < (int)bc->size(); ++i) m.base_color[i] = (float)bc->at((size_t)i).number(1.0);
        m.metallic = (float)pbr->get_number("metallicFactor", 1.0);
        m.roughness = (float)pbr->get_number("roughnessFactor", 1.0);
        if (const Json *bt = pbr->find("baseColorTexture")) {
            Texture t = texture_from_gltf(doc, *bt, renderer, gltf_dir, true);
            m.has_base_texture = true;
            m.base_texture_indice = t.bindless_indice;
        }
        // build extension (lit integrator): G = roughness, B = metallic; the reference never reads this texture
        if (const Json *mt = pbr->find("metallicRoughnessTexture"))
            if (!renderer.ref_quirks()) {
                // optional: without the image file the factors stand alone
                const Json &image = doc.at("images").at((size_t)doc.at("textures").at((size_t)mt->get_int("index", 0)).get_int("source", 0));
                std::string path = gltf_dir + "/" + image.get_string("uri", "");
                if (FILE *probe = fopen(path.c_str(), "rb")) {
                    fclose(probe);
                    m.mr_texture = texture_from_gltf(doc, *mt, renderer, gltf_dir, false).bindless_indice + 1u;
                }
            }
    }
    if (gltf_mat && !renderer.ref_quirks()) {
        // build extensions (lit integrator), optional like the one above: tangent-space normal map (linear) and emissive
        // map (sRGB); the reference reads neither
        struct { const char *key; uint32_t *field; bool srgb; } maps[2] = {{"normalTexture", &m.normal_texture, false},
                                                                           {"emissiveTexture", &m.emissive_texture, true}};
        for (auto &mp : maps)
            if (const Json *t = gltf_mat->find(mp.key)) {
                const Json &image = doc.at("images").at((size_t)doc.at("textures").at((size_t)t->get_int("index", 0)).get_int("source", 0));
                std::string path = gltf_dir + "/" + image.get_string("uri", "");
                if (FILE *probe = fopen(path.c_str(), "rb")) {
                    fclose(probe);
                    *mp.field = texture_from_gltf(doc, *t, renderer, gltf_dir, mp.srgb).bindless_indice + 1u;
                    if (mp.field == &m.normal_texture) m.normal_scale = (float)t->get_number("scale", 1.0);
                }
            }
    }
    if (gltf_mat) {
        // build extension: emission / transmission / ior (ignored by the reference and by its integrator)
        float strength = 1.0f;
        if (const Json *x = gltf_mat->find("extensions")) {
            if (const Json *es = x->find("KHR_materials_emissive_strength")) strength = (float)es->get_number("emissiveStrength", 1.0);
            if (const Json *tr = x->find("KHR_materials_transmission")) m.transmission = (float)tr->get_number("transmissionFactor", 0.0);
            if (const Json *io = x->find("KHR_materials_ior")) m.ior = (float)io->get_number("ior", 1.5);
        }
        if (const Json *ef = gltf_mat->find("emissiveFactor"))
            for (int i = 0; i < 3 && i < (int)ef->size(); ++i) m.emissive[i] = (float)ef->at((size_t)i).number(0.0) * strength;
    }
    ptx_material sm;
    memcpy(sm.base_color, m.base_color, sizeof sm.base_color);
    sm.has_base_texture = m.has_base_texture ? 1u : 0u;
    sm.base_texture_indice = m.base_texture_indice;
    sm.metallic = m.metallic; sm.roughness = m.roughness;
    m.shared_buffer_index = renderer.scene.add_material(sm);
    ptx_material_ext ext; memset(&ext, 0, sizeof ext);
    memcpy(ext.emissive, m.emissive, 12); ext.transmission = m.transmission; ext.ior = m.ior;
    ext.mr_texture = m.mr_texture; ext.normal_texture = m.normal_texture; ext.emissive_texture = m.emissive_texture;
    renderer.scene.set_material_ext(m.shared_buffer_index, ext);
    if (m.normal_scale != 1.0f) renderer.scene.set_material_normal_scale(m.shared_buffer_index, m.normal_scale);
    return m;
}

// gltf Node::transform().decomposed() for a TRS node; matrix nodes are decomposed the way the
// gltf crate does (translation = column 3, scale = column lengths, rotation from the rest).
void node_trs(const Json &node, Vec3 &t, Quat &r, Vec3 &s)
{
    t = Vec3(0, 0, 0); r = Quat{}; s = Vec3(1, 1, 1);
    if (const Json *m = node.find("matrix")) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = (float)m->at((size_t)i).number(0.0);
        t = Vec3(a[12], a[13], a[14]);
        Vec3 c0(a[0], a[1], a[2]), c1(a[4], a[5], a[6]), c2(a[8], a[9], a[10]);
        float sx = std::sqrt(c0.dot(c0)), sy = std::sqrt(c1.dot(c1)), sz = std::sqrt(c2.dot(c2));
        if (c0.cross(c1).dot(c2) < 0.0f) sx = -sx;
        s = Vec3(sx, sy, sz);
        c0 = c0 * (1.0f / sx); c1 = c1 * (1.0f / sy); c2 = c2 * (1.0f / sz);
        float tr = c0.x + c1.y + c2.z;
        if (tr > 0.0f) { float k = std::sqrt(tr + 1.0f) * 2.0f; r = {(c1.z - c2.y) / k, (c2.x - c0.z) / k, (c0.y - c1.x) / k, 0.25f * k}; }
        else if (c0.x > c1.y && c0.x > c2.z) { float k = std::sqrt(1.0f + c0.x - c1.y - c2.z) * 2.0f; r = {0.25f * k, (c1.x + c0.y) / k, (c2.x + c0.z) / k, (c1.z - c2.y) / k}; }
        else if (c1.y > c2.z) { float k = std::sqrt(1.0f + c1.y - c0.x - c2.z) * 2.0f; r = {(c1.x + c0.y) / k, 0.25f * k, (c2.y + c1.z) / k, (c2.x - c0.z) / k}; }
        else { float k = std::sqrt(1.0f + c2.z - c0.x - c1.y) * 2.0f; r = {(c2.x + c0.z) / k, (c2.y + c1.z) / k, 0.25f * k, (c0.y - c1.x) / k}; }
        return;
    }
    if (const Json *j = node.find("translation")) t = Vec3((float)j->at(0).number(0), (float)j->at(1).number(0), (float)j->at(2).number(0));
    if (const Json *j = node.find("rotation")) r = {(float)j->at(0).number(0), (float)j->at(1).number(0), (float)j->at(2).number(0), (float)j->at(3).number(1)};
    if (const Json *j = node.find("scale")) s = Vec3((float)j->at(0).number(1), (float)j->at(1).number(1), (float)j->at(2).number(1));
}

}  // namespace

std::pair<uint64_t, uint64_t> Scene::add_vertices(const std::vector<ptx_vertex> &v)
{
    uint64_t first = 0;
    if (ptx_add_vertices(ctx_, v.data(), v.size(), &first)) ptx_throw(ctx_, "Vertex buffer overflow");   // scene/mod.rs:281-283
    vertices.insert(vertices.end(), v.begin(), v.end());
    return {first, first + v.size()};
}

std::pair<uint64_t, uint64_t> Scene::add_indices(const std::vector<uint32_t> &i)
{
    uint64_t first = 0;
    if (ptx_add_indices(ctx_, i.data(), i.size(), &first)) ptx_throw(ctx_, "Index buffer overflow");     // scene/mod.rs:412-414
    indices.insert(indices.end(), i.begin(), i.end());
    return {first, first + i.size()};
}

void Scene::set_material_ext(uint64_t index, const ptx_material_ext &e)
{
    if (ptx_set_material_ext(ctx_, (uint32_t)index, &e)) ptx_throw(ctx_, "set_material_ext");
    if (material_ext.size() <= index) material_ext.resize(index + 1);
    material_ext[index] = e;
}

void Scene::set_material_normal_scale(uint64_t index, float scale)
{
    if (ptx_set_material_normal_scale(ctx_, (uint32_t)index, scale)) ptx_throw(ctx_, "set_material_normal_scale");
}

uint64_t Scene::add_material(const ptx_material &m)
{
    uint32_t idx = 0;
    if (ptx_add_material(ctx_, &m, &idx)) ptx_throw(ctx_, "Material buffer overflow");                   // scene/mod.rs:347-349
    materials.push_back(m);
    return idx;
}

const Geometry *Scene::get_geometry(const std::string &ident) const
{
    auto it = geometries_.find(ident);
    return it == geometries_.end() ? nullptr : &it->second;
}

Geometry Scene::create_geometry(const std::string &ident, const std::vector<CustomVertex> &verts,
                                const std::vector<uint32_t> &idx)
{
    if (const Geometry *g = get_geometry(ident)) return *g;      // geometry/mod.rs:62-65: first one wins
    std::vector<ptx_vertex> sv(verts.size());
    for (size_t i = 0; i < verts.size(); ++i) {                  // Vertex::from_custom_vertex, renderer.rs:102-110
        memset(&sv[i], 0, sizeof(ptx_vertex));
        memcpy(sv[i].position, verts[i].position, 12);
        memcpy(sv[i].normal, verts[i].normal, 12);
        memcpy(sv[i].uv, verts[i].tex_coords, 8);
    }
    auto vr = add_vertices(sv);
    auto ir = add_indices(idx);
    Geometry g;
    if (ptx_create_geometry(ctx_, vr.first, vr.second - vr.first, ir.first, ir.second - ir.first, &g.handle))
        ptx_throw(ctx_, "Geometry::create");
    g.vertex_count = (uint32_t)verts.size(); g.index_count = (uint32_t)idx.size();
    g.shared_vertex_buffer_offset = vr.first; g.shared_vertex_buffer_count = vr.second - vr.first;
    g.shared_index_buffer_offset = ir.first; g.shared_index_buffer_count = ir.second - ir.first;
    add_geometry(ident, g);
    return g;
}

std::vector<Vec3> calculate_normals(const std::vector<Vec3> &positions, const std::vector<uint32_t> &indices)
{
    std::vector<Vec3> normals(positions.size(), Vec3(0, 0, 0));
    for (size_t i = 0; i + 2 < indices.size(); i += 3) {
        Vec3 a = positions[indices[i]], b = positions[indices[i + 1]], c = positions[indices[i + 2]];
        Vec3 n = (b - a).cross(c - a).normalize();
        normals[indices[i]] = normals[indices[i]] + n;
        normals[indices[i + 1]] = normals[indices[i + 1]] + n;
        normals[indices[i + 2]] = normals[indices[i + 2]] + n;
    }
    for (Vec3 &n : normals) n = n.normalize();
    return normals;
}

void Scene::import_gltf(Renderer &renderer, const std::string &path, Vec3 position_offset)
{
    std::vector<uint8_t> text = slurp(path);
    std::string s(text.begin(), text.end());
    Json doc = JsonParser(s).parse();
    std::string gltf_dir = dir_of(path);
    std::vector<std::vector<uint8_t>> buffers;
    if (const Json *bj = doc.find("buffers"))
        for (size_t i = 0; i < bj->size(); ++i) {
            const Json &b = bj->at(i);
            if (!b.has("uri")) throw std::runtime_error("glTF: GLB / embedded buffers are not supported");
            std::string uri = b.get_string("uri", "");
            if (uri.compare(0, 5, "data:") == 0) throw std::runtime_error("glTF: data: URIs are not supported");
            buffers.push_back(slurp(gltf_dir + "/" + uri));
        }
    const Json *scenes = doc.find("scenes");
    if (!scenes) return;
    for (size_t si = 0; si < scenes->size(); ++si) {
        const Json *nodes = scenes->at(si).find("nodes");
        if (!nodes) continue;
        // root nodes only (gltf.rs:32-34); without the reference's quirks the children follow their parent
        // depth first with the composed transform
        struct Work { size_t node; bool has_parent; Vec3 pt; Quat pr; Vec3 ps; };
        std::vector<Work> work;
        for (size_t ni = nodes->size(); ni-- > 0;) work.push_back({(size_t)nodes->at(ni).number(0), false, Vec3(0, 0, 0), Quat{}, Vec3(1, 1, 1)});
        while (!work.empty()) {
            Work w = work.back(); work.pop_back();
            const Json &node = doc.at("nodes").at(w.node);
            Vec3 t, sc; Quat r;
            node_trs(node, t, r, sc);
            if (!renderer.ref_quirks()) {
                if (w.has_parent) {     // parent * child for TRS triples (exact for uniformly scaled parents)
                    t = w.pt + w.pr.mul_vec3(Vec3(w.ps.x * t.x, w.ps.y * t.y, w.ps.z * t.z));
                    r = w.pr.mul(r);
                    sc = Vec3(w.ps.x * sc.x, w.ps.y * sc.y, w.ps.z * sc.z);
                }
                if (const Json *ch = node.find("children"))
                    for (size_t ci = ch->size(); ci-- > 0;) work.push_back({(size_t)ch->at(ci).number(0), true, t, r, sc});
            }
            if (!node.has("mesh")) continue;
            size_t mesh_index = (size_t)node.get_int("mesh", 0);
            const Json &mesh = doc.at("meshes").at(mesh_index);
            Transform transform;
            transform.position = t + position_offset;                            // gltf.rs:39-41
            transform.rotation = r;
            transform.scale = sc;
            const Json &prims = mesh.at("primitives");
            for (size_t pi = 0; pi < prims.size(); ++pi) {
                const Json &prim = prims.at(pi);
                const Json *mat_json = nullptr;
                if (prim.has("material")) mat_json = &doc.at("materials").at((size_t)prim.get_int("material", 0));
                Material material = material_from_gltf(doc, mat_json, renderer, gltf_dir);
                const Json &attrs = prim.at("attributes");
                if (!attrs.has("POSITION")) throw std::runtime_error("Mesh has no positions");          // gltf.rs:74
                AccessorView pv = accessor(doc, buffers, (size_t)attrs.get_int("POSITION", 0));
                std::vector<Vec3> positions(pv.count);
                for (size_t i = 0; i < pv.count; ++i) positions[i] = Vec3(read_float(pv, i, 0), read_float(pv, i, 1), read_float(pv, i, 2));
                std::vector<uint32_t> idx;
                if (prim.has("indices")) {                                        // gltf.rs:60-68
                    AccessorView iv = accessor(doc, buffers, (size_t)prim.get_int("indices", 0));
                    idx.resize(iv.count);
                    for (size_t i = 0; i < iv.count; ++i) idx[i] = read_index(iv, i);
                } else {
                    idx.resize(positions.size());
                    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
                }
                for (uint32_t v : idx) if (v >= positions.size()) throw std::runtime_error("glTF: index out of range");
                std::vector<Vec3> normals;
                if (attrs.has("NORMAL")) {                                        // gltf.rs:78-91
                    AccessorView nv = accessor(doc, buffers, (size_t)attrs.get_int("NORMAL", 0));
                    normals.resize(nv.count);
                    for (size_t i = 0; i < nv.count; ++i) normals[i] = Vec3(read_float(nv, i, 0), read_float(nv, i, 1), read_float(nv, i, 2));
                } else {
                    normals = calculate_normals(positions, idx);
                }
                std::vector<float> uvs;
                if (attrs.has("TEXCOORD_0")) {                                    // gltf.rs:94-98
                    AccessorView uv = accessor(doc, buffers, (size_t)attrs.get_int("TEXCOORD_0", 0));
                    uvs.resize(uv.count * 2);
                    for (size_t i = 0; i < uv.count; ++i) { uvs[i * 2] = read_float(uv, i, 0); uvs[i * 2 + 1] = read_float(uv, i, 1); }
                } else {
                    uvs.assign(positions.size() * 2, 0.0f);
                }
                size_t n = std::min(positions.size(), std::min(normals.size(), uvs.size() / 2));   // zip()
                std::vector<CustomVertex> verts(n);
                for (size_t i = 0; i < n; ++i) {
                    verts[i].position[0] = positions[i].x; verts[i].position[1] = positions[i].y; verts[i].position[2] = positions[i].z;
                    verts[i].normal[0] = normals[i].x; verts[i].normal[1] = normals[i].y; verts[i].normal[2] = normals[i].z;
                    verts[i].tex_coords[0] = uvs[i * 2]; verts[i].tex_coords[1] = uvs[i * 2 + 1];
                }
                for (uint32_t v : idx) if (v >= n) throw std::runtime_error("glTF: index refers past the zipped vertices");
                std::string mesh_name = mesh.has("name") ? mesh.get_string("name", "") : std::to_string(mesh_index);
                std::string ident = mesh_name + "_" + std::to_string(pi);         // gltf.rs:112-116
                Geometry geometry = renderer.scene.create_geometry(ident, verts, idx);
                renderer.scene.push(Entity{geometry, transform, material});       // gltf.rs:122
            }
        }
    }
}

}  // namespace pth
