// Scene.cpp - host-side scene class in front of libgpurt.so, see Scene.hpp.
#include "Scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "exr.hpp"

namespace host {

Scene::Scene(int device) : m_device(device) {}

Scene::~Scene() {
    if (m_ctx) rt_destroy(m_ctx);
}

const char *Scene::lastError() const {
    if (!m_error.empty()) return m_error.c_str();
    return m_ctx ? rt_last_error(m_ctx) : "";
}

int Scene::getMaterialIndex(const std::string &name) const {
    const auto it = m_material_indices.find(name);
    return it == m_material_indices.end() ? -1 : it->second;
}

rt_material &Scene::getMaterial(const std::string &name) {
    const auto it = m_material_indices.find(name);
    if (it != m_material_indices.end()) return m_materials[it->second];
    m_material_indices[name] = (int)m_materials.size();
    m_material_names.push_back(name);
    m_materials.emplace_back();                       // value-initialised: all zero (ior 0 = no Fresnel term)
    memset(&m_materials.back(), 0, sizeof(rt_material));
    return m_materials.back();
}

static void strip_cr(std::string &line) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
}

// MTL: Kd -> albedo, Ks -> specular, Ke -> emission, Ni -> ior, Ns / 1000 -> roughness (Scene.cpp:659-702)
bool Scene::readWFMaterial(const std::string &material_name) {
    std::ifstream mtl(material_name + ".mtl");
    if (!mtl) return false;
    int current = -1;
    std::string line;
    while (std::getline(mtl, line)) {
        strip_cr(line);
        if (line.length() <= 3) continue;
        if (line.compare(0, 6, "newmtl") == 0) {
            getMaterial(line.length() > 7 ? line.substr(7) : std::string());
            current = getMaterialIndex(line.length() > 7 ? line.substr(7) : std::string());
        }
        if (current < 0) continue;
        rt_material &m = m_materials[current];
        if (line.compare(0, 2, "Kd") == 0) sscanf(line.c_str(), "%*s %f %f %f", &m.albedo[0], &m.albedo[1], &m.albedo[2]);
        else if (line.compare(0, 2, "Ks") == 0)
            sscanf(line.c_str(), "%*s %f %f %f", &m.specular_roughness[0], &m.specular_roughness[1], &m.specular_roughness[2]);
        else if (line.compare(0, 2, "Ke") == 0)
            sscanf(line.c_str(), "%*s %f %f %f", &m.emission_ior[0], &m.emission_ior[1], &m.emission_ior[2]);
        else if (line.compare(0, 2, "Ni") == 0) sscanf(line.c_str(), "%*s %f", &m.emission_ior[3]);
        else if (line.compare(0, 2, "Ns") == 0) {
            if (sscanf(line.c_str(), "%*s %f", &m.specular_roughness[3]) == 1) m.specular_roughness[3] /= 1000.0f;
        }
    }
    return true;
}

namespace {
struct Face { unsigned pos_i[3], tex_i[3], nrm_i[3]; };
}

// OBJ: one material per `o` block (the last usemtl wins), 1-based global indices, v/vt/vn triples only
// (Scene.cpp:585-657).
bool Scene::addWavefrontModel(const std::string &name) {
    std::ifstream obj(name + ".obj");
    if (!obj || !readWFMaterial(name)) { m_error = "cannot open " + name + ".obj/.mtl"; return false; }
    std::vector<vec3> positions, normals;
    std::vector<vec2> uvs;
    std::vector<Face> faces;
    int object = -1;
    bool ok = true;
    auto flush = [&]() {
        if (object < 0) return;
        Object &o = m_objects[object];
        for (const Face &f : faces) {
            for (int k = 0; k < 3; k++)
                if (f.pos_i[k] < 1 || f.pos_i[k] > positions.size() || f.nrm_i[k] < 1 || f.nrm_i[k] > normals.size() ||
                    f.tex_i[k] < 1 || f.tex_i[k] > uvs.size()) {
                    m_error = "face index out of range in object " + o.name;
                    ok = false;
                    return;
                }
            o.triangles.emplace_back(positions[f.pos_i[0] - 1], positions[f.pos_i[1] - 1], positions[f.pos_i[2] - 1],
                                     normals[f.nrm_i[0] - 1], normals[f.nrm_i[1] - 1], normals[f.nrm_i[2] - 1],
                                     uvs[f.tex_i[0] - 1], uvs[f.tex_i[1] - 1], uvs[f.tex_i[2] - 1],
                                     (uint32_t)o.material_index);
        }
    };
    std::string line;
    while (ok && std::getline(obj, line)) {
        strip_cr(line);
        if (line.length() <= 2) continue;
        if (line[0] == 'o') {
            flush();
            faces.clear();
            getObject(std::string(line.begin() + 2, line.end()));
            object = (int)m_objects.size() - 1;
        } else if (line[0] == 'v') {
            if (line[1] == ' ') { vec3 p; sscanf(line.c_str(), "%*s %f %f %f", &p.x, &p.y, &p.z); positions.push_back(p); }
            else if (line[1] == 't') { vec2 t; sscanf(line.c_str(), "%*s %f %f", &t.x, &t.y); uvs.push_back(t); }
            else if (line[1] == 'n') { vec3 n; sscanf(line.c_str(), "%*s %f %f %f", &n.x, &n.y, &n.z); normals.push_back(n); }
        } else if (object >= 0 && line.compare(0, 6, "usemtl") == 0) {
            m_objects[object].material_index = getMaterialIndex(line.length() > 7 ? line.substr(7) : std::string());
        } else if (line[0] == 'f') {
            Face f;
            memset(&f, 0, sizeof(f));
            sscanf(line.c_str(), "%*s %i/%i/%i %i/%i/%i %i/%i/%i", (int *)&f.pos_i[0], (int *)&f.tex_i[0], (int *)&f.nrm_i[0],
                   (int *)&f.pos_i[1], (int *)&f.tex_i[1], (int *)&f.nrm_i[1], (int *)&f.pos_i[2], (int *)&f.tex_i[2],
                   (int *)&f.nrm_i[2]);
            faces.push_back(f);
        }
    }
    if (ok) flush();
    return ok;
}

// Scene::createTrianglesBuffers (Scene.cpp:283-340): emissive triangles go to [0, lights) in object and
// face order, the others follow by descending |cross(u, v)|.  The reference sorts with std::sort, which
// leaves the order of equal keys to the C++ library; here equal keys keep file order (stable sort).
void Scene::buildBuffers(std::vector<rt_triangle_model> &models, std::vector<rt_triangle_shading> &shadings,
                         uint32_t &lights) const {
    std::vector<const Triangle *> lit, rest;
    for (const Object &o : m_objects) {
        bool emissive = false;
        if (o.material_index >= 0 && (size_t)o.material_index < m_materials.size()) {
            const float *ke = m_materials[o.material_index].emission_ior;
            emissive = (ke[0] * ke[0] + ke[1] * ke[1]) + ke[2] * ke[2] > 0.0f;
        }
        for (const Triangle &t : o.triangles) (emissive ? lit : rest).push_back(&t);
    }
    std::stable_sort(rest.begin(), rest.end(),
                     [](const Triangle *a, const Triangle *b) { return a->area_key() > b->area_key(); });
    lights = (uint32_t)lit.size();
    models.clear();
    shadings.clear();
    for (const std::vector<const Triangle *> *group : {&lit, &rest})
        for (const Triangle *t : *group) {
            models.push_back(to_model(*t));
            shadings.push_back(to_shading(*t));
        }
}

bool Scene::finalizeObjects() {
    if (m_initialized) return true;
    m_error.clear();
    std::vector<rt_triangle_model> models;
    std::vector<rt_triangle_shading> shadings;
    uint32_t lights = 0;
    buildBuffers(models, shadings, lights);
    std::cout << "Triangles: " << models.size() << "\t " << std::round(models.size() * 148 / 1024.0f * 100.0f) / 100.0f
              << " KB\n";
    if (!m_ctx && rt_create(m_device, &m_ctx) != RT_OK) { m_error = "rt_create failed (no GPU?)"; return false; }
    if (rt_set_atlas_size(m_ctx, m_atlas_size) != RT_OK) return false;
    if (rt_upload_scene(m_ctx, models.data(), shadings.data(), (uint32_t)models.size(), lights, m_materials.data(),
                        (uint32_t)m_materials.size()) != RT_OK)
        return false;
    m_active_textures.assign(m_materials.size(), 0);
    std::cout << "Material Count: " << m_materials.size() << '\n';
    m_initialized = true;
    return true;
}

bool Scene::loadMaterial(const std::string &name, uint32_t material_id) {
    if (!m_initialized || material_id >= m_materials.size()) return false;
    const char *suffix[3] = {"_albedo.exr", "_normal.exr", "_arm.exr"};
    bool all = true;
    for (uint32_t slot = 0; slot < 3; slot++) {
        int w = 0, h = 0;
        std::vector<float> rgba;
        std::string err;
        if (!readEXR(name + suffix[slot], w, h, rgba, &err)) {
            std::cerr << "Couldn't load " << name + suffix[slot] << "\n\t" << err << std::endl;
            all = false;
            continue;
        }
        std::cout << name + suffix[slot] << '\t' << w << 'x' << h << std::endl;
        if (rt_upload_atlas_layer(m_ctx, material_id, slot, (uint32_t)w, (uint32_t)h, rgba.data()) != RT_OK) all = false;
    }
    m_active_textures[material_id] = all ? 1 : 0;
    if (rt_set_has_texture(m_ctx, m_active_textures.data(), (uint32_t)m_active_textures.size()) != RT_OK) return false;
    return all;
}

bool Scene::loadEnvironmentTexture(const std::string &texture_name) {
    int w = 0, h = 0;
    std::vector<float> rgba;
    std::string err;
    if (!readEXR(texture_name, w, h, rgba, &err)) {
        std::cerr << "Couldn't load " << texture_name << "\n\t" << err << std::endl;
        return false;
    }
    std::cout << texture_name << '\t' << w << 'x' << h << std::endl;
    if (!setEnvironment((uint32_t)w, (uint32_t)h, rgba.data())) return false;
    // Scene.cpp:163-199: "<name>-irradiance.exr" is used if present, otherwise computed and written.  The path
    // tracer never samples it, so a file that is present is left alone here.
    const std::string irr_name = texture_name + "-irradiance.exr";
    int iw = 0, ih = 0;
    std::vector<float> have;
    if (!readEXR(irr_name, iw, ih, have, &err)) {
        uint32_t ow = 0, oh = 0;
        if (rt_prefilter_irradiance(m_ctx, nullptr, &ow, &oh) != RT_OK) return false;
        std::vector<float> rgb((size_t)ow * oh * 3);
        if (ow && oh) {
            if (rt_prefilter_irradiance(m_ctx, rgb.data(), &ow, &oh) != RT_OK) return false;
            if (!writeEXR(irr_name, (int)ow, (int)oh, 3, rgb.data(), &err, exrCompressionFor((int)ow, (int)oh)))
                std::cerr << "Save EXR err: " << err << std::endl;           // Scene.cpp:195-198: reported, not fatal
        }
    }
    return true;
}

bool Scene::setEnvironment(uint32_t w, uint32_t h, const float *rgba) {
    if (!m_ctx && rt_create(m_device, &m_ctx) != RT_OK) { m_error = "rt_create failed (no GPU?)"; return false; }
    return rt_upload_env(m_ctx, w, h, rgba) == RT_OK;
}

// Scene.cpp:397-421 makes two targets: renderTarget at the window's size and renderTargetLow for the moving preview.
// Here the preview target gets the size prepare() traces at in moving mode, 240 rows (the reference allocates 180 rows
// and dispatches 240: OpenGL drops the stores beyond its texture).
bool Scene::adaptResolution(int width, int height) {
    if (!m_ctx || width <= 0 || height <= 0) return false;
    if (width == m_width && height == m_height) return true;
    const int low_w = (int)(240.0 / height * width), low_h = 240;
    if (rt_select_target(m_ctx, 1) != RT_OK || rt_resize(m_ctx, (uint32_t)(low_w > 0 ? low_w : 1), (uint32_t)low_h) != RT_OK) return false;
    if (rt_select_target(m_ctx, 0) != RT_OK || rt_resize(m_ctx, (uint32_t)width, (uint32_t)height) != RT_OK) return false;
    m_moving = false;
    m_width = width; m_height = height;
    m_low_width = low_w > 0 ? low_w : 1; m_low_height = low_h;
    return true;
}

bool Scene::prepare(int &width, int &height, bool moving, const float camera[16], uint32_t clock) {
    if (!m_ctx) return false;
    if (m_first_prepare || (!moving && (width != m_width || height != m_height))) {
        if (!adaptResolution(width, height)) return false;
        m_first_prepare = false;
    }
    if (moving) {                                     // Scene.cpp:449-457: the other target, its size, RECURSION 3
        width = (int)(240.0 / height * width);
        height = 240;
    }
    // which target the next traceScene / export calls work on (glBindImageTexture(0, ...) at Scene.cpp:450, :459); the
    // one left behind keeps its sums and sample count
    if (rt_select_target(m_ctx, moving ? 1 : 0) != RT_OK) return false;
    m_moving = moving;
    return rt_set_frame(m_ctx, camera, clock, moving ? recursion_moving : recursion_still, exposure) == RT_OK;
}

bool Scene::traceScene(uint32_t width, uint32_t height, uint32_t sample) {
    if ((int)width != targetWidth() || (int)height != targetHeight()) { m_error = "traceScene: size differs from prepare()"; return false; }
    // SAMPLE 0 starts an image over: the reference's running mean multiplies what is there by 0 / 1 (raytracer.glsl:611-613)
    if (sample == 0 && rt_clear(m_ctx) != RT_OK) return false;
    return rt_trace(m_ctx, sample, 1, 1) == RT_OK;
}

bool Scene::traceSamples(uint32_t sample_begin, uint32_t sample_count, uint32_t stride) {
    return m_ctx && rt_trace(m_ctx, sample_begin, sample_count, stride) == RT_OK;
}

uint64_t Scene::rays() {
    rt_stats st;
    if (!m_ctx || rt_get_stats(m_ctx, &st) != RT_OK) return 0;
    return st.rays;
}

bool Scene::addSumsOf(Scene &other) {
    return m_ctx && other.m_ctx && rt_accumulate_from(m_ctx, other.m_ctx) == RT_OK;
}

bool Scene::exportEXR(const char *name) {
    if (!m_ctx || m_width == 0) return false;
    const int m_width = targetWidth(), m_height = targetHeight();      // of the target prepare() selected
    std::vector<float> rgb((size_t)m_width * m_height * 3);
    uint64_t n = 0;
    if (rt_read_sum(m_ctx, rgb.data(), &n) != RT_OK || n == 0) return false;
    const float inv = 1.0f / (float)n;
    for (float &v : rgb) v *= inv;
    std::string err;
    if (!writeEXR(name, m_width, m_height, 3, rgb.data(), &err, exrCompressionFor(m_width, m_height))) {
        fprintf(stderr, "Save EXR err: %s\n", err.c_str());
        return false;
    }
    return true;
}

bool Scene::exportPNG8(const char *name) {
    if (!m_ctx || m_width == 0) return false;
    const int m_width = targetWidth(), m_height = targetHeight();
    std::vector<uint8_t> rgb((size_t)m_width * m_height * 3);
    if (rt_read_srgb8(m_ctx, rgb.data()) != RT_OK) return false;
    std::ofstream f(name, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << m_width << ' ' << m_height << "\n255\n";
    for (int y = m_height - 1; y >= 0; y--)          // row 0 of the target is the bottom of the view
        f.write((const char *)&rgb[(size_t)y * m_width * 3], (std::streamsize)m_width * 3);
    return (bool)f;
}

void cameraMatrix(const double t[3], const double rot[2], float out[16]) {
    const double cy = std::cos(-rot[1]), sy = std::sin(-rot[1]), cx = std::cos(-rot[0]), sx = std::sin(-rot[0]);
    // R = Ry * Rx (column vectors), then translate
    const double R[3][3] = {{cy, sy * sx, sy * cx}, {0.0, cx, -sx}, {-sy, cy * sx, cy * cx}};
    for (int c = 0; c < 3; c++) {
        for (int r = 0; r < 3; r++) out[c * 4 + r] = (float)R[r][c];
        out[c * 4 + 3] = 0.0f;
    }
    out[12] = (float)t[0]; out[13] = (float)t[1]; out[14] = (float)t[2]; out[15] = 1.0f;
}

}  // namespace host

// ---------------------------------------------------------------- C entry points for tooling/tests
// (CPU-only: ingest and EXR I/O can be exercised without a GPU.)

extern "C" {

// Loads <base>.obj/.mtl and returns the ordered record arrays in malloc'ed memory (free with rth_free).
int rth_load_wavefront(const char *base, rt_triangle_model **models, rt_triangle_shading **shadings, uint32_t *count,
                       uint32_t *lights, rt_material **materials, uint32_t *n_materials) {
    host::Scene sc(0);
    if (!sc.addWavefrontModel(base)) return -1;
    std::vector<rt_triangle_model> m;
    std::vector<rt_triangle_shading> s;
    uint32_t l = 0;
    sc.buildBuffers(m, s, l);
    for (const rt_triangle_shading &t : s)
        if (t.material_id >= sc.materials().size()) return -2;
    *count = (uint32_t)m.size();
    *lights = l;
    *n_materials = (uint32_t)sc.materials().size();
    *models = (rt_triangle_model *)malloc(sizeof(rt_triangle_model) * (m.size() + 1));
    *shadings = (rt_triangle_shading *)malloc(sizeof(rt_triangle_shading) * (s.size() + 1));
    *materials = (rt_material *)malloc(sizeof(rt_material) * (sc.materials().size() + 1));
    memcpy(*models, m.data(), sizeof(rt_triangle_model) * m.size());
    memcpy(*shadings, s.data(), sizeof(rt_triangle_shading) * s.size());
    memcpy(*materials, sc.materials().data(), sizeof(rt_material) * sc.materials().size());
    return 0;
}
void rth_free(void *p) { free(p); }

int rth_write_exr(const char *path, int w, int h, int channels, const float *pixels) {
    std::string err;
    return host::writeEXR(path, w, h, channels, pixels, &err) ? 0 : -1;
}
// compression: 0 none, 3 ZIP, -1 what tinyexr's SaveEXR picks for the size
int rth_write_exr_compressed(const char *path, int w, int h, int channels, const float *pixels, int compression) {
    std::string err;
    if (compression < 0) compression = host::exrCompressionFor(w, h);
    return host::writeEXR(path, w, h, channels, pixels, &err, compression) ? 0 : -1;
}
// Reads an EXR into malloc'ed RGBA float32.
int rth_read_exr(const char *path, int *w, int *h, float **rgba) {
    try {
        std::vector<float> px;
        std::string err;
        if (!host::readEXR(path, *w, *h, px, &err)) return -1;
        *rgba = (float *)malloc(sizeof(float) * (px.size() + 1));
        if (!*rgba) return -1;
        memcpy(*rgba, px.data(), sizeof(float) * px.size());
        return 0;
    } catch (...) {                  // out of memory: no exception may cross the C boundary
        return -1;
    }
}
void rth_camera_matrix(const double t[3], const double rot[2], float out[16]) { host::cameraMatrix(t, rot, out); }

}  // extern "C"
