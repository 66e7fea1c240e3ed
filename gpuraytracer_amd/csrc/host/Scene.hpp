// Scene.hpp - host-side scene class in front of the C ABI of libgpurt.so.
//
// Same public method set as the reference's Scene (Scene.hpp:13-90) for the path-tracing side:
// addWavefrontModel, loadMaterial, loadEnvironmentTexture, finalizeObjects, adaptResolution, prepare,
// traceScene, exportEXR.  Where the reference issues OpenGL calls, this class calls rt_* entry points;
// the window, the Phong preview and the wireframe renderer have no counterpart here.
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "Objects.hpp"

namespace host {

struct Object {
    explicit Object(std::string n) : name(std::move(n)) {}
    std::string name;
    std::vector<Triangle> triangles;
    int material_index = -1;
};

class Scene {
public:
    explicit Scene(int device = 0);
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // ---- CPU-only part (no GPU needed): ingest and buffer construction
    bool addWavefrontModel(const std::string &model_name);              // Scene.cpp:585-657
    // Orders the triangles (emissive first, rest by descending area) and fills the three record
    // arrays that finalizeObjects uploads (Scene.cpp:283-340).  Equal areas keep file order.
    void buildBuffers(std::vector<rt_triangle_model> &models, std::vector<rt_triangle_shading> &shadings,
                      uint32_t &lights) const;
    const std::vector<rt_material> &materials() const { return m_materials; }
    const std::vector<std::string> &materialNames() const { return m_material_names; }

    // ---- GPU part
    bool finalizeObjects();                                             // Scene.cpp:382-395
    bool loadMaterial(const std::string &name, uint32_t material_id);   // Scene.cpp:206-229: <name>_albedo/_normal/_arm.exr
    bool loadEnvironmentTexture(const std::string &texture_name);       // Scene.cpp:104-160
    bool setEnvironment(uint32_t w, uint32_t h, const float *rgba);     // environment from memory (no file)
    bool adaptResolution(int width, int height);                        // Scene.cpp:397-421: the still target and the preview target
    // CAMERA is column-major (glm::fmat4); moving = the reference's low-resolution preview mode (its own target of 240
    // rows, RECURSION 3 instead of 6, Scene.cpp:449-463): width and height come back as that target's size, and the
    // still target keeps what it has accumulated.
    bool prepare(int &width, int &height, bool moving, const float camera[16], uint32_t clock);   // Scene.cpp:435-464
    bool traceScene(uint32_t width, uint32_t height, uint32_t sample);  // Scene.cpp:423-433
    bool traceSamples(uint32_t sample_begin, uint32_t sample_count, uint32_t stride = 1);
    // Adds what `other` (same scene and target on this or another GPU) has rendered to this scene's sums; the reference
    // has one GPU and no counterpart.
    bool addSumsOf(Scene &other);
    bool exportEXR(const char *name);                                   // Scene.cpp:496-509 (row 0 = TEXEL.y 0)
    bool exportPNG8(const char *name);                                  // sRGB transfer of displayQuad.fs:19-34, binary PPM
    uint64_t rays();
    const char *lastError() const;

    void setAtlasSize(uint32_t size) { m_atlas_size = size; }
    int recursion_still = 6, recursion_moving = 3;                      // Scene.cpp:456,462
    float exposure = 1.0f;                                              // Scene.cpp:331

private:
    Object &getObject(std::string name) { m_objects.emplace_back(std::move(name)); return m_objects.back(); }
    int getMaterialIndex(const std::string &name) const;
    rt_material &getMaterial(const std::string &name);
    bool readWFMaterial(const std::string &material_name);             // Scene.cpp:659-702

    std::vector<Object> m_objects;
    std::vector<rt_material> m_materials;
    std::vector<std::string> m_material_names;
    std::unordered_map<std::string, int> m_material_indices;
    std::vector<int32_t> m_active_textures;

    int m_device;
    rt_context *m_ctx = nullptr;
    bool m_initialized = false;
    bool m_first_prepare = true;
    int m_width = 0, m_height = 0;                  // still target
    int m_low_width = 0, m_low_height = 0;          // preview target
    bool m_moving = false;                          // which one prepare() selected
    int targetWidth() const { return m_moving ? m_low_width : m_width; }
    int targetHeight() const { return m_moving ? m_low_height : m_height; }
    uint32_t m_atlas_size = 4096;
    std::string m_error;
};

// Camera of the reference's render loop: CAMERA = translate(T) * rotate(-rot.y, Y) * rotate(-rot.x, X)
// in double, narrowed to float, column-major (RayTracer.cpp:30-31).
void cameraMatrix(const double translation[3], const double rot_xy[2], float out[16]);

}  // namespace host
