// rt_oracle.cpp - CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker the HIP path is compared against.  Nothing in the product
// (libgpurt.so, the gpuraytracer_amd package) may link, import or call it; only tests/,
// __graft_entry__.smoke() and the cpu_baseline leg of bench.py do.
//
// It follows res/shader/raytracer.glsl of the reference function by function (each
// function below cites the lines it restates) in IEEE float32, with no floating-point
// contraction (build with -ffp-contract=off).
//
// PINNED AGAINST THE REFERENCE: the reference ships no tests or golden images for this path,
// but its unmodified shader runs in the build container under Mesa 23.2 llvmpipe
// (oracle/glsl_ref/run_reference_glsl.c).  Where GLSL leaves the arithmetic to the GL
// implementation (association order of dot() and of matrix products, fused or separate
// multiply-add, x/y or x*(1/y), the elementary functions) this restatement follows what Mesa
// generates for this shader, so that hit ids, (u, v, t) and whole images can be compared with
// the llvmpipe run; tests/golden/ holds those outputs and tests/test_reference_glsl.py checks them.
// Each such choice is marked "Mesa:" below.
//
// Two math back ends, chosen at compile time:
//   ORC_LIBM = 1  elementary functions from the C library (independent of the product)
//   ORC_LIBM = 0  the functions of glslmath_pinned.h (bit-identical to llvmpipe's: the checker's own copy of those
//                 operation sequences, not the header the device code includes), so that whole images can be
//                 compared bit for bit
//
// Places where the shader's behaviour is undefined and this restatement defines it:
//   * sampleLight reads triangleModels[light_id - 1] with light_id possibly 0, and
//     triangleShadings[light_id] with light_id possibly LIGHTS == COUNT
//     (raytracer.glsl:336-338, :379): out-of-range reads return an all-zero record.
//   * `out vec3 light` left unwritten on a false return (:333-334, :369-370, :376-377,
//     :420-425).  The value still reaches global_light and through it the Russian-roulette
//     brightness (:557, :575).  Mesa: after inlining, each of the two functions keeps one `light`
//     variable for the whole pixel-sample, so an unwritten result is the value that function wrote
//     last at an earlier depth (zero before its first write); this restatement does the same.
//   * asin / acos arguments outside [-1, 1] (:274, :527-528): NaN with glslmath_pinned.h (as Mesa), clamped
//     with the C library back end; a non-finite texture coordinate samples texel (0, 0).

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#ifndef ORC_LIBM
#define ORC_LIBM 0
#endif
#ifndef ORC_DOT_ORDER
#define ORC_DOT_ORDER 1   // 0: (x*x' + y*y') + z*z'    1: (z*z' + y*y') + x*x'  (Mesa: the last component first)
#endif

#if !ORC_LIBM
#include "glslmath_pinned.h"
#endif

namespace {

// ---------------------------------------------------------------- records (SSBO layouts)

struct TriModel {            // raytracer.glsl:13-18, 3Dobjects.hpp:102-113
    float n[3], p[3], u[3], v[3];
};
struct TriShading {          // raytracer.glsl:20-29, 3Dobjects.hpp:115-135
    uint32_t material_id;
    float normals[3][3];
    float tangents[3][3];
    float tex_p[2], tex_u[2], tex_v[2];
};
struct Material {            // raytracer.glsl:36-40, Material.hpp:9-15
    float albedo[4], specular_roughness[4], emission_ior[4];
};
static_assert(sizeof(TriModel) == 48, "layout");
static_assert(sizeof(TriShading) == 100, "layout");
static_assert(sizeof(Material) == 48, "layout");

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }

inline float dot(V3 a, V3 b) {
#if ORC_DOT_ORDER == 0
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
#else
    return (a.z * b.z + a.y * b.y) + a.x * b.x;
#endif
}
// GLSL cross(): (x1*y2 - y1*x2, x2*y0 - y2*x0, x0*y1 - y0*x1)
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
// normalize(v) is fixed as v * (1 / sqrt(dot(v, v))).
inline V3 normalize(V3 a) {
    const float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
// GLSL reflect(I, N) = I - 2 * dot(N, I) * N
inline V3 reflect(V3 I, V3 N) {
    const float k = 2.0f * dot(N, I);
    return I - N * k;
}
// Mesa: max/min are IEEE maxNum/minNum (a NaN operand yields the other operand)
inline float gmax(float x, float y) { return fmaxf(x, y); }
inline float gmin(float x, float y) { return fminf(x, y); }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

// Elementary functions.
#if ORC_LIBM
inline void m_sincos(float x, float &s, float &c) { s = sinf(x); c = cosf(x); }
inline float m_sin(float x) { return sinf(x); }
inline float m_tan(float x) { return tanf(x); }
inline float m_acos(float x) { return acosf(gclamp(x, -1.0f, 1.0f)); }
inline float m_asin(float x) { return asinf(gclamp(x, -1.0f, 1.0f)); }
inline float m_atan2(float y, float x) { return atan2f(y, x); }
inline float m_pow5(float x) { return powf(x, 5.0f); }
#else
inline void m_sincos(float x, float &s, float &c) { gm::sincos(x, s, c); }
inline float m_sin(float x) { return gm::sin(x); }
inline float m_tan(float x) { return gm::tan(x); }
inline float m_acos(float x) { return gm::acos(x); }
inline float m_asin(float x) { return gm::asin(x); }
inline float m_atan2(float y, float x) { return gm::atan2(y, x); }
inline float m_pow5(float x) { return gm::pow(x, 5.0f); }
#endif

const float PI = 3.141592653589793f;
const float TWO_PI = 6.283185307179586f;
const float INV_PI = (float)(1.0 / 3.141592653589793);   // raytracer.glsl:79, constant-folded
const float INV_UINT_MAX = 1.0f / 4294967296.0f;         // :86, float(0xFFFFFFFFu) rounds to 2^32

// ---------------------------------------------------------------- scene

struct Texture2D {            // RGBA32F, LINEAR, REPEAT (Scene.cpp:126-133)
    uint32_t w = 0, h = 0;
    std::vector<float> rgba;
};

struct Scene {
    std::vector<TriModel> models;
    std::vector<TriShading> shadings;
    std::vector<Material> materials;
    std::vector<int32_t> has_texture;
    uint32_t count = 0, lights = 0;
    uint32_t atlas_size = 4096;                      // Scene.cpp:365
    std::vector<std::vector<uint32_t>> atlas;        // layer -> packed R11G11B10F texels, empty = all zero
    Texture2D env;
};

struct Frame {
    float camera[16];      // column-major (Scene.cpp:446)
    uint32_t clock;
    int32_t recursion;
    float exposure;
    uint32_t width, height;
};

struct Stats {
    uint64_t pixel_samples, rays, shadow_probes, glossy_probes, closest_tests, probe_tests;
};

struct Ray { V3 position, direction; };

// ORC_DEBUG=1 in the environment prints one line per bounce to stderr
inline bool orc_debug() {
    static const bool on = getenv("ORC_DEBUG") != nullptr;
    return on;
}

const TriModel ZERO_MODEL = {};
const TriShading ZERO_SHADING = {};

inline const TriModel &model_at(const Scene &sc, int i) {
    return (i >= 0 && (uint32_t)i < sc.count) ? sc.models[i] : ZERO_MODEL;
}
inline const TriShading &shading_at(const Scene &sc, int i) {
    return (i >= 0 && (uint32_t)i < sc.count) ? sc.shadings[i] : ZERO_SHADING;
}
inline const Material &material_at(const Scene &sc, uint32_t i) {
    static const Material zero = {};
    return (i < sc.materials.size()) ? sc.materials[i] : zero;
}

// ---------------------------------------------------------------- RNG  (raytracer.glsl:83-98)

inline uint32_t pcgHash(uint32_t k) {                       // :89-93
    const uint32_t state = k * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
inline float unitFloat(uint32_t &seed) {                    // :95-98
    seed = pcgHash(seed);
    return (float)seed * INV_UINT_MAX;
}
inline V3 randomSphere(uint32_t &seed) {                    // :100-113
    const float theta = TWO_PI * unitFloat(seed);
    const float x = unitFloat(seed) * 2.0f - 1.0f;
    const float sinx = sqrtf(1.0f - x * x);
    float s, c;
    m_sincos(theta, s, c);
    return V3{sinx * c, sinx * s, x};
}

// ---------------------------------------------------------------- intersection

// intersectTriangle, raytracer.glsl:125-162
inline bool intersectTriangle(const Ray &ray, const TriModel &tri, V4 &isec) {
    const V3 true_normal = v3(tri.n);
    const float determinant = -dot(ray.direction, true_normal);
    if (determinant <= 0.0f) return false;

    const V3 delta = ray.position - v3(tri.p);
    const float relative_depth = dot(true_normal, delta);
    if (relative_depth <= 0.0f) return false;

    if (isec.z >= 0.0f && relative_depth * isec.w > isec.z * determinant) return false;

    const V3 minor = cross(ray.direction, delta);
    const float rx = -dot(minor, v3(tri.v));
    const float ry = dot(minor, v3(tri.u));
    if (rx < 0.0f || ry < 0.0f || rx + ry > determinant) return false;

    isec = V4{rx, ry, relative_depth, determinant};
    return true;
}

// intersectTriangleShadow, raytracer.glsl:164-196
inline bool intersectTriangleShadow(const Ray &ray, const TriModel &tri, float max_t) {
    const V3 true_normal = v3(tri.n);
    const float determinant = dot(ray.direction, true_normal);
    if (determinant <= 0.0f) return false;

    const V3 delta = v3(tri.p) - ray.position;
    const float relative_depth = dot(true_normal, delta);
    if (relative_depth <= 0.0f) return false;

    if (relative_depth > determinant * max_t) return false;

    const V3 minor = cross(ray.direction, delta);
    const float rx = -dot(minor, v3(tri.v));
    const float ry = dot(minor, v3(tri.u));
    if (rx < 0.0f || ry < 0.0f || rx + ry > determinant) return false;
    return true;
}

// findIntersection, raytracer.glsl:217-235
inline int findIntersection(const Scene &sc, const Ray &ray, V3 &current_intersection) {
    int current_tri = -1;
    V4 isec = V4{0.0f, 0.0f, -1.0f, 1.0f};
    const int count = (int)sc.count;
    for (int triID = 0; triID < count; triID++)
        if (intersectTriangle(ray, sc.models[triID], isec)) current_tri = triID;
    const float inv_det = 1.0f / isec.w;
    current_intersection = V3{isec.x * inv_det, isec.y * inv_det, isec.z * inv_det};
    return current_tri;
}

// getTransmission, raytracer.glsl:198-215
inline float getTransmission(float cosThetaI, float etaI, float etaT) {
    const float sinThetaI = sqrtf(1.0f - cosThetaI * cosThetaI);
    const float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1.0f) return 1.0f;
    const float cosThetaT = sqrtf(1.0f - sinThetaT * sinThetaT);
    const float r0 = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    const float r1 = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
#if ORC_DOT_ORDER == 0
    return (r0 * r0 + r1 * r1) * 0.5f;
#else
    return (r1 * r1 + r0 * r0) * 0.5f;
#endif
}

// calculateN, raytracer.glsl:239-248: normalize(N * vec3(1-u-v, u, v)), mat3 * vec3 summed column by column
inline V3 calculateN(const TriShading &tri, V3 isec) {
    const float w0 = 1.0f - isec.x - isec.y;
    const V3 n = (v3(tri.normals[0]) * w0 + v3(tri.normals[1]) * isec.x) + v3(tri.normals[2]) * isec.y;
    return normalize(n);
}
// calculateUV, raytracer.glsl:250-259
inline void calculateUV(const TriShading &tri, V3 isec, float &s, float &t) {
    s = (tri.tex_p[0] * 1.0f + tri.tex_u[0] * isec.x) + tri.tex_v[0] * isec.y;
    t = (tri.tex_p[1] * 1.0f + tri.tex_u[1] * isec.x) + tri.tex_v[1] * isec.y;
}
// calculateTBN, raytracer.glsl:261-270; returns the three columns T, cross(T, N), N
inline void calculateTBN(const TriShading &tri, V3 N, V3 isec, V3 &T, V3 &Bt) {
    const float w0 = 1.0f - isec.x - isec.y;
    V3 t = (v3(tri.tangents[0]) * w0 + v3(tri.tangents[1]) * isec.x) + v3(tri.tangents[2]) * isec.y;
    t = normalize(t - N * dot(N, t));
    T = t;
    Bt = cross(t, N);
}

// ---------------------------------------------------------------- textures

// Unsigned small floats of GL_R11F_G11F_B10F: 5 exponent bits (bias 15), `mbits` mantissa bits.
inline float uf_decode(uint32_t code, int mbits) {
    const uint32_t e = code >> mbits, m = code & ((1u << mbits) - 1u);
    uint32_t bits;
    if (e == 0) return (float)m * (1.0f / (float)(1u << (14 + mbits)));
    if (e == 31) bits = 0x7F800000u | (m << (23 - mbits));
    else bits = ((e + 112u) << 23) | (m << (23 - mbits));
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
// Encoding (the conversion of the texture upload, Scene.cpp:93-98; OpenGL allows either neighbouring value).
// Mesa: negative -> 0, NaN -> NaN, +Inf -> Inf, above the largest finite value -> the largest finite value,
// below 2^-14 -> 0, otherwise the mantissa is truncated.
inline uint32_t uf_encode(float f, int mbits) {
    uint32_t bits;
    memcpy(&bits, &f, 4);
    const uint32_t sign = bits >> 31, ex = (bits >> 23) & 0xFFu, man = bits & 0x7FFFFFu;
    const uint32_t max_finite = (30u << mbits) | ((1u << mbits) - 1u);
    if (ex == 255) {
        if (man) return (31u << mbits) | 1u;
        return sign ? 0u : (31u << mbits);
    }
    if (sign) return 0u;
    if (f > (mbits == 6 ? 65024.0f : 64512.0f)) return max_finite;
    const int E = (int)ex - 127;
    if (E > -15) return ((uint32_t)(E + 15) << mbits) | (man >> (23 - mbits));
    return 0u;
}
inline uint32_t pack_r11g11b10(float r, float g, float b) {
    return uf_encode(r, 6) | (uf_encode(g, 6) << 11) | (uf_encode(b, 5) << 22);
}
inline V3 unpack_r11g11b10(uint32_t p) {
    return V3{uf_decode(p & 0x7FFu, 6), uf_decode((p >> 11) & 0x7FFu, 6), uf_decode(p >> 22, 5)};
}

// Bilinear filtering with REPEAT wrapping.  Mesa (llvmpipe), per axis of length L:
//   L a power of two:  x = c*L - 0.5;  i = floor(x);  texels i & (L-1), (i+1) & (L-1)
//   otherwise:         c = c - floor(c);  x = c*L - 0.5;  i = floor(x);  texels (x < 0 ? L-1 : i), (i+1 == L ? 0 : i+1)
//   weight w = x - floor(x);  lerp(w, v0, v1) = fma(w, v1 - v0, v0), first along s for both rows, then along t.
// A coordinate whose texel index does not fit an int (NaN included) uses index 0; its weight follows from
// the same arithmetic (NaN for NaN and infinities, which makes the sample NaN).
struct Bilinear { int i0, i1, j0, j1; float a, b; };
inline void wrap_repeat_linear(float c, uint32_t L, int &i0, int &i1, float &w) {
    const int n = (int)L;
    const bool pot = (L & (L - 1u)) == 0u;
    if (!pot) c = c - floorf(c);
    const float x = c * (float)L - 0.5f;
    const float fl = floorf(x);
    w = x - fl;
    const bool in_range = fabsf(fl) < 2147483648.0f;
    const int i = in_range ? (int)fl : 0;
    if (pot) {
        i0 = i & (n - 1);
        i1 = (i + 1) & (n - 1);
    } else {
        i0 = (x < 0.0f) ? n - 1 : i;
        i1 = (i + 1 >= n) ? 0 : i + 1;
        if (!in_range || i0 >= n) { i0 = 0; i1 = 0; }
    }
}
inline Bilinear bilinear_setup(float s, float t, uint32_t W, uint32_t H) {
    Bilinear bl;
    wrap_repeat_linear(s, W, bl.i0, bl.i1, bl.a);
    wrap_repeat_linear(t, H, bl.j0, bl.j1, bl.b);
    return bl;
}
inline float lerp1(float w, float v0, float v1) { return fmaf(w, v1 - v0, v0); }
inline V3 bilinear_mix(V3 t00, V3 t10, V3 t01, V3 t11, float a, float b) {
    const V3 top = V3{lerp1(a, t00.x, t10.x), lerp1(a, t00.y, t10.y), lerp1(a, t00.z, t10.z)};
    const V3 bot = V3{lerp1(a, t01.x, t11.x), lerp1(a, t01.y, t11.y), lerp1(a, t01.z, t11.z)};
    return V3{lerp1(b, top.x, bot.x), lerp1(b, top.y, bot.y), lerp1(b, top.z, bot.z)};
}

inline V3 sampleEnv(const Scene &sc, float s, float t) {
    const Texture2D &tx = sc.env;
    if (tx.w == 0 || tx.h == 0) return V3{0, 0, 0};
    const Bilinear bl = bilinear_setup(s, t, tx.w, tx.h);
    auto at = [&](int i, int j) { return v3(&tx.rgba[((size_t)j * tx.w + (size_t)i) * 4]); };
    return bilinear_mix(at(bl.i0, bl.j0), at(bl.i1, bl.j0), at(bl.i0, bl.j1), at(bl.i1, bl.j1), bl.a, bl.b);
}

inline V3 sampleAtlas(const Scene &sc, float s, float t, uint32_t layer) {
    if (layer >= sc.atlas.size() || sc.atlas[layer].empty()) return V3{0, 0, 0};
    const std::vector<uint32_t> &tx = sc.atlas[layer];
    const uint32_t S = sc.atlas_size;
    const Bilinear bl = bilinear_setup(s, t, S, S);
    auto at = [&](int i, int j) { return unpack_r11g11b10(tx[(size_t)j * S + (size_t)i]); };
    return bilinear_mix(at(bl.i0, bl.j0), at(bl.i1, bl.j0), at(bl.i0, bl.j1), at(bl.i1, bl.j1), bl.a, bl.b);
}

// skyColor, raytracer.glsl:273-276
inline V3 skyColor(const Scene &sc, const Frame &fr, V3 direction) {
    const float s = m_atan2(direction.z, direction.x) * INV_PI * 0.5f + 0.5f;
    const float t = -m_asin(direction.y) * INV_PI + 0.5f;
    if (orc_debug()) fprintf(stderr, "  sky: direction.y %.9g asin %.9g\n", direction.y, m_asin(direction.y));
    return sampleEnv(sc, s, t) * fr.exposure;
}

// ---------------------------------------------------------------- light probes

// sampleLight, raytracer.glsl:326-388
inline bool sampleLight(const Scene &sc, V3 position, V3 normal, int current, uint32_t &seed, V3 &light,
                        Stats &st) {
    const int LIGHTS = (int)sc.lights;
    if (LIGHTS == 0) return false;

    const int light_id = (int)(unitFloat(seed) * (float)LIGHTS);
    const TriModel &light_tri = model_at(sc, light_id - 1);          // sic, :338
    const V3 U = v3(light_tri.u), V = v3(light_tri.v), P = v3(light_tri.p);

    float cx = unitFloat(seed);
    float cy = unitFloat(seed);
    if (cx + cy > 1.0f) { cx = 1.0f - cx; cy = 1.0f - cy; }

    const V3 rand_tri = (U * cx + P) + V * cy;     // UVP * vec3(coords, 1.0); Mesa: the constant-1 column is added second
    const V3 delta = rand_tri - position;
    const float max_t = length(delta);

    Ray probe;
    probe.position = position;
    probe.direction = V3{delta.x / max_t, delta.y / max_t, delta.z / max_t};

    const V3 true_normal = v3(light_tri.n);
    const float DdN = dot(probe.direction, normal);
    const float DdL = dot(probe.direction, true_normal);
    if (DdN <= 0.0f || DdL >= 0.0f) return false;

    st.shadow_probes++;
    const int count = (int)sc.count;
    if (orc_debug())
        fprintf(stderr, "  shadow probe from %.9g %.9g %.9g dir %.9g %.9g %.9g max_t %.9g light %d\n", position.x,
                position.y, position.z, probe.direction.x, probe.direction.y, probe.direction.z, max_t, light_id);
    for (int triID = 0; triID < count; triID++) {
        if (triID == current) continue;
        if (intersectTriangleShadow(probe, sc.models[triID], max_t)) {
            if (orc_debug()) fprintf(stderr, "  shadow probe stopped by %d\n", triID);
            return false;
        }
    }
    const Material &material = material_at(sc, shading_at(sc, light_id).material_id);   // :379

    const float tri_area = 0.5f * length(cross(U, V));
    const float light_vis_area_ratio = -DdL;
    const float probe_ratio = DdN;
    const float distance_ratio = max_t * (max_t + 2.0f) + 1.0f;     // fma() of :385; Mesa: separate multiply and add
    const float k = tri_area * light_vis_area_ratio * probe_ratio / distance_ratio * (float)LIGHTS;
    light = v3(material.emission_ior) * k;
    return true;
}

// sampleLightGlossy, raytracer.glsl:391-438
// `view_delta` is position - view.position.  Mesa: position is view.position + direction * t, and the compiler
// simplifies the difference to direction * t, so that is what the caller passes.
inline bool sampleLightGlossy(const Scene &sc, const Frame &fr, V3 position, V3 normal, V3 view_delta,
                              float roughness, int current, uint32_t &seed, V3 &light, Stats &st) {
    const V3 viewDir = normalize(view_delta);
    Ray probe;
    probe.position = position;
    probe.direction = reflect(viewDir, normalize(randomSphere(seed) * roughness + normal));

    V4 isec = V4{0.0f, 0.0f, -1.0f, 0.0f};
    int lightID = -1;
    st.glossy_probes++;
    const int count = (int)sc.count, LIGHTS = (int)sc.lights;
    for (int triID = 0; triID < count; triID++) {
        if (triID == current) continue;
        if (intersectTriangle(probe, sc.models[triID], isec)) {
            if (orc_debug()) fprintf(stderr, "  glossy probe takes %d: %.9g %.9g %.9g %.9g\n", triID, isec.x, isec.y, isec.z, isec.w);
            if (triID < LIGHTS) lightID = triID;
            else { lightID = -1; break; }
        }
    }
    if (orc_debug())
        fprintf(stderr, "  glossy probe from %.9g %.9g %.9g dir %.9g %.9g %.9g light %d\n", position.x, position.y,
                position.z, probe.direction.x, probe.direction.y, probe.direction.z, lightID);
    if (lightID < 0) {
        if (isec.z < 0.0f) light = skyColor(sc, fr, probe.direction);
        return isec.z < 0.0f;
    }
    const float inv_det = 1.0f / isec.w;
    const V3 hit = V3{isec.x * inv_det, isec.y * inv_det, isec.z * inv_det};
    const TriShading &light_tri = sc.shadings[lightID];
    const Material &material = material_at(sc, light_tri.material_id);
    const V3 N = calculateN(light_tri, hit);
    light = v3(material.emission_ior) * gmax(dot(probe.direction, N), 0.0f);
    return true;
}

// ---------------------------------------------------------------- trace, raytracer.glsl:440-584

V3 trace(const Scene &sc, const Frame &fr, Ray ray, uint32_t seed, uint32_t light_seed, Stats &st) {
    V3 energy = V3{0, 0, 0};
    V3 path = V3{1, 1, 1};
    const int LIGHTS = (int)sc.lights;
    const int MAX_RECURSION = fr.recursion > 1 ? fr.recursion : 1;
    // the `light` result variables of sampleLight / sampleLightGlossy, kept across depths (see the header)
    V3 kept_diff_light = V3{0, 0, 0}, kept_spec_light = V3{0, 0, 0};

    for (int depth = 0; depth < MAX_RECURSION; depth++) {
        V3 sec;
        st.rays++;
        if (orc_debug())
            fprintf(stderr, "ray from %.9g %.9g %.9g dir %.9g %.9g %.9g\n", ray.position.x, ray.position.y, ray.position.z,
                    ray.direction.x, ray.direction.y, ray.direction.z);
        const int current_tri = findIntersection(sc, ray, sec);
        if (current_tri < 0) {
            energy = energy + skyColor(sc, fr, ray.direction) * path;
            break;
        }
        const TriShading &tri = sc.shadings[current_tri];
        const uint32_t mat_id = tri.material_id;
        const Material &material = material_at(sc, mat_id);
        const bool has_texture = mat_id < sc.has_texture.size() && sc.has_texture[mat_id] == 1;

        if (current_tri < LIGHTS) energy = energy + path * v3(material.emission_ior);

        const V3 N = calculateN(tri, sec);
        V3 normal = N, albedo = v3(material.albedo), specular = v3(material.specular_roughness);
        float occlusion = 1.0f, roughness = material.specular_roughness[3];
        float metallic = (roughness < 0.125f) ? 1.0f : 0.0f;
        if (has_texture) {                                         // :471-487
            float ts, tt;
            calculateUV(tri, sec, ts, tt);
            V3 T, Bt;
            calculateTBN(tri, N, sec, T, Bt);
            const V3 texel_albedo = sampleAtlas(sc, ts, tt, mat_id * 3u + 0u);
            const V3 tn_raw = sampleAtlas(sc, ts, tt, mat_id * 3u + 1u);
            const V3 tex_normal = normalize(V3{tn_raw.x * 2.0f - 1.0f, tn_raw.y * 2.0f - 1.0f, tn_raw.z * 2.0f - 1.0f});
            const V3 texel_arm = sampleAtlas(sc, ts, tt, mat_id * 3u + 2u);
            normal = normalize((T * tex_normal.x + Bt * tex_normal.y) + N * tex_normal.z);   // TBNi * tex_normal
            albedo = texel_albedo;
            specular = texel_albedo;
            occlusion = texel_arm.x;
            roughness = texel_arm.y;
            metallic = texel_arm.z;
            if (orc_debug())
                fprintf(stderr, "  textured: N %.9g %.9g %.9g T %.9g %.9g %.9g tex_normal %.9g %.9g %.9g -> normal %.9g %.9g %.9g\n",
                        N.x, N.y, N.z, T.x, T.y, T.z, tex_normal.x, tex_normal.y, tex_normal.z, normal.x, normal.y, normal.z);
        }
        const float ior = material.emission_ior[3];
        const float variance = roughness * roughness;

        const V3 specular_ray = normalize(reflect(ray.direction, normal));                  // :489
        const V3 scattered_specular_ray = normalize(randomSphere(seed) * variance + specular_ray);
        const V3 scattered_diffuse = normalize(randomSphere(seed) + normal);
        const V3 scattered_glossy_ray = normalize(randomSphere(seed) * variance + normal);

        const float cos_theta = gmax(-dot(normal, ray.direction), 0.0f);                    // :494
        const float transmission = (ior == 0.0f) ? 0.0f : getTransmission(cos_theta, 1.00029f, ior);
        // fma() of :497; Mesa: separate multiply and add
        const float fresnel_reflectance = (1.0f - transmission) * m_pow5(1.0f - cos_theta) + transmission;
        if (orc_debug())
            fprintf(stderr, "  fresnel: cos_theta %.9g (1 - cos^2 = %.9g) ior %.9g transmission %.9g reflectance %.9g\n", cos_theta,
                    1.0f - cos_theta * cos_theta, ior, transmission, fresnel_reflectance);

        const float rand_reflectance = unitFloat(seed);
        const float rand_scatter = unitFloat(seed);
        const float rand_metallic = unitFloat(seed);

        const Ray old_ray = ray;
        ray.position = ray.position + ray.direction * sec.z;                                // :507

        if (rand_reflectance < fresnel_reflectance) {                                       // :509-514
            path = path * ((rand_metallic < metallic) ? V3{1.0f - roughness, 1.0f - roughness, 1.0f - roughness}
                                                      : specular);
            ray.direction = specular_ray;
        } else if (rand_metallic < metallic) {                                              // :516-522
            path = path * albedo;
            ray.direction = scattered_specular_ray;
        } else if (rand_scatter < roughness) {                                              // :524-538 Oren-Nayar
            const float theta_i = m_acos(cos_theta);
            const float theta_o = m_acos(dot(scattered_diffuse, normal));
            const float var2 = roughness * roughness;
            const float A = 1.0f - 0.5f * var2 / (var2 + 0.33f);
            const float B = 0.45f * var2 / (var2 + 0.09f);
            const float alpha = gmax(theta_i, theta_o);
            const float beta = gmin(theta_i, theta_o);
            const float w = (A + B * gmax(0.0f, theta_i - theta_o) * m_sin(alpha) * m_tan(beta)) * INV_PI * occlusion;
            path = path * (albedo * w);
            ray.direction = scattered_diffuse;
        } else {                                                                            // :539-545 glossy
            path = path * (specular * (1.0f - roughness));
            ray.direction = scattered_glossy_ray;
        }

        V3 diff_light = V3{0, 0, 0}, spec_light = V3{0, 0, 0};
        bool has_light;
        if (unitFloat(light_seed) < 0.5f) {                                                 // :553-556
            has_light = sampleLight(sc, ray.position, normal, current_tri, seed, kept_diff_light, st);
            diff_light = kept_diff_light;
        } else {
            has_light = sampleLightGlossy(sc, fr, ray.position, normal, old_ray.direction * sec.z, roughness,
                                          current_tri, seed, kept_spec_light, st);
            spec_light = kept_spec_light;
        }
        const float sw = fresnel_reflectance + 1.0f - roughness;
        const V3 global_light = ((albedo * diff_light) * roughness + (specular * spec_light) * sw) * 2.0f;   // :557

        if (has_light) energy = energy + path * global_light;
        if (orc_debug())
            fprintf(stderr, "depth %d tri %d mat %u sec %g %g %g N %g %g %g rough %g fres %g path %g %g %g dir %g %g %g "
                    "has_light %d gl %g %g %g energy %g %g %g\n", depth, current_tri, mat_id, sec.x, sec.y, sec.z,
                    normal.x, normal.y, normal.z, roughness, fresnel_reflectance, path.x, path.y, path.z,
                    ray.direction.x, ray.direction.y, ray.direction.z, (int)has_light, global_light.x, global_light.y,
                    global_light.z, energy.x, energy.y, energy.z);

        const V3 true_normal = v3(sc.models[current_tri].n);                                // :568-571
        if (orc_debug()) fprintf(stderr, "  term dot %.9g\n", dot(ray.direction, true_normal));
        if (dot(ray.direction, true_normal) <= 0.0f) break;

        const float brightness =
            gclamp(gmax(dot(global_light, path), (path.x + path.y + path.z) / 3.0f), 1.0f / 8.0f, 1.0f);   // :575
        if (orc_debug()) { uint32_t s2 = seed; fprintf(stderr, "  roulette brightness %.9g rand %.9g\n", brightness, unitFloat(s2)); }
        if (unitFloat(seed) > brightness) break;
        path = V3{path.x / brightness, path.y / brightness, path.z / brightness};
    }
    return energy;
}

// One pixel-sample, raytracer.glsl:586-607 (everything of main() except the image update)
V3 pixel_sample(const Scene &sc, const Frame &fr, uint32_t tx, uint32_t ty, uint32_t sample, Stats &st) {
    const uint32_t W = fr.width, H = fr.height;
    const float inv_width = 1.0f / (float)W;
    const float inv_height = 1.0f / (float)H;

    uint32_t seed = fr.clock + (tx * H + ty + 394587u) * (sample + 1u) * 13u + sample;       // :594
    uint32_t light_seed = fr.clock;
    for (uint32_t i = 0; i < sample; i++) light_seed = pcgHash(light_seed);                 // :595-598

    const float j1 = unitFloat(seed) * inv_width;
    const float j2 = unitFloat(seed) * inv_height;
    const V3 d0 = V3{((float)(int)tx - 0.5f * (float)W + 0.5f) * inv_height,
                     ((float)(int)ty + 0.5f) * inv_height - 0.5f, 0.5f};
    const V3 dtctor = normalize(d0 + V3{j1, j2, 0.0f});                                     // :602
    const V3 dir3 = normalize(dtctor - V3{0, 0, 0});                                        // :603
    const float *M = fr.camera;
    // CAMERA * vec4(0,0,0,1) and CAMERA * vec4(dir, 0), column by column             :605-606
    auto mul = [&](float x, float y, float z, float w, float out[4]) {
        for (int r = 0; r < 4; r++) out[r] = ((M[0 + r] * x + M[4 + r] * y) + M[8 + r] * z) + M[12 + r] * w;
    };
    float o4[4], d4[4];
    mul(0.0f, 0.0f, 0.0f, 1.0f, o4);
    mul(dir3.x, dir3.y, dir3.z, 0.0f, d4);
    // dot of the vec4 with itself, summed in the same order as dot() above
#if ORC_DOT_ORDER == 0
    const float len2 = ((d4[0] * d4[0] + d4[1] * d4[1]) + d4[2] * d4[2]) + d4[3] * d4[3];
#else
    const float len2 = ((d4[3] * d4[3] + d4[2] * d4[2]) + d4[1] * d4[1]) + d4[0] * d4[0];
#endif
    const float inv = 1.0f / sqrtf(len2);
    Ray ray;
    ray.position = V3{o4[0], o4[1], o4[2]};
    ray.direction = V3{d4[0] * inv, d4[1] * inv, d4[2] * inv};
    st.pixel_samples++;
    return trace(sc, fr, ray, seed, light_seed, st);
}

}  // namespace

// ---------------------------------------------------------------- C interface (ctypes)

// llvmpipe runs its compiled shaders with denormal inputs read as zero and denormal results flushed to zero (the
// x86 control bits DAZ and FTZ).  orc_set_flush_denormals(1) makes the batch entry points below do the same, per
// thread and only for the duration of the call.
static int g_flush_denormals = 0;
struct FlushScope {
    unsigned saved;
    FlushScope() : saved(__builtin_ia32_stmxcsr()) { if (g_flush_denormals) __builtin_ia32_ldmxcsr(saved | 0x8040u); }
    ~FlushScope() { __builtin_ia32_ldmxcsr(saved); }
};

extern "C" {

void orc_set_flush_denormals(int on) { g_flush_denormals = on; }
uint32_t orc_pcg_hash(uint32_t k) { return pcgHash(k); }
float orc_unit_float(uint32_t *seed) { return unitFloat(*seed); }
int orc_libm(void) { return ORC_LIBM; }
int orc_dot_order(void) { return ORC_DOT_ORDER; }
void orc_random_sphere(uint32_t *seed, float out[3]) {
    const V3 v = randomSphere(*seed);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
uint32_t orc_pack_r11g11b10(float r, float g, float b) { return pack_r11g11b10(r, g, b); }
void orc_unpack_r11g11b10(uint32_t p, float out[3]) {
    const V3 v = unpack_r11g11b10(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
// Elementary functions of the active back end, for accuracy tests: which = 0 sin, 1 cos, 2 tan,
// 3 acos, 4 asin, 5 atan2(x, y), 6 pow5
float orc_math(int which, float x, float y) {
    float s, c;
    switch (which) {
        case 0: m_sincos(x, s, c); return s;
        case 1: m_sincos(x, s, c); return c;
        case 2: return m_tan(x);
        case 3: return m_acos(x);
        case 4: return m_asin(x);
        case 5: return m_atan2(x, y);
        default: return m_pow5(x);
    }
}

void orc_math_batch(int which, const float *x, const float *y, uint64_t n, float *out) {
    for (uint64_t i = 0; i < n; i++) out[i] = orc_math(which, x[i], y ? y[i] : 0.0f);
}

// Bilinear lookups in the environment map, (s, t) pairs in, rgb out.
void orc_sample_env(const void *h, const float *st, uint64_t n, float *rgb) {
    const Scene *sc = (const Scene *)h;
    for (uint64_t i = 0; i < n; i++) {
        const V3 c = sampleEnv(*sc, st[2 * i], st[2 * i + 1]);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

// Irradiance pre-filter of the environment map: Scene::loadEnvironmentTexture (Scene.cpp:150-186) runs
// irradiance.glsl once per radiance row; every texel of the (w/4) x (h/4) irradiance image accumulates, in row
// order and left to right within a row, weight * radiance with weight = max(uvDotUV(uv, UV), 0) * 2 / (w * h)
// (irradiance.glsl:16-18, :27-38).  uv = (i, row) / size and UV = 4 * texel / size are corner coordinates, as in
// the reference.  Mesa: products left to right; no fused multiply-add.
void orc_irradiance(const float *env_rgba, uint32_t w, uint32_t h, float *out_rgb) {
    const float PI = 3.141592653589793f, TWO_PI = 2.0f * PI;
    const uint32_t iw = w / 4, ih = h / 4;
    const float isx = 1.0f / (float)w, isy = 1.0f / (float)h;
    const float renorm = (2.0f / (float)w) / (float)h;     // Mesa: these two divisions stay divisions
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < (int64_t)iw * ih; t++) {
        const uint32_t X = (uint32_t)(t % iw), Y = (uint32_t)(t / iw);
        const float Ux = ((float)X * isx) * 4.0f, Uy = ((float)Y * isy) * 4.0f;
        float sU, cU;
        m_sincos(PI * Uy, sU, cU);
        float r = 0.0f, g = 0.0f, b = 0.0f;
        for (uint32_t row = 0; row < h; row++) {
            const float uy = (float)row * isy;
            float sy, cy;
            m_sincos(PI * uy, sy, cy);
            const float polar = cy * cU;
            for (uint32_t i = 0; i < w; i++) {
                const float ux = (float)i * isx;
                float s_, c;
                m_sincos(TWO_PI * (ux - Ux), s_, c);
                const float weight = gmax(polar + (c * sy) * sU, 0.0f) * renorm;
                const float *px = env_rgba + 4 * ((size_t)row * w + i);
                r = r + weight * px[0];
                g = g + weight * px[1];
                b = b + weight * px[2];
            }
        }
        out_rgb[3 * t] = r; out_rgb[3 * t + 1] = g; out_rgb[3 * t + 2] = b;
    }
}

void *orc_scene_create(const void *models, const void *shadings, uint32_t count, uint32_t lights,
                       const void *materials, uint32_t n_materials) {
    Scene *sc = new Scene();
    sc->count = count;
    sc->lights = lights;
    sc->models.resize(count);
    sc->shadings.resize(count);
    sc->materials.resize(n_materials);
    if (count) {
        memcpy(sc->models.data(), models, sizeof(TriModel) * count);
        memcpy(sc->shadings.data(), shadings, sizeof(TriShading) * count);
    }
    if (n_materials) memcpy(sc->materials.data(), materials, sizeof(Material) * n_materials);
    sc->has_texture.assign(n_materials, 0);
    sc->atlas.resize((size_t)n_materials * 3);
    return sc;
}
void orc_scene_destroy(void *h) { delete (Scene *)h; }
void orc_scene_set_atlas_size(void *h, uint32_t size) {
    Scene *sc = (Scene *)h;
    sc->atlas_size = size;
    for (auto &l : sc->atlas) l.clear();
}
// Scene::loadTexture, Scene.cpp:75-102: texels [0,w) x [0,h) of the layer, the rest stays zero.
int orc_scene_upload_atlas_layer(void *h, uint32_t material, uint32_t slot, uint32_t w, uint32_t hgt,
                                 const float *rgba) {
    Scene *sc = (Scene *)h;
    const size_t layer = (size_t)material * 3 + slot;
    if (layer >= sc->atlas.size() || slot > 2 || w > sc->atlas_size || hgt > sc->atlas_size) return -1;
    std::vector<uint32_t> &tx = sc->atlas[layer];
    tx.assign((size_t)sc->atlas_size * sc->atlas_size, 0u);
    for (uint32_t j = 0; j < hgt; j++)
        for (uint32_t i = 0; i < w; i++) {
            const float *p = &rgba[((size_t)j * w + i) * 4];
            tx[(size_t)j * sc->atlas_size + i] = pack_r11g11b10(p[0], p[1], p[2]);
        }
    return 0;
}
void orc_scene_set_has_texture(void *h, const int32_t *flags, uint32_t n) {
    Scene *sc = (Scene *)h;
    for (uint32_t i = 0; i < n && i < sc->has_texture.size(); i++) sc->has_texture[i] = flags[i];
}
void orc_scene_upload_env(void *h, uint32_t w, uint32_t hgt, const float *rgba) {
    Scene *sc = (Scene *)h;
    sc->env.w = w;
    sc->env.h = hgt;
    sc->env.rgba.assign(rgba, rgba + (size_t)w * hgt * 4);
}

// findIntersection over a batch of rays.
void orc_closest_hit_batch(const void *h, const float *o, const float *d, uint64_t n, int32_t *id, float *uvt) {
    const Scene *sc = (const Scene *)h;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        FlushScope fpmode;
        Ray r;
        r.position = v3(&o[i * 3]);
        r.direction = v3(&d[i * 3]);
        V3 sec;
        id[i] = findIntersection(*sc, r, sec);
        uvt[i * 3 + 0] = sec.x;
        uvt[i * 3 + 1] = sec.y;
        uvt[i * 3 + 2] = sec.z;
    }
}

// Boolean any-hit of intersectTriangleShadow over all triangles != current (the loop of sampleLight).
void orc_shadow_batch(const void *h, const float *o, const float *d, const float *max_t, const int32_t *current,
                      uint64_t n, int32_t *occluded) {
    const Scene *sc = (const Scene *)h;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        FlushScope fpmode;
        Ray r;
        r.position = v3(&o[i * 3]);
        r.direction = v3(&d[i * 3]);
        int occ = 0;
        for (int t = 0; t < (int)sc->count; t++) {
            if (t == current[i]) continue;
            if (intersectTriangleShadow(r, sc->models[t], max_t[i])) { occ = 1; break; }
        }
        occluded[i] = occ;
    }
}

// Radiance of single pixel-samples (for spot checks): xs, ys, samples have n entries, rgb gets n*3.
void orc_pixel_samples(const void *h, const float camera[16], uint32_t clock, int32_t recursion, float exposure,
                       uint32_t width, uint32_t height, const uint32_t *xs, const uint32_t *ys,
                       const uint32_t *samples, uint64_t n, float *rgb) {
    const Scene *sc = (const Scene *)h;
    Frame fr;
    memcpy(fr.camera, camera, sizeof(fr.camera));
    fr.clock = clock; fr.recursion = recursion; fr.exposure = exposure; fr.width = width; fr.height = height;
    Stats st = {};
    FlushScope fpmode;
    for (uint64_t i = 0; i < n; i++) {
        const V3 c = pixel_sample(*sc, fr, xs[i], ys[i], samples[i], st);
        rgb[i * 3 + 0] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    }
}

// Renders SAMPLE = sample_begin + k*sample_stride, k in [0, sample_count), for every pixel.
//   mode 0: rgb (width*height*3) receives per-pixel SUMS added in ascending sample order to the
//           values already in rgb (so calls chain);
//   mode 1: the reference's running mean (raytracer.glsl:609-616) with SAMPLE as the weight, which
//           assumes sample_begin == number of samples already in rgb and sample_stride == 1.
// stats (6 uint64: pixel_samples, rays, shadow_probes, glossy_probes, closest_tests, probe_tests) may be NULL.
void orc_render(const void *h, const float camera[16], uint32_t clock, int32_t recursion, float exposure,
                uint32_t width, uint32_t height, uint32_t sample_begin, uint32_t sample_count,
                uint32_t sample_stride, int mode, float *rgb, uint64_t *stats) {
    const Scene *sc = (const Scene *)h;
    Frame fr;
    memcpy(fr.camera, camera, sizeof(fr.camera));
    fr.clock = clock; fr.recursion = recursion; fr.exposure = exposure; fr.width = width; fr.height = height;
    uint64_t tot[4] = {0, 0, 0, 0};
#pragma omp parallel
    {
        Stats st = {};
        FlushScope fpmode;
#pragma omp for schedule(dynamic, 256) collapse(2)
        for (int64_t y = 0; y < (int64_t)height; y++)
            for (int64_t x = 0; x < (int64_t)width; x++) {
                float *px = &rgb[((size_t)y * width + (size_t)x) * 3];
                for (uint32_t k = 0; k < sample_count; k++) {
                    const uint32_t S = sample_begin + k * sample_stride;
                    const V3 c = pixel_sample(*sc, fr, (uint32_t)x, (uint32_t)y, S, st);
                    if (mode == 0) {
                        px[0] += c.x; px[1] += c.y; px[2] += c.z;
                    } else if (S > 0) {
                        const float inv_samples = 1.0f / (float)(S + 1u);
                        const float scale = (float)S * inv_samples;
                        px[0] = px[0] * scale + inv_samples * c.x;
                        px[1] = px[1] * scale + inv_samples * c.y;
                        px[2] = px[2] * scale + inv_samples * c.z;
                    } else {
                        px[0] = c.x; px[1] = c.y; px[2] = c.z;
                    }
                }
            }
#pragma omp critical
        {
            tot[0] += st.pixel_samples; tot[1] += st.rays; tot[2] += st.shadow_probes; tot[3] += st.glossy_probes;
        }
    }
    if (stats) {
        stats[0] = tot[0]; stats[1] = tot[1]; stats[2] = tot[2]; stats[3] = tot[3];
        stats[4] = tot[1] * (uint64_t)sc->count;
        stats[5] = (tot[2] + tot[3]) * (uint64_t)(sc->count ? sc->count - 1 : 0);
    }
}

// displayQuad.fs:19-25 sRGB transfer of one channel
float orc_linear_to_srgb(float c) {
    return (c <= 0.0031308f) ? c * 12.92f : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}

}  // extern "C"
