// rt_device.h - device-side building blocks of the gfx950 path tracer.
//
// Arithmetic contract: every value that can influence a hit decision or a random
// branch is computed in IEEE float32 with separate multiplies and adds (the translation
// unit is compiled with -ffp-contract=off), dot products are summed in the order fixed by
// RT_DOT_ORDER, divisions and square roots are correctly rounded, and the elementary
// functions come from rt_detmath.h.  Under that contract the device reproduces a host
// evaluation of the same formulas bit for bit.
//
// Each function names the lines of the reference shader (res/shader/raytracer.glsl) whose
// behaviour it implements.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_detmath.h"

#ifndef RT_DOT_ORDER
#define RT_DOT_ORDER 0   // 0: (x*x' + y*y') + z*z'    1: (z*z' + y*y') + x*x'
#endif

#define RT_DEV __device__ __forceinline__

namespace rt {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }

RT_DEV float dot(f3 a, f3 b) {
#if RT_DOT_ORDER == 0
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
#else
    return (a.z * b.z + a.y * b.y) + a.x * b.x;
#endif
}
RT_DEV f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
RT_DEV float length(f3 a) { return __builtin_sqrtf(dot(a, a)); }
RT_DEV f3 normalize(f3 a) {
    const float inv = 1.0f / __builtin_sqrtf(dot(a, a));
    return a * inv;
}
RT_DEV f3 reflect(f3 I, f3 N) {
    const float k = 2.0f * dot(N, I);
    return I - N * k;
}
RT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }
RT_DEV float gmin(float x, float y) { return (y < x) ? y : x; }
RT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
RT_DEV float pow5(float x) { const float x2 = x * x; return (x2 * x2) * x; }

constexpr float TWO_PI_F = 6.283185307179586f;
constexpr float INV_PI_F = (float)(1.0 / 3.141592653589793);
constexpr float INV_UINT_MAX_F = 1.0f / 4294967296.0f;

// ---------------------------------------------------------------- scene as the kernels see it

// Geometry: 3 x float4 per triangle = the 12 floats of the reference's TriangleModel record
// (true_normal, position, u, v).  The index is wave-uniform in the brute-force loops, so
// these are fetched with scalar loads.
// Shading: 7 planes of float4[count], gathered per lane:
//   plane 0..2: vertex normal k (xyz), w = {material_id bits, tex_p.s, tex_p.t}
//   plane 3   : tex_u.s, tex_u.t, tex_v.s, tex_v.t
//   plane 4..6: vertex tangent k (xyz)   (read only for textured materials)
struct DScene {
    const float4 *geo;
    const float4 *shade;
    const float4 *mats;          // 3 x float4 per material
    const int *has_tex;
    const uint32_t *const *atlas; // one pointer per layer, null = layer never uploaded (all zero)
    const float4 *env;
    int count, lights, n_mat;
    int atlas_size, atlas_layers;
    int env_w, env_h;
};

struct DFrame {
    float cam[16];
    uint32_t clock;
    int recursion;
    float exposure;
    uint32_t width, height;
};

struct Ray { f3 o, d; };

// ---------------------------------------------------------------- RNG  (raytracer.glsl:83-98)

RT_DEV uint32_t pcg_hash(uint32_t k) {
    const uint32_t state = k * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
RT_DEV float unit_float(uint32_t &seed) {
    seed = pcg_hash(seed);
    return (float)seed * INV_UINT_MAX_F;
}
RT_DEV f3 random_sphere(uint32_t &seed) {          // :100-113
    const float theta = TWO_PI_F * unit_float(seed);
    const float x = unit_float(seed) * 2.0f - 1.0f;
    const float sinx = __builtin_sqrtf(1.0f - x * x);
    float s, c;
    dm::sincos(theta, s, c);
    return mk3(sinx * c, sinx * s, x);
}

// ---------------------------------------------------------------- triangle records

struct Tri { f3 n, p, u, v; };

// Triangle records never change while a kernel runs.  Reading them through the constant address
// space tells the compiler so; with a wave-uniform index the loads then become scalar loads
// (s_load_dwordx4/x8 into SGPRs) even in kernels that also store to global memory.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f *const_v4f_ptr;

RT_DEV Tri load_tri(const float4 *geo, int i) {
    const const_v4f_ptr g = (const_v4f_ptr)geo;
    const v4f a = g[3 * i + 0], b = g[3 * i + 1], c = g[3 * i + 2];
    Tri t;
    t.n = mk3(a.x, a.y, a.z);
    t.p = mk3(a.w, b.x, b.y);
    t.u = mk3(b.z, b.w, c.x);
    t.v = mk3(c.y, c.z, c.w);
    return t;
}
// Out-of-range indices read as an all-zero record (the shader's robust-buffer behaviour).
RT_DEV Tri load_tri_checked(const DScene &sc, int i) {
    if (i >= 0 && i < sc.count) return load_tri(sc.geo, i);
    Tri t;
    t.n = t.p = t.u = t.v = mk3(0.0f, 0.0f, 0.0f);
    return t;
}

// Closest-hit test, raytracer.glsl:125-162, in two steps so that a wave can skip the second one
// when no lane needs it.  `isec` = (x, y, relative_depth, determinant), all still scaled by the
// determinant.
//
// Step 1: plane tests.  Returns true when the triangle is front-facing, in front of the origin and
// its plane is not farther along the ray than the hit in isec.
struct PlaneTest { float det, rd; f3 delta; };

RT_DEV bool plane_test_closest(const Ray &ray, const Tri &t, const f4 &isec, PlaneTest &pt) {
    pt.det = -dot(ray.d, t.n);
    pt.delta = ray.o - t.p;
    pt.rd = dot(t.n, pt.delta);
    return !((pt.det <= 0.0f) | (pt.rd <= 0.0f) | ((isec.z >= 0.0f) & (pt.rd * isec.w > isec.z * pt.det)));
}
// Step 2: inside test in (u, v, ray) coordinates.  Returns true when the point lies in the triangle.
RT_DEV bool inside_test(const Ray &ray, const Tri &t, const PlaneTest &pt, float &rx, float &ry) {
    const f3 minor = cross(ray.d, pt.delta);
    rx = -dot(minor, t.v);
    ry = dot(minor, t.u);
    return !((rx < 0.0f) | (ry < 0.0f) | (rx + ry > pt.det));
}

RT_DEV bool intersect_triangle(const Ray &ray, const Tri &t, f4 &isec) {
    PlaneTest pt;
    const bool p = plane_test_closest(ray, t, isec, pt);
    float rx, ry;
    const bool hit = p & inside_test(ray, t, pt, rx, ry);
    if (hit) { isec.x = rx; isec.y = ry; isec.z = pt.rd; isec.w = pt.det; }
    return hit;
}

// Occlusion test, raytracer.glsl:164-196: only triangles facing away from the origin occlude.
RT_DEV bool plane_test_shadow(const Ray &ray, const Tri &t, float max_t, PlaneTest &pt) {
    pt.det = dot(ray.d, t.n);
    pt.delta = t.p - ray.o;
    pt.rd = dot(t.n, pt.delta);
    return !((pt.det <= 0.0f) | (pt.rd <= 0.0f) | (pt.rd > pt.det * max_t));
}

RT_DEV bool intersect_triangle_shadow(const Ray &ray, const Tri &t, float max_t) {
    PlaneTest pt;
    const bool p = plane_test_shadow(ray, t, max_t, pt);
    float rx, ry;
    return p & inside_test(ray, t, pt, rx, ry);
}

// ---------------------------------------------------------------- shading records

struct ShadeRec {
    uint32_t mat;
    f3 n0, n1, n2;
    float tex_p[2], tex_u[2], tex_v[2];
};

RT_DEV ShadeRec load_shade(const DScene &sc, int i) {
    ShadeRec r;
    if (i >= 0 && i < sc.count) {
        const float4 a = sc.shade[(size_t)0 * sc.count + i];
        const float4 b = sc.shade[(size_t)1 * sc.count + i];
        const float4 c = sc.shade[(size_t)2 * sc.count + i];
        const float4 d = sc.shade[(size_t)3 * sc.count + i];
        r.mat = __float_as_uint(a.w);
        r.n0 = mk3(a.x, a.y, a.z); r.n1 = mk3(b.x, b.y, b.z); r.n2 = mk3(c.x, c.y, c.z);
        r.tex_p[0] = b.w; r.tex_p[1] = c.w;
        r.tex_u[0] = d.x; r.tex_u[1] = d.y; r.tex_v[0] = d.z; r.tex_v[1] = d.w;
    } else {
        r.mat = 0;
        r.n0 = r.n1 = r.n2 = mk3(0.0f, 0.0f, 0.0f);
        r.tex_p[0] = r.tex_p[1] = r.tex_u[0] = r.tex_u[1] = r.tex_v[0] = r.tex_v[1] = 0.0f;
    }
    return r;
}

struct Mat { f3 albedo, specular, emission; float roughness, ior; };

RT_DEV Mat load_mat(const DScene &sc, uint32_t m) {
    Mat r;
    if (m < (uint32_t)sc.n_mat) {
        const float4 a = sc.mats[3 * m + 0], b = sc.mats[3 * m + 1], c = sc.mats[3 * m + 2];
        r.albedo = mk3(a.x, a.y, a.z);
        r.specular = mk3(b.x, b.y, b.z);
        r.roughness = b.w;
        r.emission = mk3(c.x, c.y, c.z);
        r.ior = c.w;
    } else {
        r.albedo = r.specular = r.emission = mk3(0.0f, 0.0f, 0.0f);
        r.roughness = r.ior = 0.0f;
    }
    return r;
}

// calculateN, raytracer.glsl:239-248
RT_DEV f3 calc_normal(const ShadeRec &s, float u, float v) {
    const float w0 = 1.0f - u - v;
    return normalize((s.n0 * w0 + s.n1 * u) + s.n2 * v);
}

// ---------------------------------------------------------------- textures

// Unsigned 11/10-bit floats of R11F_G11F_B10F: 5 exponent bits (bias 15), `mbits` mantissa bits.
RT_DEV float uf_decode(uint32_t code, int mbits) {
    const uint32_t e = code >> mbits, m = code & ((1u << mbits) - 1u);
    if (e == 0) return (float)m * (1.0f / (float)(1u << (14 + mbits)));
    if (e == 31) return __uint_as_float(0x7F800000u | (m << (23 - mbits)));
    return __uint_as_float(((e + 112u) << 23) | (m << (23 - mbits)));
}
RT_DEV f3 unpack_r11g11b10(uint32_t p) {
    return mk3(uf_decode(p & 0x7FFu, 6), uf_decode((p >> 11) & 0x7FFu, 6), uf_decode(p >> 22, 5));
}

struct Bilinear { int i0, i1, j0, j1; float a, b; };

// LINEAR filtering with REPEAT wrap: x = s*W - 0.5, i0 = floor(x), a = x - i0.
RT_DEV Bilinear bilinear_setup(float s, float t, int W, int H) {
    float x = s * (float)W - 0.5f, y = t * (float)H - 0.5f;
    if (!(__builtin_fabsf(x) < 1073741824.0f)) x = 0.0f;
    if (!(__builtin_fabsf(y) < 1073741824.0f)) y = 0.0f;
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    Bilinear bl;
    bl.a = x - fx;
    bl.b = y - fy;
    const int ix = (int)fx, iy = (int)fy;
    bl.i0 = ((ix % W) + W) % W;
    bl.i1 = (((ix + 1) % W) + W) % W;
    bl.j0 = ((iy % H) + H) % H;
    bl.j1 = (((iy + 1) % H) + H) % H;
    return bl;
}
RT_DEV f3 bilinear_mix(f3 t00, f3 t10, f3 t01, f3 t11, float a, float b) {
    const float ia = 1.0f - a, ib = 1.0f - b;
    const f3 top = t00 * ia + t10 * a;
    const f3 bot = t01 * ia + t11 * a;
    return top * ib + bot * b;
}

RT_DEV f3 sample_env(const DScene &sc, float s, float t) {
    if (sc.env_w == 0 || sc.env_h == 0) return mk3(0.0f, 0.0f, 0.0f);
    const Bilinear bl = bilinear_setup(s, t, sc.env_w, sc.env_h);
    const float4 a = sc.env[(size_t)bl.j0 * sc.env_w + bl.i0], b = sc.env[(size_t)bl.j0 * sc.env_w + bl.i1];
    const float4 c = sc.env[(size_t)bl.j1 * sc.env_w + bl.i0], d = sc.env[(size_t)bl.j1 * sc.env_w + bl.i1];
    return bilinear_mix(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), bl.a, bl.b);
}

RT_DEV f3 sample_atlas(const DScene &sc, float s, float t, uint32_t layer) {
    if (layer >= (uint32_t)sc.atlas_layers) return mk3(0.0f, 0.0f, 0.0f);
    const uint32_t *tx = sc.atlas[layer];
    if (tx == nullptr) return mk3(0.0f, 0.0f, 0.0f);
    const int S = sc.atlas_size;
    const Bilinear bl = bilinear_setup(s, t, S, S);
    const f3 a = unpack_r11g11b10(tx[(size_t)bl.j0 * S + bl.i0]), b = unpack_r11g11b10(tx[(size_t)bl.j0 * S + bl.i1]);
    const f3 c = unpack_r11g11b10(tx[(size_t)bl.j1 * S + bl.i0]), d = unpack_r11g11b10(tx[(size_t)bl.j1 * S + bl.i1]);
    return bilinear_mix(a, b, c, d, bl.a, bl.b);
}

// skyColor, raytracer.glsl:273-276
RT_DEV f3 sky_color(const DScene &sc, const DFrame &fr, f3 dir) {
    const float s = dm::atan2(dir.z, dir.x) * INV_PI_F * 0.5f + 0.5f;
    const float t = -dm::asin(dir.y) * INV_PI_F + 0.5f;
    return sample_env(sc, s, t) * fr.exposure;
}

// getTransmission, raytracer.glsl:198-215
RT_DEV float get_transmission(float cosThetaI, float etaI, float etaT) {
    const float sinThetaI = __builtin_sqrtf(1.0f - cosThetaI * cosThetaI);
    const float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1.0f) return 1.0f;
    const float cosThetaT = __builtin_sqrtf(1.0f - sinThetaT * sinThetaT);
    const float r0 = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    const float r1 = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    return (r0 * r0 + r1 * r1) * 0.5f;
}

// ---------------------------------------------------------------- camera, raytracer.glsl:586-607

RT_DEV uint32_t pixel_seed(const DFrame &fr, uint32_t tx, uint32_t ty, uint32_t sample) {
    return fr.clock + (tx * fr.height + ty + 394587u) * (sample + 1u) * 13u + sample;    // :594
}

RT_DEV Ray camera_ray(const DFrame &fr, uint32_t tx, uint32_t ty, uint32_t &seed) {
    const float W = (float)fr.width, H = (float)fr.height;
    const float inv_width = 1.0f / W, inv_height = 1.0f / H;
    const float j1 = unit_float(seed) * inv_width;
    const float j2 = unit_float(seed) * inv_height;
    const f3 d0 = mk3(((float)(int)tx - 0.5f * W + 0.5f) * inv_height, ((float)(int)ty + 0.5f) * inv_height - 0.5f, 0.5f);
    const f3 dtctor = normalize(d0 + mk3(j1, j2, 0.0f));
    const f3 dir3 = normalize(dtctor - mk3(0.0f, 0.0f, 0.0f));
    const float *M = fr.cam;
    float o4[4], d4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        o4[r] = ((M[0 + r] * 0.0f + M[4 + r] * 0.0f) + M[8 + r] * 0.0f) + M[12 + r] * 1.0f;
        d4[r] = ((M[0 + r] * dir3.x + M[4 + r] * dir3.y) + M[8 + r] * dir3.z) + M[12 + r] * 0.0f;
    }
#if RT_DOT_ORDER == 0
    const float len2 = ((d4[0] * d4[0] + d4[1] * d4[1]) + d4[2] * d4[2]) + d4[3] * d4[3];
#else
    const float len2 = ((d4[3] * d4[3] + d4[2] * d4[2]) + d4[1] * d4[1]) + d4[0] * d4[0];
#endif
    const float inv = 1.0f / __builtin_sqrtf(len2);
    Ray ray;
    ray.o = mk3(o4[0], o4[1], o4[2]);
    ray.d = mk3(d4[0] * inv, d4[1] * inv, d4[2] * inv);
    return ray;
}

// ---------------------------------------------------------------- one bounce, raytracer.glsl:447-581
//
// The bounce is cut in three at the light probe, so that the probe loops can run as their
// own (densely packed) kernel stage:
//   shade_pre   :458-556 up to the probe loop: material, new direction, throughput, all
//               random draws of the bounce except the roulette draw, the probe ray
//   probe       the occlusion loop of sampleLight (:372-378) or the search loop of
//               sampleLightGlossy (:406-419)
//   shade_post  :557-579: add the probed light, back-side stop, Russian roulette

enum ProbeKind : int { PROBE_NONE = 0, PROBE_SHADOW = 1, PROBE_GLOSSY = 2 };

struct PathState {
    f3 o, d;          // current ray
    f3 path;          // throughput
    f3 energy;        // radiance gathered so far
    uint32_t seed;
};

struct Bounce {
    int kind;         // ProbeKind
    int current;      // triangle the path sits on
    f3 probe_d;       // probe ray direction (origin = PathState::o after shade_pre)
    float max_t;      // shadow probes: distance to the sampled light point
    // shadow probes: c = the light term global_light would have if unoccluded, w unused
    // glossy probes: c = specular colour, w = fresnel_reflectance + 1 - roughness
    f3 c;
    float w;
};

// sampleLight up to its occlusion loop, raytracer.glsl:326-370 and :379-386.  Returns true when
// the loop has to run; `lit` then holds the light the call returns if nothing occludes.
RT_DEV bool sample_light_setup(const DScene &sc, f3 position, f3 normal, uint32_t &seed, f3 &probe_d, float &max_t,
                               f3 &lit) {
    if (sc.lights == 0) return false;
    const int light_id = (int)(unit_float(seed) * (float)sc.lights);
    const Tri lt = load_tri_checked(sc, light_id - 1);          // sic, :338
    float cx = unit_float(seed);
    float cy = unit_float(seed);
    if (cx + cy > 1.0f) { cx = 1.0f - cx; cy = 1.0f - cy; }
    const f3 rand_tri = (lt.u * cx + lt.v * cy) + lt.p * 1.0f;
    const f3 delta = rand_tri - position;
    max_t = length(delta);
    probe_d = mk3(delta.x / max_t, delta.y / max_t, delta.z / max_t);
    const float DdN = dot(probe_d, normal);
    const float DdL = dot(probe_d, lt.n);
    if (DdN <= 0.0f || DdL >= 0.0f) return false;
    const ShadeRec ls = load_shade(sc, light_id);               // :379
    const Mat lm = load_mat(sc, ls.mat);
    const float tri_area = 0.5f * length(cross(lt.u, lt.v));
    const float distance_ratio = __builtin_fmaf(max_t, max_t + 2.0f, 1.0f);
    const float k = tri_area * (-DdL) * DdN / distance_ratio * (float)sc.lights;
    lit = lm.emission * k;
    return true;
}

// Everything of a bounce between the closest hit and the probe loop.  `hit` >= 0.
RT_DEV void shade_pre(const DScene &sc, const DFrame &fr, PathState &ps, int hit, float hu, float hv, float ht,
                      bool coin_diffuse, Bounce &bn) {
    const ShadeRec tri = load_shade(sc, hit);
    const uint32_t mat_id = tri.mat;
    const Mat material = load_mat(sc, mat_id);
    const bool has_texture = (mat_id < (uint32_t)sc.n_mat) && (sc.has_tex[mat_id] == 1);

    if (hit < sc.lights) ps.energy = ps.energy + ps.path * material.emission;               // :463-466

    const f3 N = calc_normal(tri, hu, hv);
    f3 normal = N, albedo = material.albedo, specular = material.specular;
    float occlusion = 1.0f, roughness = material.roughness;
    float metallic = (roughness < 0.125f) ? 1.0f : 0.0f;
    if (has_texture) {                                                                     // :471-487
        const float ts = (tri.tex_p[0] * 1.0f + tri.tex_u[0] * hu) + tri.tex_v[0] * hv;
        const float tt = (tri.tex_p[1] * 1.0f + tri.tex_u[1] * hu) + tri.tex_v[1] * hv;
        const float4 t0 = sc.shade[(size_t)4 * sc.count + hit], t1 = sc.shade[(size_t)5 * sc.count + hit],
                     t2 = sc.shade[(size_t)6 * sc.count + hit];
        const float w0 = 1.0f - hu - hv;
        f3 T = (mk3(t0.x, t0.y, t0.z) * w0 + mk3(t1.x, t1.y, t1.z) * hu) + mk3(t2.x, t2.y, t2.z) * hv;
        T = normalize(T - N * dot(N, T));
        const f3 Bt = cross(T, N);
        const f3 texel_albedo = sample_atlas(sc, ts, tt, mat_id * 3u + 0u);
        const f3 tn = sample_atlas(sc, ts, tt, mat_id * 3u + 1u);
        const f3 tex_normal = normalize(mk3(tn.x * 2.0f - 1.0f, tn.y * 2.0f - 1.0f, tn.z * 2.0f - 1.0f));
        const f3 texel_arm = sample_atlas(sc, ts, tt, mat_id * 3u + 2u);
        normal = normalize((T * tex_normal.x + Bt * tex_normal.y) + N * tex_normal.z);
        albedo = texel_albedo;
        specular = texel_albedo;
        occlusion = texel_arm.x;
        roughness = texel_arm.y;
        metallic = texel_arm.z;
    }
    const float ior = material.ior;
    const float variance = roughness * roughness;

    // The shader builds all three scattered directions (:489-492) and then uses one (:509-547).  Their
    // six random numbers are drawn here in the shader's order; sin/cos/sqrt/normalize run only for
    // the direction that is used.
    const float th1 = TWO_PI_F * unit_float(ps.seed), x1 = unit_float(ps.seed) * 2.0f - 1.0f;   // specular scatter
    const float th2 = TWO_PI_F * unit_float(ps.seed), x2 = unit_float(ps.seed) * 2.0f - 1.0f;   // diffuse
    const float th3 = TWO_PI_F * unit_float(ps.seed), x3 = unit_float(ps.seed) * 2.0f - 1.0f;   // glossy

    const float cos_theta = gmax(-dot(normal, ps.d), 0.0f);
    const float transmission = (ior == 0.0f) ? 0.0f : get_transmission(cos_theta, 1.00029f, ior);
    const float fresnel = __builtin_fmaf(1.0f - transmission, pow5(1.0f - cos_theta), transmission);

    const float rand_reflectance = unit_float(ps.seed);
    const float rand_scatter = unit_float(ps.seed);
    const float rand_metallic = unit_float(ps.seed);

    const f3 old_o = ps.o;
    ps.o = ps.o + ps.d * ht;                                                               // :507

    const bool is_mirror = rand_reflectance < fresnel;                                     // :509
    const bool is_metal = !is_mirror && (rand_metallic < metallic);                        // :516
    const bool is_diffuse = !is_mirror && !is_metal && (rand_scatter < roughness);         // :524
    f3 specular_ray = mk3(0.0f, 0.0f, 0.0f);
    if (is_mirror || is_metal) specular_ray = normalize(reflect(ps.d, normal));            // :489
    f3 new_d = specular_ray;
    if (!is_mirror) {
        const float th = is_metal ? th1 : (is_diffuse ? th2 : th3);
        const float x = is_metal ? x1 : (is_diffuse ? x2 : x3);
        const float scale = is_diffuse ? 1.0f : variance;          // sphere * 1.0f + normal == sphere + normal
        const f3 base = is_metal ? specular_ray : normal;
        const float sinx = __builtin_sqrtf(1.0f - x * x);
        float sn, cs;
        dm::sincos(th, sn, cs);
        new_d = normalize(mk3(sinx * cs, sinx * sn, x) * scale + base);                    // :490-492
    }
    if (is_mirror) {                                                                       // :509-514
        const float k = 1.0f - roughness;
        ps.path = ps.path * ((rand_metallic < metallic) ? mk3(k, k, k) : specular);
    } else if (is_metal) {                                                                 // :516-522
        ps.path = ps.path * albedo;
    } else if (is_diffuse) {                                                               // :524-538
        const float theta_i = dm::acos(cos_theta);
        const float theta_o = dm::acos(dot(new_d, normal));
        const float var2 = roughness * roughness;
        const float A = 1.0f - 0.5f * var2 / (var2 + 0.33f);
        const float B = 0.45f * var2 / (var2 + 0.09f);
        const float alpha = gmax(theta_i, theta_o);
        const float beta = gmin(theta_i, theta_o);
        const float w = (A + B * gmax(0.0f, theta_i - theta_o) * dm::sin(alpha) * dm::tan(beta)) * INV_PI_F * occlusion;
        ps.path = ps.path * (albedo * w);
    } else {                                                                               // :539-545
        ps.path = ps.path * (specular * (1.0f - roughness));
    }
    ps.d = new_d;

    bn.current = hit;
    bn.kind = PROBE_NONE;
    bn.c = mk3(0.0f, 0.0f, 0.0f);
    bn.w = 0.0f;
    bn.max_t = 0.0f;
    bn.probe_d = mk3(0.0f, 0.0f, 0.0f);
    if (coin_diffuse) {                                                                    // :553-554
        f3 lit;
        if (sample_light_setup(sc, ps.o, normal, ps.seed, bn.probe_d, bn.max_t, lit)) {
            bn.kind = PROBE_SHADOW;
            // global_light with spec_light = 0, raytracer.glsl:557
            const float sw = fresnel + 1.0f - roughness;
            bn.c = ((albedo * lit) * roughness + (specular * mk3(0.0f, 0.0f, 0.0f)) * sw) * 2.0f;
        }
    } else {                                                                               // :555-556, :398-401
        const f3 viewDir = normalize(ps.o - old_o);
        bn.probe_d = reflect(viewDir, normalize(random_sphere(ps.seed) * roughness + normal));
        bn.kind = PROBE_GLOSSY;
        bn.c = specular;
        bn.w = fresnel + 1.0f - roughness;
        // albedo * diff_light * roughness with diff_light = 0 contributes (albedo*0)*roughness;
        // that term is +-0 for finite inputs and is dropped in shade_post.
    }
}

// Light a glossy probe returns, raytracer.glsl:420-437.  `light_id` < 0: nothing was hit at all.
RT_DEV f3 glossy_light(const DScene &sc, const DFrame &fr, f3 probe_d, int light_id, f4 isec) {
    if (light_id < 0) return sky_color(sc, fr, probe_d);
    const float inv_det = 1.0f / isec.w;
    const ShadeRec ls = load_shade(sc, light_id);
    const Mat lm = load_mat(sc, ls.mat);
    const f3 N = calc_normal(ls, isec.x * inv_det, isec.y * inv_det);
    return lm.emission * gmax(dot(probe_d, N), 0.0f);
}

// After the probe, raytracer.glsl:557-579.  `has_light` and `global_light` are the probe's
// outcome.  Returns true when the path continues with another bounce.
RT_DEV bool shade_post(const DScene &sc, PathState &ps, int current, bool has_light, f3 global_light) {
    if (has_light) ps.energy = ps.energy + ps.path * global_light;
    const float4 g0 = sc.geo[3 * current];
    if (dot(ps.d, mk3(g0.x, g0.y, g0.z)) <= 0.0f) return false;                            // :568-571
    const float brightness =
        gclamp(gmax(dot(global_light, ps.path), (ps.path.x + ps.path.y + ps.path.z) / 3.0f), 1.0f / 8.0f, 1.0f);
    if (unit_float(ps.seed) > brightness) return false;                                    // :576
    ps.path = mk3(ps.path.x / brightness, ps.path.y / brightness, ps.path.z / brightness);
    return true;
}

// global_light of a glossy probe that returned `spec_light`, raytracer.glsl:557 with diff_light = 0.
RT_DEV f3 glossy_global_light(const Bounce &bn, f3 spec_light) {
    return ((bn.c * spec_light) * bn.w) * 2.0f;
}

}  // namespace rt
