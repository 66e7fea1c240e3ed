// rt_device.h - device-side building blocks of the gfx950 path tracer.
//
// Arithmetic contract: IEEE float32 with separate multiplies and adds (the translation unit is
// compiled with -ffp-contract=off), correctly rounded divisions and square roots, denormals kept.
// Where GLSL leaves a choice to the GL implementation, the choice is the one Mesa 23.2 (llvmpipe)
// makes when it compiles the reference's shader, because that is the implementation under which
// the unmodified shader can be executed and compared with: dot products are summed from the last
// component to the first, fma() is a separate multiply and add, and the elementary functions and
// the bilinear texture filter are those of rt_glslmath.h / bilinear_mix below.  Under that contract
// the device reproduces the llvmpipe run of the reference shader bit for bit.  Comments starting
// with "Mesa:" mark these choices.
//
// Each function names the lines of the reference shader (res/shader/raytracer.glsl) whose
// behaviour it implements.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_glslmath.h"

#ifndef RT_DOT_ORDER
#define RT_DOT_ORDER 1   // 0: (x*x' + y*y') + z*z'    1: (z*z' + y*y') + x*x'  (Mesa)
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
// Mesa: max/min are IEEE maxNum/minNum (a NaN operand yields the other operand)
RT_DEV float gmax(float x, float y) { return __builtin_fmaxf(x, y); }
RT_DEV float gmin(float x, float y) { return __builtin_fminf(x, y); }
RT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

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
    // LDS addresses (plain numbers, 0 = none) of a workgroup's copy of planes 0-3 of `shade` (what every hit reads) and of
    // the material table: kernels of resident scenes stage them (stage_shading) and read them with LDS instructions
    uint32_t shade_lds, mats_lds;
    uint32_t light_geo_lds;      // the same for the geometry records of the emissive triangles (the first `lights` ones), 0 = none
    const float4 *shade03;       // planes 0-3 of `shade` (what every hit reads): the same memory, or a workgroup's LDS copy of it
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
    uint32_t skip_nonfinite;      // option "skip_nonfinite": samples whose radiance is NaN or infinite are not added to the sums
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
    gm::sincos(theta, s, c);
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
// The same for an index the whole wave shares and that the compiler knows to be small (a 16-bit id from a candidate
// list): i * 48 of such a number becomes a 24-bit multiplication, which only the vector unit has, and the product has to
// be carried back to the scalar unit for the load.  The empty asm hides how small the index is: a scalar multiplication.
RT_DEV Tri load_tri_uniform(const float4 *geo, uint32_t i) {
    asm volatile("" : "+s"(i));
    const uint32_t off = i * 48u;
    const const_v4f_ptr g = (const_v4f_ptr)((const __attribute__((address_space(4))) char *)geo + off);
    const v4f a = g[0], b = g[1], c = g[2];
    Tri t;
    t.n = mk3(a.x, a.y, a.z);
    t.p = mk3(a.w, b.x, b.y);
    t.u = mk3(b.z, b.w, c.x);
    t.v = mk3(c.y, c.z, c.w);
    return t;
}
// Out-of-range indices read as an all-zero record (the shader's robust-buffer behaviour).
RT_DEV Tri load_tri_checked(const DScene &sc, int i) {
    if (sc.light_geo_lds && i >= 0 && i < sc.lights && i < sc.count) {
        typedef const __attribute__((address_space(3))) v4f *lds_f4_ptr;
        const lds_f4_ptr g = (lds_f4_ptr)(uintptr_t)sc.light_geo_lds;
        const v4f a = g[3 * i + 0], b = g[3 * i + 1], c = g[3 * i + 2];
        Tri t;
        t.n = mk3(a.x, a.y, a.z);
        t.p = mk3(a.w, b.x, b.y);
        t.u = mk3(b.z, b.w, c.x);
        t.v = mk3(c.y, c.z, c.w);
        return t;
    }
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
        float4 a, b, c, d;
        if (sc.shade_lds) {
            typedef const __attribute__((address_space(3))) v4f *lds_f4_ptr;
            const lds_f4_ptr p = (lds_f4_ptr)(uintptr_t)sc.shade_lds;
            const v4f va = p[0 * sc.count + i], vb = p[1 * sc.count + i], vc = p[2 * sc.count + i], vd = p[3 * sc.count + i];
            a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w);
            c = make_float4(vc.x, vc.y, vc.z, vc.w); d = make_float4(vd.x, vd.y, vd.z, vd.w);
        } else {
            a = sc.shade03[(size_t)0 * sc.count + i];
            b = sc.shade03[(size_t)1 * sc.count + i];
            c = sc.shade03[(size_t)2 * sc.count + i];
            d = sc.shade03[(size_t)3 * sc.count + i];
        }
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

struct Mat { f3 albedo, specular, emission; float roughness, ior; bool textured; };      // textured: hasTexture[m] == 1

RT_DEV Mat load_mat(const DScene &sc, uint32_t m) {
    Mat r;
    if (m < (uint32_t)sc.n_mat) {
        float4 a, b, c;
        if (sc.mats_lds) {
            typedef const __attribute__((address_space(3))) v4f *lds_f4_ptr;
            const lds_f4_ptr p = (lds_f4_ptr)(uintptr_t)sc.mats_lds;
            const v4f va = p[3 * m + 0], vb = p[3 * m + 1], vc = p[3 * m + 2];
            a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w); c = make_float4(vc.x, vc.y, vc.z, vc.w);
            r.textured = __float_as_int(va.w) == 1;        // the LDS copy carries the flag in the record's unused word
        } else {
            a = sc.mats[3 * m + 0]; b = sc.mats[3 * m + 1]; c = sc.mats[3 * m + 2];
            r.textured = sc.has_tex[m] == 1;
        }
        r.albedo = mk3(a.x, a.y, a.z);
        r.specular = mk3(b.x, b.y, b.z);
        r.roughness = b.w;
        r.emission = mk3(c.x, c.y, c.z);
        r.ior = c.w;
    } else {
        r.albedo = r.specular = r.emission = mk3(0.0f, 0.0f, 0.0f);
        r.roughness = r.ior = 0.0f;
        r.textured = false;
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

// LINEAR filtering with REPEAT wrap.  Mesa (llvmpipe), per axis of length L:
//   L a power of two:  x = c*L - 0.5;  i = floor(x);  texels i & (L-1), (i+1) & (L-1)
//   otherwise:         c = c - floor(c);  x = c*L - 0.5;  i = floor(x);  texels (x < 0 ? L-1 : i), (i+1 == L ? 0 : i+1)
//   weight w = x - floor(x);  lerp(w, v0, v1) = fma(w, v1 - v0, v0), along s for both rows, then along t.
// A coordinate whose texel index does not fit an int (NaN included) uses index 0; its weight follows
// from the same arithmetic (NaN for NaN and infinities, which makes the sample NaN).
RT_DEV void wrap_repeat_linear(float c, int L, int &i0, int &i1, float &w) {
    const bool pot = (L & (L - 1)) == 0;
    if (!pot) c = c - __builtin_floorf(c);
    const float x = c * (float)L - 0.5f;
    const float fl = __builtin_floorf(x);
    w = x - fl;
    const bool in_range = __builtin_fabsf(fl) < 2147483648.0f;
    const int i = in_range ? (int)fl : 0;
    if (pot) {
        i0 = i & (L - 1);
        i1 = (i + 1) & (L - 1);
    } else {
        i0 = (x < 0.0f) ? L - 1 : i;
        i1 = (i + 1 >= L) ? 0 : i + 1;
        if (!in_range || i0 >= L) { i0 = 0; i1 = 0; }
    }
}
RT_DEV Bilinear bilinear_setup(float s, float t, int W, int H) {
    Bilinear bl;
    wrap_repeat_linear(s, W, bl.i0, bl.i1, bl.a);
    wrap_repeat_linear(t, H, bl.j0, bl.j1, bl.b);
    return bl;
}
RT_DEV float lerp1(float w, float v0, float v1) { return __builtin_fmaf(w, v1 - v0, v0); }
RT_DEV f3 bilinear_mix(f3 t00, f3 t10, f3 t01, f3 t11, float a, float b) {
    const f3 top = mk3(lerp1(a, t00.x, t10.x), lerp1(a, t00.y, t10.y), lerp1(a, t00.z, t10.z));
    const f3 bot = mk3(lerp1(a, t01.x, t11.x), lerp1(a, t01.y, t11.y), lerp1(a, t01.z, t11.z));
    return mk3(lerp1(b, top.x, bot.x), lerp1(b, top.y, bot.y), lerp1(b, top.z, bot.z));
}

RT_DEV f3 sample_env(const DScene &sc, float s, float t) {
    if (sc.env_w == 0 || sc.env_h == 0) return mk3(0.0f, 0.0f, 0.0f);
    const Bilinear bl = bilinear_setup(s, t, sc.env_w, sc.env_h);
    const float4 a = sc.env[(size_t)bl.j0 * sc.env_w + bl.i0], b = sc.env[(size_t)bl.j0 * sc.env_w + bl.i1];
    const float4 c = sc.env[(size_t)bl.j1 * sc.env_w + bl.i0], d = sc.env[(size_t)bl.j1 * sc.env_w + bl.i1];
    return bilinear_mix(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), bl.a, bl.b);
}

// The three layers of a material (albedo, normal, ARM) at one texture coordinate, bilinear with repeat wrap
// (raytracer.glsl:471-487: three texture() calls).  The accesses of the three are arranged so that they wait for memory
// together (the three layer pointers in one go, then the twelve texels in one go) instead of one layer after the other:
// six dependent round trips become two.  A layer that does not exist, or was never uploaded, yields zero (its lanes read
// a texel of some valid array instead).
// (the pointers are read from memory: told to be device memory, the texel reads are global loads, not generic-address ones)
typedef const __attribute__((address_space(1))) uint32_t *texels_ptr;
struct AtlasLayers { texels_ptr p0, p1, p2; bool ok0, ok1, ok2; };
// The layer pointers of a material.
RT_DEV AtlasLayers atlas_layers_of(const DScene &sc, uint32_t first_layer) {
    AtlasLayers al;
    al.p0 = al.p1 = al.p2 = nullptr;
    al.ok0 = al.ok1 = al.ok2 = false;
    if (sc.atlas_layers <= 0) return al;                    // (the same for every lane)
    const uint32_t n_layers = (uint32_t)sc.atlas_layers;
    const bool in0 = first_layer < n_layers, in1 = first_layer + 1u < n_layers, in2 = first_layer + 2u < n_layers;
    al.p0 = (texels_ptr)sc.atlas[in0 ? first_layer : 0u];
    al.p1 = (texels_ptr)sc.atlas[in1 ? first_layer + 1u : 0u];
    al.p2 = (texels_ptr)sc.atlas[in2 ? first_layer + 2u : 0u];
    al.ok0 = in0 && al.p0 != nullptr; al.ok1 = in1 && al.p1 != nullptr; al.ok2 = in2 && al.p2 != nullptr;
    return al;
}
RT_DEV void sample_atlas3(const DScene &sc, const AtlasLayers &al, float s, float t, f3 &r0, f3 &r1, f3 &r2) {
    r0 = r1 = r2 = mk3(0.0f, 0.0f, 0.0f);
    if (sc.atlas_layers <= 0) return;
    const bool ok0 = al.ok0, ok1 = al.ok1, ok2 = al.ok2;
    const texels_ptr spare = (texels_ptr)(const void *)sc.shade;
    const int S = sc.atlas_size;
    const Bilinear bl = bilinear_setup(s, t, S, S);
    const size_t o00 = (size_t)bl.j0 * S + bl.i0, o10 = (size_t)bl.j0 * S + bl.i1, o01 = (size_t)bl.j1 * S + bl.i0, o11 = (size_t)bl.j1 * S + bl.i1;
    const texels_ptr q0 = ok0 ? al.p0 : spare, q1 = ok1 ? al.p1 : spare, q2 = ok2 ? al.p2 : spare;
    const uint32_t a0 = q0[ok0 ? o00 : 0], b0 = q0[ok0 ? o10 : 0], c0 = q0[ok0 ? o01 : 0], d0 = q0[ok0 ? o11 : 0];
    const uint32_t a1 = q1[ok1 ? o00 : 0], b1 = q1[ok1 ? o10 : 0], c1 = q1[ok1 ? o01 : 0], d1 = q1[ok1 ? o11 : 0];
    const uint32_t a2 = q2[ok2 ? o00 : 0], b2 = q2[ok2 ? o10 : 0], c2 = q2[ok2 ? o01 : 0], d2 = q2[ok2 ? o11 : 0];
    if (ok0) r0 = bilinear_mix(unpack_r11g11b10(a0), unpack_r11g11b10(b0), unpack_r11g11b10(c0), unpack_r11g11b10(d0), bl.a, bl.b);
    if (ok1) r1 = bilinear_mix(unpack_r11g11b10(a1), unpack_r11g11b10(b1), unpack_r11g11b10(c1), unpack_r11g11b10(d1), bl.a, bl.b);
    if (ok2) r2 = bilinear_mix(unpack_r11g11b10(a2), unpack_r11g11b10(b2), unpack_r11g11b10(c2), unpack_r11g11b10(d2), bl.a, bl.b);
}

// skyColor, raytracer.glsl:273-276
RT_DEV f3 sky_color(const DScene &sc, const DFrame &fr, f3 dir) {
    const float s = gm::atan2(dir.z, dir.x) * INV_PI_F * 0.5f + 0.5f;
    const float t = -gm::asin(dir.y) * INV_PI_F + 0.5f;
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
#if RT_DOT_ORDER == 0
    return (r0 * r0 + r1 * r1) * 0.5f;
#else
    return (r1 * r1 + r0 * r0) * 0.5f;
#endif
}

// ---------------------------------------------------------------- camera, raytracer.glsl:586-607

RT_DEV uint32_t pixel_seed(const DFrame &fr, uint32_t tx, uint32_t ty, uint32_t sample) {
    return fr.clock + (tx * fr.height + ty + 394587u) * (sample + 1u) * 13u + sample;    // :594
}

// Origin of every camera ray: CAMERA * (0, 0, 0, 1) as the shader's matrix product evaluates it (raytracer.glsl:604).
RT_DEV f3 camera_origin(const DFrame &fr) {
    const float *M = fr.cam;
    float o4[3];
#pragma unroll
    for (int r = 0; r < 3; r++) o4[r] = ((M[0 + r] * 0.0f + M[4 + r] * 0.0f) + M[8 + r] * 0.0f) + M[12 + r] * 1.0f;
    return mk3(o4[0], o4[1], o4[2]);
}

RT_DEV Ray camera_ray(const DFrame &fr, uint32_t tx, uint32_t ty, uint32_t &seed, const float *matrix = nullptr) {
    const float W = (float)fr.width, H = (float)fr.height;
    const float inv_width = 1.0f / W, inv_height = 1.0f / H;
    const float j1 = unit_float(seed) * inv_width;
    const float j2 = unit_float(seed) * inv_height;
    const f3 d0 = mk3(((float)(int)tx - 0.5f * W + 0.5f) * inv_height, ((float)(int)ty + 0.5f) * inv_height - 0.5f, 0.5f);
    const f3 dtctor = normalize(d0 + mk3(j1, j2, 0.0f));
    const f3 dir3 = normalize(dtctor - mk3(0.0f, 0.0f, 0.0f));
    const float *M = matrix ? matrix : fr.cam;
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

// The shader's sampleLight / sampleLightGlossy leave their `out vec3 light` unwritten when they return
// false (raytracer.glsl:333-334, :369-370, :376-377, :420-425), yet the value still reaches
// global_light (:557) and through it the Russian-roulette brightness (:575).  Mesa: after inlining,
// each of the two functions keeps one `light` variable for the whole pixel-sample, so an unwritten
// result is the value that function wrote at an earlier depth (zero before its first write).
// kept_diff / kept_spec are those two variables.
struct PathState {
    f3 o, d;          // current ray
    f3 path;          // throughput
    f3 energy;        // radiance gathered so far
    uint32_t seed;
    f3 kept_diff, kept_spec;
};

struct Bounce {
    int kind;         // ProbeKind
    int current;      // triangle the path sits on
    f3 probe_d;       // probe ray direction (origin = PathState::o after shade_pre)
    float max_t;      // shadow probes: distance to the sampled light point
    // coin = sampleLight: lit = the light the call returns if nothing occludes; albedo and roughness
    //                     are the factors of global_light (:557)
    // coin = sampleLightGlossy: c = specular colour, w = fresnel_reflectance + 1 - roughness
    f3 lit, albedo;
    float roughness;
    f3 c;
    float w;
};

// global_light (:557) of a bounce whose coin chose sampleLight: spec_light is zero, so the second term
// (specular * 0) * w is a zero of either sign for finite inputs; shade_pre folds what it does for other inputs into `albedo`.
RT_DEV f3 diffuse_global_light(f3 albedo, float roughness, f3 diff_light) {
    return ((albedo * diff_light) * roughness) * 2.0f;
}
// ... and of a bounce whose coin chose sampleLightGlossy (the term (albedo * 0) * roughness folded into `specular`).
RT_DEV f3 glossy_global_light(f3 specular, float w, f3 spec_light) {
    return ((specular * spec_light) * w) * 2.0f;
}

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
    const f3 rand_tri = (lt.u * cx + lt.p) + lt.v * cy;     // UVP * vec3(coords, 1); Mesa: the constant-1 column is added second
    const f3 delta = rand_tri - position;
    max_t = length(delta);
    probe_d = mk3(delta.x / max_t, delta.y / max_t, delta.z / max_t);
    const float DdN = dot(probe_d, normal);
    const float DdL = dot(probe_d, lt.n);
    if (DdN <= 0.0f || DdL >= 0.0f) return false;
    const ShadeRec ls = load_shade(sc, light_id);               // :379
    const Mat lm = load_mat(sc, ls.mat);
    const float tri_area = 0.5f * length(cross(lt.u, lt.v));
    const float distance_ratio = max_t * (max_t + 2.0f) + 1.0f;      // fma() of :385; Mesa: separate multiply and add
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
    const bool has_texture = material.textured;

    if (hit < sc.lights) ps.energy = ps.energy + ps.path * material.emission;               // :463-466

    const f3 N = calc_normal(tri, hu, hv);
    f3 normal = N, albedo = material.albedo, specular = material.specular;
    float occlusion = 1.0f, roughness = material.roughness;
    float metallic = (roughness < 0.125f) ? 1.0f : 0.0f;
    if (has_texture) {                                                                     // :471-487
        const float ts = (tri.tex_p[0] * 1.0f + tri.tex_u[0] * hu) + tri.tex_v[0] * hv;
        const float tt = (tri.tex_p[1] * 1.0f + tri.tex_u[1] * hu) + tri.tex_v[1] * hv;
        // (the tangents and the layer pointers wait for memory together; asking for them before the material is known,
        // with the shading planes and the material record, was measured as well: -0.5 % on 1 M textured triangles)
        const float4 t0 = sc.shade[(size_t)4 * sc.count + hit], t1 = sc.shade[(size_t)5 * sc.count + hit],
                     t2 = sc.shade[(size_t)6 * sc.count + hit];
        const AtlasLayers layers = atlas_layers_of(sc, mat_id * 3u);
        const float w0 = 1.0f - hu - hv;
        f3 T = (mk3(t0.x, t0.y, t0.z) * w0 + mk3(t1.x, t1.y, t1.z) * hu) + mk3(t2.x, t2.y, t2.z) * hv;
        T = normalize(T - N * dot(N, T));
        const f3 Bt = cross(T, N);
        f3 texel_albedo, tn, texel_arm;
        sample_atlas3(sc, layers, ts, tt, texel_albedo, tn, texel_arm);
        const f3 tex_normal = normalize(mk3(tn.x * 2.0f - 1.0f, tn.y * 2.0f - 1.0f, tn.z * 2.0f - 1.0f));
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
    // fma() of :497; Mesa: separate multiply and add
    const float fresnel = (1.0f - transmission) * gm::pow(1.0f - cos_theta, 5.0f) + transmission;

    const float rand_reflectance = unit_float(ps.seed);
    const float rand_scatter = unit_float(ps.seed);
    const float rand_metallic = unit_float(ps.seed);

    const f3 travelled = ps.d * ht;
    ps.o = ps.o + travelled;                                                               // :507

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
        gm::sincos(th, sn, cs);
        new_d = normalize(mk3(sinx * cs, sinx * sn, x) * scale + base);                    // :490-492
    }
    if (is_mirror) {                                                                       // :509-514
        const float k = 1.0f - roughness;
        ps.path = ps.path * ((rand_metallic < metallic) ? mk3(k, k, k) : specular);
    } else if (is_metal) {                                                                 // :516-522
        ps.path = ps.path * albedo;
    } else if (is_diffuse) {                                                               // :524-538
        const float theta_i = gm::acos(cos_theta);
        const float theta_o = gm::acos(dot(new_d, normal));
        const float var2 = roughness * roughness;
        const float A = 1.0f - 0.5f * var2 / (var2 + 0.33f);
        const float B = 0.45f * var2 / (var2 + 0.09f);
        const float alpha = gmax(theta_i, theta_o);
        const float beta = gmin(theta_i, theta_o);
        const float w = (A + B * gmax(0.0f, theta_i - theta_o) * gm::sin(alpha) * gm::tan(beta)) * INV_PI_F * occlusion;
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
    bn.lit = mk3(0.0f, 0.0f, 0.0f);
    bn.albedo = albedo;
    bn.roughness = roughness;
    // global_light (:557) is a sum of a diffuse and a specular term, and the term of the function the coin did not
    // choose is (colour * 0) * factor: a zero for finite inputs, but NaN when a texel made the colour or the factor
    // infinite or NaN.  That is all the term can do, so it is folded into the colour of the term that is kept
    // (x + 0 = x, x + NaN = NaN) instead of being carried through the probe.
    const float spec_w = fresnel + 1.0f - roughness;
    if (coin_diffuse) {                                                                    // :553-554
        bn.albedo = albedo + (specular * 0.0f) * spec_w;
        if (sample_light_setup(sc, ps.o, normal, ps.seed, bn.probe_d, bn.max_t, bn.lit)) bn.kind = PROBE_SHADOW;
    } else {                                                                               // :555-556, :398-401
        // position - view.position; Mesa: simplified to direction * t, the vector that was just added
        const f3 viewDir = normalize(travelled);
        bn.probe_d = reflect(viewDir, normalize(random_sphere(ps.seed) * roughness + normal));
        bn.kind = PROBE_GLOSSY;
        bn.c = specular + (albedo * 0.0f) * roughness;
        bn.w = spec_w;
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

// Rest of a bounce whose coin chose sampleLight.  `written`: the call returned true (the probe ran and
// nothing occluded), so its result replaces the kept one.
RT_DEV bool finish_diffuse(const DScene &sc, PathState &ps, int current, bool written, f3 lit, f3 albedo,
                           float roughness) {
    if (written) ps.kept_diff = lit;
    return shade_post(sc, ps, current, written, diffuse_global_light(albedo, roughness, ps.kept_diff));
}
// Rest of a bounce whose coin chose sampleLightGlossy.  outcome 0: a non-emissive triangle was hit (nothing
// written, false), 1: nothing was hit (sky), 2: the closest hit is the emissive triangle light_id.
RT_DEV bool finish_glossy(const DScene &sc, const DFrame &fr, PathState &ps, int current, int outcome, f3 probe_d,
                          int light_id, f4 lsec, f3 specular, float w) {
    if (outcome != 0) ps.kept_spec = glossy_light(sc, fr, probe_d, outcome == 2 ? light_id : -1, lsec);
    return shade_post(sc, ps, current, outcome != 0, glossy_global_light(specular, w, ps.kept_spec));
}

}  // namespace rt
