// Objects.hpp - host-side triangle record and its pre-processing.
//
// Mirrors the reference's Triangle (3Dobjects.hpp:36-91): what the path tracer needs of a
// triangle is derived once on the CPU, in float32 with separate multiplies and adds:
// u = p1 - p0, v = p2 - p0, true_normal = cross(u, v) (not normalised), texture-space deltas, and one
// Gram-Schmidt tangent per vertex.  The two device records (rt_triangle_model, rt_triangle_shading in
// gpurt.h) are the reference's TriangleModel / TriangleShading (3Dobjects.hpp:102-135).
#pragma once

#include <cmath>
#include <cstdint>

#include "../../../include/gpurt.h"

namespace host {

struct vec2 { float x = 0.0f, y = 0.0f; };
struct vec3 { float x = 0.0f, y = 0.0f, z = 0.0f; };

inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return (1.0f / std::sqrt(dot(a, a))) * a; }

struct Triangle {
    Triangle() = default;
    // 3Dobjects.hpp:39-55
    Triangle(vec3 p0, vec3 p1, vec3 p2, vec3 n0, vec3 n1, vec3 n2, vec2 t0, vec2 t1, vec2 t2, uint32_t mat)
        : position(p0), u(p1 - p0), v(p2 - p0), normals{n0, n1, n2}, true_normal(cross(u, v)), tex_p(t0),
          tex_u(t1 - t0), tex_v(t2 - t0), material_id(mat) {
        for (int i = 0; i < 3; i++) {
            const vec3 T = tex_v.x * u - tex_u.x * v;
            tangents[i] = normalize(T - dot(normals[i], T) * normals[i]);   // NaN for degenerate uv, as in the reference
        }
    }
    float area_key() const { return length(cross(u, v)); }   // sort key of Scene.cpp:334

    vec3 position, u, v;
    vec3 normals[3];
    vec3 tangents[3];
    vec3 true_normal;
    vec2 tex_p, tex_u, tex_v;
    uint32_t material_id = 0;
};

inline rt_triangle_model to_model(const Triangle &t) {
    rt_triangle_model m;
    const vec3 *src[4] = {&t.true_normal, &t.position, &t.u, &t.v};
    float *dst[4] = {m.true_normal, m.position, m.u, m.v};
    for (int k = 0; k < 4; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    return m;
}
inline rt_triangle_shading to_shading(const Triangle &t) {
    rt_triangle_shading s;
    s.material_id = t.material_id;
    for (int k = 0; k < 3; k++) {
        s.normals[k][0] = t.normals[k].x; s.normals[k][1] = t.normals[k].y; s.normals[k][2] = t.normals[k].z;
        s.tangents[k][0] = t.tangents[k].x; s.tangents[k][1] = t.tangents[k].y; s.tangents[k][2] = t.tangents[k].z;
    }
    s.tex_p[0] = t.tex_p.x; s.tex_p[1] = t.tex_p.y;
    s.tex_u[0] = t.tex_u.x; s.tex_u[1] = t.tex_u.y;
    s.tex_v[0] = t.tex_v.x; s.tex_v[1] = t.tex_v.y;
    return s;
}

}  // namespace host
