// rt_kernels.h - gfx950 kernels of the path tracer (included once, by rt_api.hip).
//
// Mapping: one lane = one ray/path, one wave64 = 64 rays.  In the brute-force loops the
// triangle index is the same in every lane, so triangle records are read with scalar
// loads into SGPRs and the vector ALU only runs the intersection arithmetic.
#pragma once

#include "rt_device.h"

namespace rt {

// Device-side work counters (see rt_stats in gpurt.h).
struct DStats {
    unsigned long long pixel_samples, rays, shadow_probes, glossy_probes;
    // hierarchy traversal work: node visits and triangle tests of the closest-hit and of the probe stages
    unsigned long long ext_nodes, ext_tris, probe_nodes, probe_tris;
    unsigned long long skipped;       // pixel-samples left out of the sums (option "skip_nonfinite")
};
// NaN or infinite in any channel (x - x is NaN for both)
RT_DEV bool nonfinite3(float x, float y, float z) { return !(((x - x) + (y - y)) + (z - z) == 0.0f); }

// The per-pass table travels by value in the kernel arguments (2 KB), so a pass needs no host-to-device copy and no
// staging buffer: the SAMPLE of slot k of the batch is first_sample + k * stride, and light_seed[k] is its light seed =
// CLOCK hashed SAMPLE times (raytracer.glsl:595-598).  The coin of a bounce (raytracer.glsl:553: unitFloat(light_seed)
// < 0.5, drawn once per bounce from a stream all pixels of a sample share) is drawn from light_seed where it is needed; the
// staged kernels of deeper bounces get a copy of the table whose seeds the host has advanced to their depth.
#ifndef RT_MAX_BATCH
#define RT_MAX_BATCH 512
#endif
constexpr int MAX_BATCH = RT_MAX_BATCH;
struct PassTables {
    uint32_t n, first_sample, stride;
    uint32_t light_seed[MAX_BATCH];
};
RT_DEV uint32_t sample_of_slot(const PassTables &tab, uint32_t k) { return tab.first_sample + k * tab.stride; }
// Draws the bounce's coin: true = sampleLight, false = sampleLightGlossy.
RT_DEV bool draw_coin(uint32_t &light_seed) { return unit_float(light_seed) < 0.5f; }

RT_DEV unsigned long long wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return (unsigned long long)v;
}

// ---------------------------------------------------------------- brute-force loops

// In all three loops the plane tests of a triangle run for the whole wave; the costlier inside
// test runs only if at least one lane passed them (a ballot), which is rare for the coherent rays
// of a camera tile.  Skipping it for a wave changes nothing: a lane that fails the plane tests never
// updates its state.  The any-hit loops look for "no lane
// left" only once per block of BF_BLOCK triangles, which keeps the inner loop free of exits.

constexpr int BF_BLOCK = 16;

// findIntersection, raytracer.glsl:217-235.  Returns the triangle id (-1 = miss); isec as left
// by the fold (x, y, relative_depth, determinant).
RT_DEV int closest_hit_range(const float4 *geo, int begin, int end, const Ray &ray, bool active, int current,
                             f4 &isec, int best) {
    for (int i = begin; i < end; i++) {
        const Tri t = load_tri(geo, i);
        PlaneTest pt;
        const bool p = active & plane_test_closest(ray, t, isec, pt) & (i != current);
        if (__builtin_amdgcn_ballot_w64(p) != 0ull) {
            float rx, ry;
            if (p & inside_test(ray, t, pt, rx, ry)) {
                isec.x = rx; isec.y = ry; isec.z = pt.rd; isec.w = pt.det;
                best = i;
            }
        }
    }
    return best;
}

// The same fold for few rays over many triangles: one ray at a time, and the 64 lanes of the wave test 64 consecutive
// triangles of it each round.  Whether a triangle is a candidate (front-facing, in front of the origin, hit inside,
// not `current`) does not depend on the running state; the candidates of a round are then folded in ascending index
// order with the shader's "not farther" comparison, the same expression on the same values as in closest_hit_range,
// so the result is the same bits.  This is what the hierarchy falls back to for the rare ray it cannot decide (three
// or more candidates within the depth margin, non-finite rays): with closest_hit_range one such ray kept its wave busy
// for all COUNT triangles, 50 ms and more on a million of them, as long again as the whole launch it is part of.
// `todo` selects the lanes whose rays are redone; their isec (initialised by the caller) and best are replaced.
RT_DEV void closest_hit_range_by_wave(const float4 *geo, int begin, int end, const Ray &ray, bool todo, int current, f4 &isec,
                                      int &best) {
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long left = __builtin_amdgcn_ballot_w64(todo);
    while (left != 0ull) {
        const int src = (int)__builtin_ctzll(left);
        left &= left - 1ull;
        Ray r;
        r.o = mk3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.o.x), src)),
                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.o.y), src)),
                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.o.z), src)));
        r.d = mk3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.d.x), src)),
                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.d.y), src)),
                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ray.d.z), src)));
        const int cur = __builtin_amdgcn_readlane(current, src);
        f4 st;
        st.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isec.x), src));
        st.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isec.y), src));
        st.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isec.z), src));
        st.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isec.w), src));
        int b = __builtin_amdgcn_readlane(best, src);
        for (int i0 = begin; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool in_range = i < end;
            const float4 *g = geo + 3 * (size_t)(in_range ? i : begin);
            const float4 qa = g[0], qb = g[1], qc = g[2];
            Tri t;
            t.n = mk3(qa.x, qa.y, qa.z);
            t.p = mk3(qa.w, qb.x, qb.y);
            t.u = mk3(qb.z, qb.w, qc.x);
            t.v = mk3(qc.y, qc.z, qc.w);
            PlaneTest pt;
            pt.det = -dot(r.d, t.n);
            pt.delta = r.o - t.p;
            pt.rd = dot(t.n, pt.delta);
            float rx, ry;
            const bool inside = inside_test(r, t, pt, rx, ry);
            const bool candidate = in_range & (i != cur) & !((pt.det <= 0.0f) | (pt.rd <= 0.0f)) & inside;
            unsigned long long m = __builtin_amdgcn_ballot_w64(candidate);
            while (m != 0ull) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                const float rd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pt.rd), l));
                const float det = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pt.det), l));
                if (!((st.z >= 0.0f) & (rd * st.w > st.z * det))) {
                    st.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rx), l));
                    st.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ry), l));
                    st.z = rd;
                    st.w = det;
                    b = i0 + l;
                }
            }
        }
        if (lane == src) { isec = st; best = b; }
    }
}

RT_DEV int closest_hit_bf(const DScene &sc, const Ray &ray, f4 &isec) {
    isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 1.0f;
    return closest_hit_range(sc.geo, 0, sc.count, ray, true, -1, isec, -1);
}

// Occlusion loop of sampleLight, raytracer.glsl:372-378.  `active` lanes run the loop; returns
// true for lanes whose probe ray is blocked.
RT_DEV bool shadow_any_hit_bf(const DScene &sc, const Ray &ray, float max_t, int current, bool active) {
    bool open = active;          // lane still looking for an occluder
    const int count = sc.count;
    for (int i0 = 0; i0 < count; i0 += BF_BLOCK) {
        if (__builtin_amdgcn_ballot_w64(open) == 0ull) break;
        const int i1 = (i0 + BF_BLOCK < count) ? i0 + BF_BLOCK : count;
        for (int i = i0; i < i1; i++) {
            const Tri t = load_tri(sc.geo, i);
            PlaneTest pt;
            const bool p = open & plane_test_shadow(ray, t, max_t, pt) & (i != current);
            if (__builtin_amdgcn_ballot_w64(p) != 0ull) {
                float rx, ry;
                open = open & !(p & inside_test(ray, t, pt, rx, ry));
            }
        }
    }
    return active & !open;
}

// Any-hit of the closest-hit test against a fixed state `isec` over triangles [begin, end):
// true for lanes where some triangle != current passes (second phase of sampleLightGlossy).
RT_DEV bool closer_any_hit_range(const float4 *geo, int begin, int end, const Ray &ray, const f4 &isec, int current,
                                 bool active) {
    bool open = active;
    for (int i0 = begin; i0 < end; i0 += BF_BLOCK) {
        if (__builtin_amdgcn_ballot_w64(open) == 0ull) break;
        const int i1 = (i0 + BF_BLOCK < end) ? i0 + BF_BLOCK : end;
        for (int i = i0; i < i1; i++) {
            const Tri t = load_tri(geo, i);
            PlaneTest pt;
            const bool p = open & plane_test_closest(ray, t, isec, pt) & (i != current);
            if (__builtin_amdgcn_ballot_w64(p) != 0ull) {
                float rx, ry;
                open = open & !(p & inside_test(ray, t, pt, rx, ry));
            }
        }
    }
    return active & !open;
}

// Search loop of sampleLightGlossy, raytracer.glsl:403-419, in its two-phase form: triangles
// [0, lights) fold to the closest light hit; any triangle of [lights, count) that passes the test
// against that state ends the search as "occluded".  Outcome: 0 occluded, 1 nothing hit,
// 2 light `light_id` visible with `isec`.
RT_DEV int glossy_search_bf(const DScene &sc, const Ray &ray, int current, bool active, int &light_id, f4 &isec) {
    isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 0.0f;
    const int count = sc.count, lights = sc.lights < sc.count ? sc.lights : sc.count;
    light_id = closest_hit_range(sc.geo, 0, lights, ray, active, current, isec, -1);
    const bool blocked = closer_any_hit_range(sc.geo, lights, count, ray, isec, current, active);
    if (blocked) return 0;
    return light_id < 0 ? 1 : 2;
}

// ---------------------------------------------------------------- single-kernel path tracer
//
// One wave per 8x8 pixel tile (the reference's workgroup shape, raytracer.glsl:3), one lane per
// pixel; a lane walks its whole path.  Simple and exact, but lanes whose path has ended idle until
// the longest path of the wave is done; the staged pipeline below removes that.
__global__ __launch_bounds__(64) void k_trace_mega(DScene sc, DFrame fr, float4 *__restrict__ sum, PassTables tab,
                                                   DStats *__restrict__ stats) {
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    const uint32_t tile = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t tx = (tile % tiles_x) * 8u + (lane & 7u);
    const uint32_t ty = (tile / tiles_x) * 8u + (lane >> 3);
    const bool in_image = (tx < fr.width) && (ty < fr.height);

    uint32_t n_rays = 0, n_shadow = 0, n_glossy = 0, n_paths = 0, n_skipped = 0;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    const size_t pix = (size_t)ty * fr.width + tx;
    if (in_image) {
        const float4 s = sum[pix];
        acc = mk3(s.x, s.y, s.z);
    }
    const int max_depth = fr.recursion > 1 ? fr.recursion : 1;

    for (uint32_t k = 0; k < tab.n; k++) {
        const uint32_t sample = sample_of_slot(tab, k);
        uint32_t light_seed = tab.light_seed[k];
        PathState ps;
        ps.seed = pixel_seed(fr, tx, ty, sample);
        const Ray cam = camera_ray(fr, tx, ty, ps.seed);
        ps.o = cam.o; ps.d = cam.d;
        ps.path = mk3(1.0f, 1.0f, 1.0f);
        ps.energy = mk3(0.0f, 0.0f, 0.0f);
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        bool alive = in_image;
        n_paths += alive ? 1u : 0u;

        for (int depth = 0; depth < max_depth; depth++) {
            if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
            const bool coin_diffuse = draw_coin(light_seed);               // :553, same for every pixel
            Ray ray; ray.o = ps.o; ray.d = ps.d;
            f4 isec;
            const int hit = closest_hit_bf(sc, ray, isec);
            n_rays += alive ? 1u : 0u;

            Bounce bn;
            bn.kind = PROBE_NONE; bn.current = 0; bn.max_t = 0.0f; bn.w = 0.0f; bn.roughness = 0.0f;
            bn.c = bn.lit = bn.albedo = bn.probe_d = mk3(0.0f, 0.0f, 0.0f);
            bool shaded = false;
            if (alive) {
                if (hit < 0) {
                    ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;     // :452-456
                    alive = false;
                } else {
                    const float inv_det = 1.0f / isec.w;
                    shade_pre(sc, fr, ps, hit, isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, coin_diffuse, bn);
                    shaded = true;
                }
            }
            Ray probe; probe.o = ps.o; probe.d = bn.probe_d;
            if (coin_diffuse) {
                const bool run = shaded && bn.kind == PROBE_SHADOW;
                bool blocked = true;
                if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                    blocked = shadow_any_hit_bf(sc, probe, bn.max_t, bn.current, run);
                    n_shadow += run ? 1u : 0u;
                }
                if (shaded) alive = finish_diffuse(sc, ps, bn.current, run && !blocked, bn.lit, bn.albedo, bn.roughness);
            } else {
                const bool run = shaded && bn.kind == PROBE_GLOSSY;
                int outcome = 0, light_id = -1;
                f4 lsec;
                lsec.x = lsec.y = lsec.z = 0.0f; lsec.w = 1.0f;
                if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                    outcome = glossy_search_bf(sc, probe, bn.current, run, light_id, lsec);
                    n_glossy += run ? 1u : 0u;
                }
                if (shaded)
                    alive = finish_glossy(sc, fr, ps, bn.current, run ? outcome : 0, bn.probe_d, light_id, lsec, bn.c, bn.w);
            }
        }
        if (fr.skip_nonfinite && nonfinite3(ps.energy.x, ps.energy.y, ps.energy.z)) n_skipped += in_image ? 1u : 0u;
        else acc = acc + ps.energy;
    }
    if (in_image) sum[pix] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    if (fr.skip_nonfinite) {
        const unsigned long long k = wave_sum_u32(n_skipped);
        if (lane == 0 && k) atomicAdd(&stats->skipped, k);
    }

    const unsigned long long r = wave_sum_u32(n_rays), s = wave_sum_u32(n_shadow), g = wave_sum_u32(n_glossy),
                             p = wave_sum_u32(n_paths);
    if (lane == 0) {
        atomicAdd(&stats->rays, r);
        atomicAdd(&stats->shadow_probes, s);
        atomicAdd(&stats->glossy_probes, g);
        atomicAdd(&stats->pixel_samples, p);
    }
}

// ---------------------------------------------------------------- closest hit for a ray batch

__global__ __launch_bounds__(256) void k_closest_hit_batch(DScene sc, const float *__restrict__ o,
                                                           const float *__restrict__ d, unsigned long long n,
                                                           int *__restrict__ id, float *__restrict__ uvt) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const unsigned long long j = live ? i : 0ull;
    Ray ray;
    ray.o = mk3(o[j * 3 + 0], o[j * 3 + 1], o[j * 3 + 2]);
    ray.d = mk3(d[j * 3 + 0], d[j * 3 + 1], d[j * 3 + 2]);
    f4 isec;
    const int hit = closest_hit_bf(sc, ray, isec);
    if (live) {
        const float inv_det = 1.0f / isec.w;
        id[i] = hit;
        uvt[i * 3 + 0] = isec.x * inv_det;
        uvt[i * 3 + 1] = isec.y * inv_det;
        uvt[i * 3 + 2] = isec.z * inv_det;
    }
}

// 8-bit output of the mean image.  aces = 0: the display path, displayQuad.fs:19-34 (sRGB transfer, rounded to
// the nearest code).  aces = 1: the file path of Scene::exportBMP, Scene.cpp:564-567: ACESFilm (Scene.cpp:66-73),
// then the sRGB transfer (Scene.cpp:58-64), then clamp(v * 256, 0, 255) truncated.
__global__ void k_srgb8(const float4 *__restrict__ sum, float inv_n, unsigned long long n_pixels, int aces,
                        unsigned char *__restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const float4 s = sum[i];
    const float c[3] = {s.x * inv_n, s.y * inv_n, s.z * inv_n};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float x = c[k];
        if (aces) {
            x = (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f);
            x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        }
        float v = (x <= 0.0031308f) ? x * 12.92f : 1.055f * __powf(x, 1.0f / 2.4f) - 0.055f;
        v = aces ? v * 256.0f : v * 255.0f + 0.5f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        out[i * 3 + k] = (unsigned char)v;
    }
}

}  // namespace rt
