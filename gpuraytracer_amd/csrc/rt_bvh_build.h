// rt_bvh_build.h - host-side construction of the bounding volume hierarchy used by the
// accelerated closest-hit / any-hit kernels.
//
// The reference has no acceleration structure: every query loops over all COUNT triangles
// (raytracer.glsl:225-229, :372-378, :406-419).  The hierarchy only decides WHICH triangles a ray
// is tested against; the test itself and the order-dependent fold stay those of the shader (see
// rt_bvh.h for why the result is identical).
//
// Layout handed to the device:
//   nodes   4 x float4 per inner node:  (lo0.xyz, hi0.x) (hi0.yz, lo1.xy) (lo1.z, hi1.xyz) (c0, c1, -, -)
//           child >= 0: inner node index;  child < 0: leaf, ~child = (first << 3) | count, count <= 4
//   leaves  one 64-byte record per triangle in leaf order: 3 x float4 in the TriangleModel layout, then
//           (original triangle index, -, -, -)
// Boxes are padded by a few ulps of their largest coordinate so that float rounding of p + u, p + v
// can never put a vertex outside its box.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/gpurt.h"

namespace rt {

struct HostBvh {
    std::vector<float> nodes;          // 16 floats per node
    std::vector<float> leaf_geo;       // 16 floats per leaf slot: the 12 of the triangle record, its index, padding
    std::vector<int32_t> leaf_ids;     // original triangle index per leaf slot
    int depth = 0;
    float max_abs_coord = 0.0f;        // largest |coordinate| of any vertex
    // Smallest non-zero |true_normal|^2 of any triangle.  The reference compares depths as cross-multiplied products
    // rd_a * det_b (raytracer.glsl:150), both factors scaling with |normal|; where such products underflow the
    // comparison is "not farther" for every pair and the later index wins whatever the depths, which a traversal
    // that prunes by depth does not reproduce.  Callers keep to the loop over all triangles below TINY_NORMAL_LEN2.
    double min_normal_len2 = INFINITY;
};
constexpr double TINY_NORMAL_LEN2 = 8.271806125530277e-25;      // 2^-80: |cross(u, v)| below 2^-40, edges below about 1e-6

namespace bvh_detail {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const float *p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box &b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0.0f) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Item { Box box; float c[3]; int32_t id; };

// Largest leaf, by scene size: triangle tests cost about as much as node steps, and the balance moves towards smaller
// leaves as traversals get longer (measured on MI355X: 424 triangles 5 = 4 > 3 > 2 > 6; 12 030 triangles 3 = 5 > 4, 6;
// 1 M triangles 2 > 3 > 4).  The leaf code holds up to 7.
#ifndef RT_SMALL_LEAF
#define RT_SMALL_LEAF 4
#endif
inline int max_leaf_for(uint32_t count) { return count >= 65536u ? 2 : (count < 2048u ? RT_SMALL_LEAF : 3); }
#ifndef RT_BVH_BINS
#define RT_BVH_BINS 32
#endif
constexpr int BINS = RT_BVH_BINS;

}  // namespace bvh_detail

// Builds the hierarchy over `count` triangles.  `force_median`: split by object median only (bounded
// depth); used when the SAH tree would come out deeper than the traversal stack.
inline void build_bvh_impl(const rt_triangle_model *tris, uint32_t count, bool force_median, HostBvh &out) {
    using namespace bvh_detail;
    const int MAX_LEAF = max_leaf_for(count);
    std::vector<Item> items(count);
    float max_abs = 0.0f;
    for (uint32_t i = 0; i < count; i++) {
        const rt_triangle_model &t = tris[i];
        float v[3][3];
        for (int a = 0; a < 3; a++) {
            v[0][a] = t.position[a];
            v[1][a] = t.position[a] + t.u[a];
            v[2][a] = t.position[a] + t.v[a];
        }
        Item &it = items[i];
        it.box.reset();
        for (int k = 0; k < 3; k++) it.box.grow(v[k]);
        for (int a = 0; a < 3; a++) {
            // non-finite vertices: keep the triangle reachable by every ray
            if (!std::isfinite(it.box.lo[a]) || !std::isfinite(it.box.hi[a])) { it.box.lo[a] = -INFINITY; it.box.hi[a] = INFINITY; }
            it.c[a] = 0.5f * it.box.lo[a] + 0.5f * it.box.hi[a];
            if (!std::isfinite(it.c[a])) it.c[a] = 0.0f;
            if (std::isfinite(it.box.lo[a])) max_abs = std::max(max_abs, std::max(std::fabs(it.box.lo[a]), std::fabs(it.box.hi[a])));
        }
        it.id = (int32_t)i;
        const double n2 = (double)t.true_normal[0] * t.true_normal[0] + (double)t.true_normal[1] * t.true_normal[1] +
                          (double)t.true_normal[2] * t.true_normal[2];
        if (n2 > 0.0 && n2 < out.min_normal_len2) out.min_normal_len2 = n2;
    }
    out.nodes.clear();
    out.leaf_geo.clear();
    out.leaf_ids.clear();
    out.depth = 0;
    out.max_abs_coord = max_abs;

    auto pad_box = [](Box b) {
        for (int a = 0; a < 3; a++) {
            const float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
            const float e = std::isfinite(m) ? m * 4.8e-7f + 1e-30f : 0.0f;     // 4 ulps
            b.lo[a] -= e;
            b.hi[a] += e;
        }
        return b;
    };
    auto make_leaf = [&](uint32_t begin, uint32_t end) -> int32_t {
        const uint32_t first = (uint32_t)out.leaf_ids.size();
        for (uint32_t i = begin; i < end; i++) {
            const rt_triangle_model &t = tris[items[i].id];
            const float *src = (const float *)&t;
            out.leaf_geo.insert(out.leaf_geo.end(), src, src + 12);
            float idbits;
            memcpy(&idbits, &items[i].id, 4);
            out.leaf_geo.push_back(idbits);
            out.leaf_geo.insert(out.leaf_geo.end(), 3, 0.0f);
            out.leaf_ids.push_back(items[i].id);
        }
        return ~(int32_t)((first << 3) | (end - begin));
    };

    struct Task { uint32_t begin, end; int32_t parent; int slot; int depth; };
    std::vector<Task> stack;
    // root: always an inner node (a scene of <= MAX_LEAF triangles gets one leaf child and one empty child)
    stack.push_back({0, count, -1, 0, 1});
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        out.depth = std::max(out.depth, t.depth);
        const uint32_t n = t.end - t.begin;
        int32_t self;
        if (n <= (uint32_t)MAX_LEAF && t.parent >= 0) {
            self = make_leaf(t.begin, t.end);
        } else {
            // choose a split
            Box cb;
            cb.reset();
            for (uint32_t i = t.begin; i < t.end; i++) cb.grow(items[i].c);
            int axis = 0;
            float ext = -1.0f;
            for (int a = 0; a < 3; a++) if (cb.hi[a] - cb.lo[a] > ext) { ext = cb.hi[a] - cb.lo[a]; axis = a; }
            uint32_t mid = t.begin + n / 2;
            bool split_found = false;
            if (!force_median && ext > 0.0f && n > 2) {
                float best_cost = INFINITY;
                int best_axis = -1, best_bin = -1;
                for (int a = 0; a < 3; a++) {
                    const float lo = cb.lo[a], w = cb.hi[a] - cb.lo[a];
                    if (!(w > 0.0f)) continue;
                    Box bb[BINS];
                    uint32_t bc[BINS];
                    for (int b = 0; b < BINS; b++) { bb[b].reset(); bc[b] = 0; }
                    const float scale = (float)BINS / w;
                    for (uint32_t i = t.begin; i < t.end; i++) {
                        int b = (int)((items[i].c[a] - lo) * scale);
                        b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                        bb[b].grow(items[i].box);
                        bc[b]++;
                    }
                    float right_area[BINS];
                    uint32_t right_cnt[BINS];
                    Box acc;
                    acc.reset();
                    uint32_t cnt = 0;
                    for (int b = BINS - 1; b > 0; b--) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.area(); right_cnt[b] = cnt; }
                    acc.reset();
                    cnt = 0;
                    for (int b = 0; b < BINS - 1; b++) {
                        acc.grow(bb[b]);
                        cnt += bc[b];
                        if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                        const float cost = acc.area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                    }
                }
                if (best_axis >= 0) {
                    const float lo = cb.lo[best_axis], w = cb.hi[best_axis] - cb.lo[best_axis];
                    const float scale = (float)BINS / w;
                    auto it = std::partition(items.begin() + t.begin, items.begin() + t.end, [&](const Item &x) {
                        int b = (int)((x.c[best_axis] - lo) * scale);
                        b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                        return b <= best_bin;
                    });
                    mid = (uint32_t)(it - items.begin());
                    split_found = mid > t.begin && mid < t.end;
                }
            }
            if (!split_found) {
                mid = t.begin + n / 2;
                if (n >= 2)
                    std::nth_element(items.begin() + t.begin, items.begin() + mid, items.begin() + t.end,
                                     [axis](const Item &x, const Item &y) { return x.c[axis] < y.c[axis]; });
            }
            self = (int32_t)(out.nodes.size() / 16);
            out.nodes.resize(out.nodes.size() + 16, 0.0f);
            if (n == 0 || (n <= (uint32_t)MAX_LEAF && t.parent < 0) || n == 1) {
                // tiny scene: child 0 = leaf with everything, child 1 = empty box
                float *nd = &out.nodes[(size_t)self * 16];
                Box b0;
                b0.reset();
                for (uint32_t i = t.begin; i < t.end; i++) b0.grow(items[i].box);
                b0 = pad_box(b0);
                const float v[12] = {b0.lo[0], b0.lo[1], b0.lo[2], b0.hi[0], b0.hi[1], b0.hi[2],
                                     INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                memcpy(nd, v, sizeof(v));
                const int32_t c0 = make_leaf(t.begin, t.end), c1 = ~(int32_t)0;
                memcpy(&nd[12], &c0, 4);
                memcpy(&nd[13], &c1, 4);
            } else {
                stack.push_back({mid, t.end, self, 1, t.depth + 1});
                stack.push_back({t.begin, mid, self, 0, t.depth + 1});
            }
        }
        if (t.parent >= 0) {
            float *nd = &out.nodes[(size_t)t.parent * 16];
            Box b;
            b.reset();
            for (uint32_t i = t.begin; i < t.end; i++) b.grow(items[i].box);
            b = pad_box(b);
            float *dst = nd + 6 * t.slot;
            dst[0] = b.lo[0]; dst[1] = b.lo[1]; dst[2] = b.lo[2]; dst[3] = b.hi[0]; dst[4] = b.hi[1]; dst[5] = b.hi[2];
            memcpy(&nd[12 + t.slot], &self, 4);
        }
    }
}

#ifndef RT_BVH_STACK
#define RT_BVH_STACK 32
#endif
constexpr int BVH_STACK_DEPTH = RT_BVH_STACK;   // entries of the traversal stack (rt_bvh.h)

// Renumbers the inner nodes breadth first, so that the nodes nearest the root (the ones every ray visits)
// are the first ones of the array: the kernels keep that prefix in LDS.
inline void reorder_breadth_first(HostBvh &b) {
    const size_t n = b.nodes.size() / 16;
    if (n < 2) return;
    std::vector<int32_t> order, new_index(n, -1);
    order.reserve(n);
    order.push_back(0);
    for (size_t head = 0; head < order.size(); head++) {
        const float *nd = &b.nodes[(size_t)order[head] * 16];
        for (int c = 0; c < 2; c++) {
            int32_t child;
            memcpy(&child, &nd[12 + c], 4);
            if (child >= 0) order.push_back(child);
        }
    }
    for (size_t i = 0; i < order.size(); i++) new_index[order[i]] = (int32_t)i;
    std::vector<float> nodes(b.nodes.size());
    for (size_t i = 0; i < order.size(); i++) {
        float *dst = &nodes[i * 16];
        memcpy(dst, &b.nodes[(size_t)order[i] * 16], 64);
        for (int c = 0; c < 2; c++) {
            int32_t child;
            memcpy(&child, &dst[12 + c], 4);
            if (child >= 0) { child = new_index[child]; memcpy(&dst[12 + c], &child, 4); }
        }
    }
    b.nodes.swap(nodes);
}

inline void build_bvh(const rt_triangle_model *tris, uint32_t count, HostBvh &out) {
    build_bvh_impl(tris, count, false, out);
    if (out.depth >= BVH_STACK_DEPTH - 2) build_bvh_impl(tris, count, true, out);
    reorder_breadth_first(out);
}

}  // namespace rt
