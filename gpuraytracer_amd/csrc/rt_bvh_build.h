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
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/gpurt.h"

namespace rt {

struct HostBvh {
    std::vector<float> nodes;          // 16 floats per node
    std::vector<uint32_t> qnodes;      // 8 words per node: the same nodes with their planes on a 15-bit grid (quantise_nodes)
    float qbase[3] = {0.0f, 0.0f, 0.0f}, qscale[3] = {1.0f, 1.0f, 1.0f};
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
#ifndef RT_LARGE_LEAF
#define RT_LARGE_LEAF 2
#endif
#ifndef RT_MID_LEAF
#define RT_MID_LEAF 3
#endif
inline int max_leaf_for(uint32_t count) { return count >= 65536u ? RT_LARGE_LEAF : (count < 2048u ? RT_SMALL_LEAF : RT_MID_LEAF); }
#ifndef RT_BVH_BINS
#define RT_BVH_BINS 32
#endif
constexpr int BINS = RT_BVH_BINS;
constexpr uint32_t PARALLEL_BUILD_MIN = 32768;     // triangles from which on subtrees are built by several threads

}  // namespace bvh_detail

// Builds the hierarchy over `count` triangles.  `force_median`: split by object median only (bounded
// depth); used when the SAH tree would come out deeper than the traversal stack.
inline void build_bvh_impl(const rt_triangle_model *tris, uint32_t count, bool force_median, HostBvh &result) {
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
        if (n2 > 0.0 && n2 < result.min_normal_len2) result.min_normal_len2 = n2;
    }
    result.nodes.clear();
    result.leaf_geo.clear();
    result.leaf_ids.clear();
    result.depth = 0;
    result.max_abs_coord = max_abs;

    struct Task { uint32_t begin, end; int32_t parent; int slot; int depth; };
    // Builds the subtree of `first` into `out` (a HostBvh or a Fragment: node and leaf-slot numbers count from the start
    // of `out`).  With `deferred`, ranges of `defer_n` items or fewer are not built but listed, their box written to the
    // parent and their link left open: they are built by other threads and appended afterwards.
    auto build_tasks = [&](auto &out, const Task &first, uint32_t defer_n, std::vector<Task> *deferred) {
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

    std::vector<Task> stack;
    // the first task always becomes an inner node (a scene of <= MAX_LEAF triangles gets one leaf child and one empty child)
    stack.push_back(first);
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        out.depth = std::max(out.depth, t.depth);
        const uint32_t n = t.end - t.begin;
        if (deferred && t.parent >= 0 && n > (uint32_t)MAX_LEAF && n <= defer_n) {
            float *nd = &out.nodes[(size_t)t.parent * 16];
            Box b;
            b.reset();
            for (uint32_t i = t.begin; i < t.end; i++) b.grow(items[i].box);
            b = pad_box(b);
            float *dst = nd + 6 * t.slot;
            dst[0] = b.lo[0]; dst[1] = b.lo[1]; dst[2] = b.lo[2]; dst[3] = b.hi[0]; dst[4] = b.hi[1]; dst[5] = b.hi[2];
            deferred->push_back(t);
            continue;
        }
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
    };

    // Large scenes: the top of the tree is built here, subtrees of count / 64 triangles or fewer by a pool of threads
    // (their item ranges are disjoint), then appended in the order they were met.  The tree does not depend on the
    // number of threads; only the numbering of the leaf slots differs from a build in one go.
    struct Fragment { std::vector<float> nodes, leaf_geo; std::vector<int32_t> leaf_ids; int depth = 0; };
    const uint32_t defer_n = count >= PARALLEL_BUILD_MIN ? count / 64u : 0u;
    std::vector<Task> deferred;
    build_tasks(result, Task{0, count, -1, 0, 1}, defer_n, defer_n ? &deferred : nullptr);
    if (deferred.empty()) return;
    std::vector<Fragment> parts(deferred.size());
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (size_t k = next.fetch_add(1); k < deferred.size(); k = next.fetch_add(1)) {
            Task root = deferred[k];
            root.parent = -1;
            build_tasks(parts[k], root, 0u, nullptr);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_threads = std::min<size_t>(deferred.size(), std::max(1u, std::min(hw ? hw : 1u, 16u)));
    std::vector<std::thread> pool;
    for (size_t k = 1; k < n_threads; k++) pool.emplace_back(worker);
    worker();
    for (std::thread &th : pool) th.join();
    for (size_t k = 0; k < deferred.size(); k++) {
        const Fragment &f = parts[k];
        const int32_t node0 = (int32_t)(result.nodes.size() / 16);
        const uint32_t slot0 = (uint32_t)result.leaf_ids.size();
        const size_t at = result.nodes.size();
        result.nodes.insert(result.nodes.end(), f.nodes.begin(), f.nodes.end());
        for (size_t n = 0; n < f.nodes.size() / 16; n++)
            for (int c = 0; c < 2; c++) {
                int32_t link;
                float *where = &result.nodes[at + n * 16 + 12 + c];
                memcpy(&link, where, 4);
                if (link >= 0) {
                    link += node0;
                } else {
                    const uint32_t code = (uint32_t)~link;
                    link = ~(int32_t)((((code >> 3) + slot0) << 3) | (code & 7u));
                }
                memcpy(where, &link, 4);
            }
        result.leaf_geo.insert(result.leaf_geo.end(), f.leaf_geo.begin(), f.leaf_geo.end());
        result.leaf_ids.insert(result.leaf_ids.end(), f.leaf_ids.begin(), f.leaf_ids.end());
        memcpy(&result.nodes[(size_t)deferred[k].parent * 16 + 12 + deferred[k].slot], &node0, 4);
        result.depth = std::max(result.depth, f.depth);
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

// The 32-byte form of the nodes that the traversal reads from memory: every box plane as a 15-bit position on a grid
// over the bounds of the whole hierarchy, lower planes rounded down and upper planes up, so that a quantised box
// contains its box.  A word holds the two planes of one child along one axis, lo | hi << 16, each 0x8000 | position:
// under the exponent byte 0x3F the two bytes read as the float 1 + position / 32768 (rt_bvh.h: plane_of), and
// plane = qbase + position / 32768 * qscale.  Node = (x of child 0, x of child 1, y, y, z, z, link 0, link 1).
// A grid step is 1/32768 of the scene's extent: 2 to 4 % of a triangle's size in a regular mesh of a million triangles.
inline void quantise_nodes(HostBvh &b) {
    const size_t n = b.nodes.size() / 16;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t k = 0; k < n; k++)
        for (int c = 0; c < 2; c++)
            for (int a = 0; a < 3; a++) {
                const float l = b.nodes[16 * k + 6 * c + a], h = b.nodes[16 * k + 6 * c + 3 + a];
                if (std::isfinite(l) && std::isfinite(h) && l <= h) { lo[a] = std::min(lo[a], (double)l); hi[a] = std::max(hi[a], (double)h); }
            }
    for (int a = 0; a < 3; a++) {
        if (!(lo[a] <= hi[a])) { lo[a] = 0.0; hi[a] = 0.0; }
        const double ext = hi[a] - lo[a];
        b.qbase[a] = std::nextafterf((float)lo[a], -INFINITY);
        // a little more than the extent, so that the upper bound lies below the last grid position
        b.qscale[a] = (float)std::max((hi[a] - (double)b.qbase[a]) * (1.0 + 1.0 / 1024.0), std::max(std::fabs(lo[a]), std::fabs(hi[a])) * 1e-6 + 1e-30);
        (void)ext;
    }
    b.qnodes.assign(n * 8, 0u);
    for (size_t k = 0; k < n; k++) {
        for (int c = 0; c < 2; c++)
            for (int a = 0; a < 3; a++) {
                const float l = b.nodes[16 * k + 6 * c + a], h = b.nodes[16 * k + 6 * c + 3 + a];
                uint32_t ql = 32767u, qh = 0u;       // an empty box stays empty: no ray enters it
                if (l <= h) {
                    const double base = b.qbase[a], scale = b.qscale[a];
                    double pl = std::floor(((double)l - base) / scale * 32768.0), ph = std::ceil(((double)h - base) / scale * 32768.0);
                    pl = std::min(std::max(pl, 0.0), 32767.0);
                    ph = std::min(std::max(ph, 0.0), 32767.0);
                    // as the device decodes them (exactly, up to what the widening of the boxes per ray covers)
                    while (pl > 0.0 && base + pl / 32768.0 * scale > (double)l) pl -= 1.0;
                    while (ph < 32767.0 && base + ph / 32768.0 * scale < (double)h) ph += 1.0;
                    ql = (uint32_t)pl;
                    qh = (uint32_t)ph;
                }
                b.qnodes[8 * k + 2 * a + c] = (0x8000u | ql) | ((0x8000u | qh) << 16);
            }
        memcpy(&b.qnodes[8 * k + 6], &b.nodes[16 * k + 12], 8);
    }
}

inline void build_bvh(const rt_triangle_model *tris, uint32_t count, HostBvh &out) {
    build_bvh_impl(tris, count, false, out);
    if (out.depth >= BVH_STACK_DEPTH - 2) build_bvh_impl(tris, count, true, out);
    reorder_breadth_first(out);
    quantise_nodes(out);
}

}  // namespace rt
