// rt_bvh.h - hierarchy traversal that reproduces the reference's brute-force results exactly.
//
// Why a hierarchy can be exact here.  The shader's closest hit (raytracer.glsl:217-235) is a
// sequential fold over the triangles in index order: triangle i replaces the running state iff it
// is front-facing, in front of the origin, inside, and NOT farther than the state, where "farther"
// is the cross-multiplied float comparison  rd_i * w > z * det_i  (equal depth: the later index
// wins).  Three facts make the fold reproducible from any visiting order:
//  1. A triangle that fails facing / in-front / inside never touches the state, so only the
//     "candidates" that pass those three tests matter, and a hierarchy with conservative boxes
//     visits every candidate of a ray.
//  2. Two candidates whose depths differ by far more than float rounding compare the same way in
//     either order, so every candidate clearly behind the closest one loses to it no matter when it
//     is met and cannot influence the end state.  (Margin used: relative 2^-12; rounding of the
//     compared products is below 2^-21.)
//  3. For a PAIR of candidates a < b the fold's verdict is a function of the pair alone:
//     b wins unless  rd_b*det_a > rd_a*det_b.  Evaluating that same expression gives the same
//     verdict whichever of the two is met first.
// So the traversal keeps one running best, applies rule 3 between it and each new candidate, and
// counts how many candidates lie within the 2^-12 margin of the smallest depth seen.  With at most two
// such candidates the running best IS the fold's result.  With three or more (a ray through a
// mesh vertex, stacked coplanar copies) the lane is re-run through the plain index-order loop.
//
// Any-hit queries (sampleLight's occlusion loop :372-378 and the second phase of
// sampleLightGlossy :406-419) are order-independent ORs of the same per-triangle tests, so they
// only need the conservative boxes.
//
// Conservative boxes: nodes are padded by 4 ulps at build time, and every ray widens them by
// r = 2^-16 * (largest |coordinate| of the scene and of its origin), far above the rounding of the
// triangle test (which works on the unrounded p, u, v), far below any geometric feature.
// Rays with a non-finite origin or direction take the index-order loop (for them "every comparison
// is false" makes the last triangle win, which no box test reproduces).  A direction component of zero (or
// below 2^-64) takes its axis out of the box test, see bvh_ray_setup.
#pragma once

#include "rt_kernels.h"

namespace rt {

struct DBvh {
    const float4 *nodes;     // 4 x float4 per node: the boxes in full precision (staged into LDS by resident hierarchies)
    // The nodes as the traversal reads them from memory: 32 bytes each, the 12 box planes as 15-bit positions on a
    // grid over the hierarchy's bounds, rounded outwards (see bvh_ray_setup).  2 x uint4 per node:
    //   (x planes of child 0, of child 1, y planes of child 0, of child 1) (z planes of child 0, of child 1, link 0, link 1)
    // a word of planes = lo | hi << 16, each 0x8000 | position.
    const uint4 *qnodes;
    float qbase[3], qscale[3];   // plane = qbase + position / 32768 * qscale
    const float4 *leaf_geo;  // 4 x float4 per leaf slot: triangle record + (index, -, -, -)
    const int *leaf_ids;
    float max_abs_coord;
    int enabled;
    int n_nodes;
    int n_slots;             // leaf slots (= triangles)
    int stack_entries;       // traversal stack entries per lane that this hierarchy and the one searched beside it need
};

#ifndef RT_BVH_STACK
#define RT_BVH_STACK 32
#endif
constexpr int BVH_STACK = RT_BVH_STACK;
constexpr float BVH_NEAR = 1.0f + 1.0f / 4096.0f;    // 2^-12 depth margin
constexpr int BVH_TOP_NODES = 192;                   // nodes nearest the root that a workgroup keeps in LDS (12 KB)
constexpr int BVH_LIGHT_TOP_NODES = 64;              // the same for the hierarchy over the emissive triangles (4 KB)

// Per-lane traversal state that lives in LDS: this lane's column of a stack array [BVH_STACK][64], and the
// workgroup's copy of the first n_top nodes of the hierarchy (breadth-first order, so these are the levels
// nearest the root).  Fetching those from LDS takes the most frequent node reads off the vector memory
// path, which is the busiest unit of the traversal kernels.
//
// RES ("resident"): the whole hierarchy lives in LDS, leaf triangle records included.  A scene of a few hundred
// triangles (all the reference's bundled scenes but one) fits next to the stacks, and then no step of a traversal
// touches the vector memory path: a leaf test waits for an LDS read (about 100 clocks) instead of an L1/L2 hit
// (several hundred), which matters because the traversal kernels are bound by the latency of their dependent
// accesses as much as by instruction issue (measured: 2 / 3 / 4 waves per SIMD take 18.3 / 14.2 / 12.6 ms).
typedef const __attribute__((address_space(3))) v4f *lds_v4f_ptr;
typedef const __attribute__((address_space(3))) char *lds_byte_ptr;
typedef const __attribute__((address_space(3))) float *lds_f32_ptr;
typedef const __attribute__((address_space(3))) int *lds_i32_ptr;
template <bool RES_>
struct BvhScratchT {
    static constexpr bool RES = RES_;
    int *stack;
    lds_v4f_ptr top;      // LDS address space: the reads become ds_read_b128, not generic-address loads
    int n_top;
    lds_v4f_ptr leaves;   // RES: the leaf records (4 x float4 per slot)
    int root;             // what a traversal starts with: node 0, or (RES) the LDS address of its record
    // memory form: the grid of the 15-bit planes in LDS, (qbase - qscale) x, y, z, then qscale x, y, z (bvh_stage_grid).
    // Uniform values that every search needs but that are not worth a scalar register each for the whole kernel: the
    // kernels' scalar registers do not hold all their arguments, and what overflows is parked in vector lanes.
    lds_f32_ptr grid;
};
typedef BvhScratchT<false> BvhScratch;

// Cooperative copy of the top of the hierarchy (32-byte nodes) into LDS; every thread of the workgroup calls it.
RT_DEV int bvh_stage_top(const DBvh &bv, v4f *top, int max_nodes = BVH_TOP_NODES) {
    const int n_top = bv.enabled ? (bv.n_nodes < max_nodes ? bv.n_nodes : max_nodes) : 0;
    for (int i = threadIdx.x; i < n_top * 2; i += blockDim.x) top[i] = ((const v4f *)bv.qnodes)[i];
    __syncthreads();
    return n_top;
}
RT_DEV void bvh_stage_grid(const DBvh &bv, float *grid) {
    if (threadIdx.x < 3) {
        grid[threadIdx.x] = bv.qbase[threadIdx.x] - bv.qscale[threadIdx.x];
        grid[3 + threadIdx.x] = bv.qscale[threadIdx.x];
    }
    __syncthreads();
}
// Node layout of a resident hierarchy (80 bytes, 20 words), made for slab tests without min/max.  Which plane of a
// box a ray meets first along an axis follows from the sign of its direction alone, so a lane can fetch "near" and
// "far" straight away if it may pick the address: per axis and child the words (lo, hi, lo) are stored, and a lane
// reads two consecutive words from offset 0 (direction >= 0: lo then hi) or from offset 1 (hi then lo).
//   words 0..5   x: lo0 hi0 lo0 lo1 hi1 lo1      words 6..11  y, 12..17 z likewise
//   words 18,19  child links: inner node = LDS address of its record (so that a step needs no base), leaf = the
//                 (negative) leaf code as it is
// That takes the 12 min/max of a node step away (they issue at half the rate of multiply-adds on this chip).
#ifndef RT_RES_NODE_WORDS
#define RT_RES_NODE_WORDS 20
#endif
constexpr int RES_NODE_WORDS = RT_RES_NODE_WORDS;     // 20 words of content; more = padding that changes the banks consecutive records start in
constexpr int RES_NODE_BYTES = 4 * RES_NODE_WORDS;
static_assert(RES_NODE_WORDS >= 20, "a resident node record holds 20 words");
// 16-byte units the resident node records of a hierarchy take (what follows them is read with 16-byte accesses)
RT_DEV int res_nodes_v4f(int n_nodes) { return (n_nodes * RES_NODE_WORDS + 3) / 4; }
RT_DEV void bvh_stage_nodes_resident(const DBvh &bv, float *dst) {
    const int n = bv.enabled ? bv.n_nodes * RES_NODE_WORDS : 0;
    const float *src = (const float *)bv.nodes;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int node = i / RES_NODE_WORDS, k = i - node * RES_NODE_WORDS;
        const float *s = src + 16 * node;        // (lo0.xyz, hi0.xyz, lo1.xyz, hi1.xyz, link0, link1, -, -)
        float v = 0.0f;
        if (k >= 20) {
        } else if (k < 18) {
            const int axis = k / 6, j = k - axis * 6;
            const int child = j / 3, which = (j - child * 3) == 1;      // 1 = hi
            v = s[6 * child + 3 * which + axis];
        } else {
            const int link = __float_as_int(s[12 + (k - 18)]);
            v = __int_as_float(link >= 0 ? link * RES_NODE_BYTES + (int)(uintptr_t)(lds_f32_ptr)dst : link);
        }
        dst[i] = v;
    }
    __syncthreads();
}
// The same for the leaf records of a resident hierarchy.
RT_DEV void bvh_stage_leaves(const DBvh &bv, v4f *leaves) {
    const int n = bv.enabled ? bv.n_slots * 4 : 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) leaves[i] = ((const v4f *)bv.leaf_geo)[i];
    __syncthreads();
}

struct BvhRay {
    // Slab test of a box widened by r: entry distance along an axis = near * inv - ca, exit distance = far * inv - cb,
    // where (near, far) are the box planes the ray meets first and last (which is which follows from the sign of its
    // direction alone), ca = (origin + r') * inv, cb = (origin - r') * inv with r' = r signed like the direction.
    // resident form: planes in scene units; off = byte offsets into a node record that select (near, far) per axis
    // memory form:   planes as numbers F in [1, 2) (15-bit grid positions, see DBvh::qnodes): inv is the reciprocal
    //                direction times the grid's extent and the constants hold the grid's origin; off = byte selectors
    //                (v_perm_b32) that pick the near plane of a word of planes, ^ PERM_FLIP the far one
    f3 ca, cb;
    f3 inv;
    int off_x, off_y, off_z;
    bool finite;
};
constexpr uint32_t PERM_NEAR_LO = 0x0701000Cu, PERM_NEAR_HI = 0x0703020Cu, PERM_FLIP = 0x00020200u;
constexpr uint32_t PLANE_EXPONENT = 0x3F000000u;       // with the position's 0x8000 in bit 23: 1.0f + position / 32768

// Rays the box tests can serve: finite, and with the origin and the scene inside +-2^60, so that with |1/d| capped at
// 2^64 (below) no product of a slab test overflows.  All others take the index-order loops.
constexpr float BVH_MAX_COORD = 0x1p60f;
constexpr float BVH_MAX_INV = 0x1p64f;
RT_DEV bool bvh_ray_ok(const Ray &ray, float max_abs_coord) {
    const float s = ((ray.o.x + ray.o.y) + ray.o.z) + ((ray.d.x + ray.d.y) + ray.d.z);
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.o.x), __builtin_fabsf(ray.o.y)),
                                    __builtin_fmaxf(__builtin_fabsf(ray.o.z), max_abs_coord));
    return ((s - s) == 0.0f) & (m < BVH_MAX_COORD);      // s - s is NaN when any component is NaN or infinite
}

template <bool RES, class WS>
RT_DEV BvhRay bvh_ray_setup(const Ray &ray, const DBvh &bv, const WS &ws) {
    const float max_abs_coord = bv.max_abs_coord;
    BvhRay b;
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.o.x), __builtin_fabsf(ray.o.y)),
                                    __builtin_fmaxf(__builtin_fabsf(ray.o.z), max_abs_coord));
    const float r = m * (1.0f / 65536.0f) + 1e-30f;
    // reciprocal to 1 ulp: the slab test is conservative by far more than that
    b.inv = mk3(__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z));
    // A direction component of (nearly) zero: its reciprocal is infinite or huge, the two terms of a slab test become
    // infinities, and their difference NaN or an infinity of the wrong sign.  Such an axis is left out of the test
    // instead (reciprocal 0, constants that put its planes at -max / +max): every box counts as entered along it,
    // which only costs a ray parallel to an axis some box tests.
    const bool flat_x = !(__builtin_fabsf(b.inv.x) <= BVH_MAX_INV), flat_y = !(__builtin_fabsf(b.inv.y) <= BVH_MAX_INV),
               flat_z = !(__builtin_fabsf(b.inv.z) <= BVH_MAX_INV);
    if (flat_x) b.inv.x = 0.0f;
    if (flat_y) b.inv.y = 0.0f;
    if (flat_z) b.inv.z = 0.0f;
    constexpr float FAR = 3.4028234664e38f;
    if (RES) {
        // r with the sign of the direction: the plane met first is lo for a positive direction (constant origin + r), hi
        // for a negative one (constant origin - r)
        const float rx = __builtin_copysignf(r, b.inv.x), ry = __builtin_copysignf(r, b.inv.y), rz = __builtin_copysignf(r, b.inv.z);
        b.ca = mk3((ray.o.x + rx) * b.inv.x, (ray.o.y + ry) * b.inv.y, (ray.o.z + rz) * b.inv.z);
        b.cb = mk3((ray.o.x - rx) * b.inv.x, (ray.o.y - ry) * b.inv.y, (ray.o.z - rz) * b.inv.z);
        if (flat_x) { b.ca.x = FAR; b.cb.x = -FAR; }
        if (flat_y) { b.ca.y = FAR; b.cb.y = -FAR; }
        if (flat_z) { b.ca.z = FAR; b.cb.z = -FAR; }
        b.off_x = (int)((__float_as_uint(b.inv.x) >> 29) & 4u);
        b.off_y = (int)((__float_as_uint(b.inv.y) >> 29) & 4u) + 24;
        b.off_z = (int)((__float_as_uint(b.inv.z) >> 29) & 4u) + 48;
    } else {
        // a plane is qbase + (F - 1) * qscale: F * (qscale * inv) - (origin +- r - (qbase - qscale)) * inv.  The rounding of
        // these few operations on coordinates stays as far below the widening r as that of the plain form.
        const float rx = __builtin_copysignf(r, b.inv.x), ry = __builtin_copysignf(r, b.inv.y), rz = __builtin_copysignf(r, b.inv.z);
        const f3 o = mk3(ray.o.x - ws.grid[0], ray.o.y - ws.grid[1], ray.o.z - ws.grid[2]);
        b.ca = mk3((o.x + rx) * b.inv.x, (o.y + ry) * b.inv.y, (o.z + rz) * b.inv.z);
        b.cb = mk3((o.x - rx) * b.inv.x, (o.y - ry) * b.inv.y, (o.z - rz) * b.inv.z);
        if (flat_x) { b.ca.x = FAR; b.cb.x = -FAR; }
        if (flat_y) { b.ca.y = FAR; b.cb.y = -FAR; }
        if (flat_z) { b.ca.z = FAR; b.cb.z = -FAR; }
        b.inv = mk3(b.inv.x * ws.grid[3], b.inv.y * ws.grid[4], b.inv.z * ws.grid[5]);
        b.off_x = (int)((__float_as_uint(b.inv.x) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
        b.off_y = (int)((__float_as_uint(b.inv.y) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
        b.off_z = (int)((__float_as_uint(b.inv.z) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
    }
    b.finite = bvh_ray_ok(ray, max_abs_coord);
    return b;
}
// The address offsets or byte selectors again, from the reciprocal direction (cheaper than carrying them to another lane).
template <bool RES>
RT_DEV void bvh_ray_offsets(BvhRay &b) {
    if (RES) {
        b.off_x = (int)((__float_as_uint(b.inv.x) >> 29) & 4u);
        b.off_y = (int)((__float_as_uint(b.inv.y) >> 29) & 4u) + 24;
        b.off_z = (int)((__float_as_uint(b.inv.z) >> 29) & 4u) + 48;
    } else {
        b.off_x = (int)((__float_as_uint(b.inv.x) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
        b.off_y = (int)((__float_as_uint(b.inv.y) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
        b.off_z = (int)((__float_as_uint(b.inv.z) >> 31) ? PERM_NEAR_HI : PERM_NEAR_LO);
    }
}
// A plane of a 32-byte node as the number F = 1 + position / 32768: the two bytes `sel` picks, under a fixed exponent.
RT_DEV float plane_of(uint32_t word, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(PLANE_EXPONENT, word, sel)); }

// The slab tests (in bvh_advance) use one fused multiply-add per plane; they only have to be conservative and take no
// part in the hit arithmetic.  bvh_ray_setup keeps infinities and NaN out of the terms.  The rounding of these operations
// (a few 2^-24 of |origin| * |inv|) is below 1/16 of the widening r * |inv| = 2^-16 * |origin| * |inv|.
struct BvhCounters { uint32_t nodes, tris; };
// Counting the node visits and triangle tests costs the traversal loops one vector instruction per step and the
// kernels four registers (1.4 % of the headline rate), so the product does not: the counters stay zero.  Two
// measurement builds do.  -DRT_COUNT_STEPS (libgpurt_steps.so): every lane counts its own steps.
// -DRT_COUNT_WAVE_STEPS (libgpurt_wavesteps.so): the first active lane adds 64 per step instead, i.e. the lane slots
// the wave spent; the ratio of the two builds' numbers is the lane utilisation of the traversal loops.
#if defined(RT_COUNT_WAVE_STEPS)
RT_DEV uint32_t bvh_count_step() {
    const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u)) == 0u ? 64u : 0u;
}
#elif defined(RT_COUNT_STEPS)
RT_DEV uint32_t bvh_count_step() { return 1u; }
#else
RT_DEV uint32_t bvh_count_step() { return 0u; }
#endif

// Traversal state of one lane; it survives between calls of bvh_advance, so that a wave can hand finished
// lanes new rays while the others are still under way.
struct TravState {
    int node, sp;
    bool run;
};
// The bottom entry of every stack is a leaf of no triangles: a lane that has nothing left to visit pops it like any
// other entry and leaves the node loop with it, so the loop needs no "is the stack empty" test and no per-lane
// "still running" bookkeeping (a compare, a select and four scalar operations per node step).  trav_begin puts it there.
constexpr int STACK_BOTTOM = -1;          // ~STACK_BOTTOM = leaf code 0: first slot 0, no triangles
template <class WS>
RT_DEV void trav_begin(const WS &ws, TravState &ts, bool run) {
    ws.stack[0] = STACK_BOTTOM;
    ts.node = ws.root; ts.sp = 1; ts.run = run;
}

// Advances every running lane of the wave.  `leaf(slot)` is called for every triangle slot of every leaf whose
// box the ray enters within t_limit(); it returns false to stop that lane's traversal (any-hit found).
// Returns when no lane runs any more, or as soon as at least `stop_at` lanes of the wave are not running
// (pass 65 to run to the end; `at_least_one`: not before one round of node steps and leaves has been done;
// `node_budget`: at most that many node steps per lane and round).
// "while-while" order: all lanes first descend through inner nodes until each holds a leaf (or has finished),
// then all leaves are processed together; this keeps box tests and triangle tests of a wave from
// interleaving lane by lane.
// LOOP_SHARE > 0: a round's node loop ends as soon as fewer than 1 / LOOP_SHARE of the running lanes still descend; the
// others' leaves are tested first and the stragglers go on in the next round.  Pays where a node step waits for memory
// (1 M triangles: the lane-refill extend kernel -20 %, the pass +3.9 %); where the nodes are in LDS the test itself, four
// instructions per step, costs more than the shorter loops save (424 triangles -5 %, 12 030 triangles -3 %).
// Leaf records fetched two at a time (1 M triangles +3.5 %, with the indices +1.3 % more; 12 030 triangles +1.5 %; resident
// scenes, RT_RES_LEAF_AHEAD, +0.4 %).
#ifndef RT_LEAF_AHEAD
#define RT_LEAF_AHEAD 1
#endif
constexpr bool LEAF_AHEAD = RT_LEAF_AHEAD != 0;
#ifndef RT_RES_LEAF_AHEAD
#define RT_RES_LEAF_AHEAD 1
#endif
template <bool BUDGETED = false, int LOOP_SHARE = 0, class WS, typename LeafFn, typename LimitFn>
RT_DEV void bvh_advance(const DBvh &bv, const BvhRay &br, const WS &ws, TravState &ts, int stop_at, LeafFn leaf,
                        LimitFn t_limit, BvhCounters &cnt, bool at_least_one = false, int node_budget = 1 << 30) {
    int *stack = ws.stack;
    int sp = ts.sp;
    int node = ts.node;
    bool run = ts.run;
    for (;; at_least_one = false) {
        const unsigned long long running = __builtin_amdgcn_ballot_w64(run);
        if (running == 0ull || (!at_least_one && 64 - (int)__popcll(running) >= stop_at)) break;
        // a divergent loop: lanes leave as they reach a leaf or finish, and the wave goes on when none is left
        int budget = node_budget;
        const int n_running = (int)__popcll(running);
        // the first free entry of the lane's stack as an LDS address (entries of a lane are 64 apart): a step then needs no
        // multiplication for its accesses
        typedef __attribute__((address_space(3))) int *lds_stack_ptr;
        lds_stack_ptr top = (lds_stack_ptr)stack + 64 * sp;
        while (run && node >= 0 && (!BUDGETED || budget-- > 0)) {      // `run` does not change in this loop
            float in0, out0, in1, out1;
            int c0, c1;
            cnt.nodes += bvh_count_step();
            const float lim = t_limit();
            if (WS::RES) {
                // `node` is the byte offset of the record; (near, far) per axis and child through the lane's own offsets
                const lds_byte_ptr rec = (lds_byte_ptr)(uintptr_t)node;
                const lds_f32_ptr px = (lds_f32_ptr)(rec + br.off_x), py = (lds_f32_ptr)(rec + br.off_y), pz = (lds_f32_ptr)(rec + br.off_z);
                const lds_i32_ptr links = (lds_i32_ptr)(rec + 72);
                const float n0x = px[0], f0x = px[1], n1x = px[3], f1x = px[4];
                const float n0y = py[0], f0y = py[1], n1y = py[3], f1y = py[4];
                const float n0z = pz[0], f0z = pz[1], n1z = pz[3], f1z = pz[4];
                c0 = links[0]; c1 = links[1];
#ifdef RT_EXTRA_LDS
                {   // measurement build: two more 8-byte LDS reads per node step, results unused (is the LDS pipe what limits the step?)
                    typedef volatile __attribute__((address_space(3))) unsigned long long *lds_v64_ptr;
                    (void)*(lds_v64_ptr)(rec + 0); (void)*(lds_v64_ptr)(rec + 24);
                }
#endif
                in0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(n0x, br.inv.x, -br.ca.x), __builtin_fmaf(n0y, br.inv.y, -br.ca.y)),
                                                      __builtin_fmaf(n0z, br.inv.z, -br.ca.z)), 0.0f);
                out0 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(f0x, br.inv.x, -br.cb.x), __builtin_fmaf(f0y, br.inv.y, -br.cb.y)),
                                                       __builtin_fmaf(f0z, br.inv.z, -br.cb.z)), lim);
                in1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(n1x, br.inv.x, -br.ca.x), __builtin_fmaf(n1y, br.inv.y, -br.ca.y)),
                                                      __builtin_fmaf(n1z, br.inv.z, -br.ca.z)), 0.0f);
                out1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(f1x, br.inv.x, -br.cb.x), __builtin_fmaf(f1y, br.inv.y, -br.cb.y)),
                                                       __builtin_fmaf(f1z, br.inv.z, -br.cb.z)), lim);
            } else {
                v4f q0, q1;
                if (node < ws.n_top) {
                    q0 = ws.top[2 * node + 0]; q1 = ws.top[2 * node + 1];
                } else {
                    const v4f *g = (const v4f *)bv.qnodes;
                    q0 = g[2 * node + 0]; q1 = g[2 * node + 1];
                }
                const uint32_t sx = (uint32_t)br.off_x, sy = (uint32_t)br.off_y, sz = (uint32_t)br.off_z;
                const uint32_t x0 = __float_as_uint(q0.x), x1 = __float_as_uint(q0.y), y0 = __float_as_uint(q0.z), y1 = __float_as_uint(q0.w),
                               z0 = __float_as_uint(q1.x), z1 = __float_as_uint(q1.y);
                c0 = __float_as_int(q1.z); c1 = __float_as_int(q1.w);
                in0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(plane_of(x0, sx), br.inv.x, -br.ca.x), __builtin_fmaf(plane_of(y0, sy), br.inv.y, -br.ca.y)),
                                                      __builtin_fmaf(plane_of(z0, sz), br.inv.z, -br.ca.z)), 0.0f);
                out0 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(plane_of(x0, sx ^ PERM_FLIP), br.inv.x, -br.cb.x), __builtin_fmaf(plane_of(y0, sy ^ PERM_FLIP), br.inv.y, -br.cb.y)),
                                                       __builtin_fmaf(plane_of(z0, sz ^ PERM_FLIP), br.inv.z, -br.cb.z)), lim);
                in1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(plane_of(x1, sx), br.inv.x, -br.ca.x), __builtin_fmaf(plane_of(y1, sy), br.inv.y, -br.ca.y)),
                                                      __builtin_fmaf(plane_of(z1, sz), br.inv.z, -br.ca.z)), 0.0f);
                out1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(plane_of(x1, sx ^ PERM_FLIP), br.inv.x, -br.cb.x), __builtin_fmaf(plane_of(y1, sy ^ PERM_FLIP), br.inv.y, -br.cb.y)),
                                                       __builtin_fmaf(plane_of(z1, sz ^ PERM_FLIP), br.inv.z, -br.cb.z)), lim);
            }
            // the decisions as wave masks: their combinations then cost scalar instructions, and each
            // per-lane choice a single select
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(in0 <= out0);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(in1 <= out1);
            const unsigned long long first0 = __builtin_amdgcn_ballot_w64(in0 <= in1);
            const unsigned long long both_m = m0 & m1, any_m = m0 | m1;
            const bool both = __builtin_amdgcn_inverse_ballot_w64(both_m);
            // the child to descend into: the nearer one of two, or the only one entered
            const int near = __builtin_amdgcn_inverse_ballot_w64(m0 & (first0 | ~m1)) ? c0 : c1;
            // the farther child goes on the stack if both are entered; stored in any case (slot sp is free), which
            // spares the step a branch
            *top = __builtin_amdgcn_inverse_ballot_w64(first0) ? c1 : c0;
            top += both ? 64 : 0;
            // neither child entered: on to the top of the stack (at the bottom: the empty leaf)
            const bool pop = __builtin_amdgcn_inverse_ballot_w64(~any_m);
            top -= pop ? 64 : 0;
            int next = near;
            if (pop) next = *top;
            node = next;
            // the lanes that still descend have become few: the others' leaves first, these go on in the next round
            if (LOOP_SHARE > 0 && LOOP_SHARE * (int)__popcll(__builtin_amdgcn_ballot_w64(node >= 0)) < n_running) break;
        }
        sp = (int)(((uint32_t)(uintptr_t)top - (uint32_t)(uintptr_t)(lds_stack_ptr)stack) >> 8);
        if (run && (!(BUDGETED || LOOP_SHARE > 0) || node < 0)) {      // a leaf
            const int code = ~node;
            const int first = code >> 3, n = code & 7;
            bool go_on = true;
            if ((!WS::RES || RT_RES_LEAF_AHEAD) && LEAF_AHEAD) {
                // the first two triangles of the leaf are asked for together, so that a leaf of two costs one wait for memory
                // (or for LDS) instead of two; records in memory: their indices as well, so that a triangle that passes its
                // test does not wait for memory once more (leaf_id)
                const int second = first + (n > 1 ? 1 : 0);
                const Tri t0 = load_leaf_tri(bv, ws, first), t1 = load_leaf_tri(bv, ws, second);
                const int i0 = leaf_tri_id(bv, ws, first), i1 = leaf_tri_id(bv, ws, second);
                if (n > 0) { cnt.tris += bvh_count_step(); go_on = leaf(first, t0, i0); }
                if (n > 1) { cnt.tris += bvh_count_step(); if (go_on) go_on = leaf(first + 1, t1, i1); }
                for (int k = 2; k < n; k++) {
                    cnt.tris += bvh_count_step();
                    if (go_on) go_on = leaf(first + k, load_leaf_tri(bv, ws, first + k), leaf_tri_id(bv, ws, first + k));
                }
            } else {
                for (int k = 0; k < n; k++) {
                    cnt.tris += bvh_count_step();
                    if (go_on) go_on = leaf(first + k, load_leaf_tri(bv, ws, first + k), 0);
                }
            }
            if (go_on && sp > 1) {
                sp--;
                node = stack[64 * sp];
            } else {
                run = false;
            }
        }
    }
    ts.sp = sp;
    ts.node = node;
    ts.run = run;
}

// One traversal from the root to the end for the lanes with `active`.
template <int LOOP_SHARE = 0, class WS, typename LeafFn, typename LimitFn>
RT_DEV void bvh_traverse(const DBvh &bv, const BvhRay &br, bool active, const WS &ws, LeafFn leaf, LimitFn t_limit,
                         BvhCounters &cnt) {
    TravState ts;
    trav_begin(ws, ts, active);
    bvh_advance<false, LOOP_SHARE>(bv, br, ws, ts, 65, leaf, t_limit, cnt);
}

// The triangle record of a leaf slot (3 of the 4 float4 of the slot).  Its index is the first word of the
// fourth float4; callers fetch it with leaf_tri_id only for triangles that passed their test, so the common
// case costs three vector loads instead of four.
template <class WS>
RT_DEV int leaf_tri_id(const DBvh &bv, const WS &ws, int slot) {
    if (WS::RES) return __float_as_int(ws.leaves[4 * slot + 3].x);
    return __float_as_int(bv.leaf_geo[4 * slot + 3].x);
}
// The index where bvh_advance has fetched it ahead (records in memory), else read now.
template <class WS>
RT_DEV int leaf_id(const DBvh &bv, const WS &ws, int slot, int id_ahead) {
    if (!WS::RES && LEAF_AHEAD) return id_ahead;
    return leaf_tri_id(bv, ws, slot);
}
template <class WS>
RT_DEV Tri load_leaf_tri(const DBvh &bv, const WS &ws, int slot) {
    v4f a, b, c;
    if (WS::RES) {
        a = ws.leaves[4 * slot + 0]; b = ws.leaves[4 * slot + 1]; c = ws.leaves[4 * slot + 2];
    } else {
        const v4f *g = (const v4f *)bv.leaf_geo;
        a = g[4 * slot + 0]; b = g[4 * slot + 1]; c = g[4 * slot + 2];
    }
    Tri t;
    t.n = mk3(a.x, a.y, a.z);
    t.p = mk3(a.w, b.x, b.y);
    t.u = mk3(b.z, b.w, c.x);
    t.v = mk3(c.y, c.z, c.w);
    return t;
}

// Running state of a closest-hit query: the fold of the header comment.
struct ClosestFold {
    int best;
    float t_min;
    int n_near;
    f4 st;       // (x, y, rd, det) of the running best, or (0, 0, -1, w_init)
};
RT_DEV void closest_fold_reset(ClosestFold &cf, float w_init) {
    cf.best = -1;
    cf.t_min = __builtin_inff();
    cf.n_near = 0;
    cf.st.x = 0.0f; cf.st.y = 0.0f; cf.st.z = -1.0f; cf.st.w = w_init;
}
// Tests the triangle of leaf slot `slot` and folds it in if it is a candidate; `current` is skipped.  (A hierarchy
// holds exactly the triangles it is searched for: all of them, or the emissive ones.)
template <class WS>
RT_DEV void closest_fold_leaf(const DBvh &bv, const WS &ws, const Ray &ray, int slot, const Tri &t, int id_ahead, int current, ClosestFold &cf) {
    const float det = -dot(ray.d, t.n);
    const f3 delta = ray.o - t.p;
    const float rd = dot(t.n, delta);
    if ((det <= 0.0f) | (rd <= 0.0f)) return;
    PlaneTest pt;
    pt.det = det; pt.rd = rd; pt.delta = delta;
    float rx, ry;
    if (!inside_test(ray, t, pt, rx, ry)) return;
    const int id = leaf_id(bv, ws, slot, id_ahead);
    if (id == current) return;
    // a candidate: rule 3 between it and the running best
    bool take;
    if (cf.best < 0) take = true;
    else if (id > cf.best) take = !(rd * cf.st.w > cf.st.z * det);
    else take = (cf.st.z * det > rd * cf.st.w);
    const float tc = rd * __builtin_amdgcn_rcpf(det);     // pruning only: 1 ulp is far inside the margin
    if (tc * BVH_NEAR < cf.t_min) cf.n_near = 1;
    else if (tc <= cf.t_min * BVH_NEAR) cf.n_near++;
    cf.t_min = __builtin_fminf(cf.t_min, tc);
    if (take) { cf.best = id; cf.st.x = rx; cf.st.y = ry; cf.st.z = rd; cf.st.w = det; }
}
constexpr int REDO_BY_WAVE_MIN = 1024;
// Rare cases, redone by the index-order loop: three or more candidates within the margin of the closest
// depth, or a non-finite ray.  Wave-collective; `redo` selects the lanes.
RT_DEV void closest_fold_redo(const DScene &sc, const Ray &ray, bool redo, int begin_id, int end_id, int current, float w_init,
                              ClosestFold &cf) {
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
        f4 s2;
        s2.x = 0.0f; s2.y = 0.0f; s2.z = -1.0f; s2.w = w_init;
        int b2 = -1;
        // few triangles: every lane walks them with its own ray; many: ray by ray, the lanes share the triangles
        if (end_id - begin_id < REDO_BY_WAVE_MIN) b2 = closest_hit_range(sc.geo, begin_id, end_id, ray, redo, current, s2, -1);
        else closest_hit_range_by_wave(sc.geo, begin_id, end_id, ray, redo, current, s2, b2);
        if (redo) { cf.best = b2; cf.st = s2; }
    }
}

// Closest hit with the reference's fold semantics (see the header comment).  `begin_id`..`end_id`
// restricts the triangles considered (the glossy probe folds over the light triangles only),
// `current` is skipped.  Returns the triangle id or -1; isec = (x, y, rd, det) of the winner, or
// (0, 0, -1, w_init) when nothing was hit.
template <int LOOP_SHARE = 0, class WS>
RT_DEV int closest_hit_bvh(const DScene &sc, const DBvh &bv, const Ray &ray, bool active, int begin_id, int end_id,
                           int current, float w_init, const WS &ws, f4 &isec, BvhCounters &cnt) {
    const BvhRay br = bvh_ray_setup<WS::RES>(ray, bv, ws);
    ClosestFold cf;
    closest_fold_reset(cf, w_init);
    auto leaf = [&](int slot, const Tri &t, int id_ahead) -> bool { closest_fold_leaf(bv, ws, ray, slot, t, id_ahead, current, cf); return true; };
    auto limit = [&]() -> float { return cf.t_min * BVH_NEAR; };
    bvh_traverse<LOOP_SHARE>(bv, br, active && br.finite, ws, leaf, limit, cnt);
    closest_fold_redo(sc, ray, active && (!br.finite || cf.n_near > 2), begin_id, end_id, current, w_init, cf);
    isec = cf.st;
    return cf.best;
}

// Any-hit search in which the lanes of a wave share the work.  An any-hit query is an OR over the subtrees on a
// ray's stack, so a lane with nothing to do (no query of its own, or done early: these searches end at very
// different times) can take the top stack entry of a lane that is still busy, together with that lane's ray, and
// search that subtree for it.  A hit is reported through a flag of the query's owner in LDS; every helper of
// the same owner sees it and stops.  `scratch` points to 128 bytes of LDS of this wave.
// test(ray, extra, slot) is the triangle test (it includes the exclusion of the triangle the ray starts on).
#ifndef RT_SHARE_MIN_IDLE
#define RT_SHARE_MIN_IDLE 16
#endif
constexpr int SHARE_MIN_IDLE = RT_SHARE_MIN_IDLE;
// NODE_BUDGET > 0: at most that many node steps per lane between two looks for idle lanes.  Long traversals gain
// (1 M triangles: +8 % with 4), short ones lose the bookkeeping (424 triangles: -2.5 %), so the host picks by scene size.
#ifndef RT_SHARE_NODE_BUDGET
#define RT_SHARE_NODE_BUDGET 4
#endif
constexpr int SHARE_NODE_BUDGET = RT_SHARE_NODE_BUDGET;
#ifndef RT_ANYHIT_LOOP_SHARE
#define RT_ANYHIT_LOOP_SHARE 0
#endif
constexpr int SHARE_BUDGET_MIN_TRIANGLES = 4096;
typedef volatile __attribute__((address_space(3))) unsigned char *lds_u8_ptr;

RT_DEV float lane_read(float v, int lane4) { return __int_as_float(__builtin_amdgcn_ds_bpermute(lane4, __float_as_int(v))); }

template <int NODE_BUDGET, class WS, typename TestFn>
RT_DEV bool anyhit_shared(const DBvh &bv, const WS &ws, lds_u8_ptr scratch, Ray ray, float lim, f3 extra, bool active,
                          TestFn test, BvhCounters &cnt) {
    const int lane = (int)(threadIdx.x & 63u);
    lds_u8_ptr flags = scratch, table = scratch + 64;
    flags[lane] = 0;
    int owner = lane;
    BvhRay br = bvh_ray_setup<WS::RES>(ray, bv, ws);
    TravState ts;
    trav_begin(ws, ts, active);
    auto leaf = [&](int slot, const Tri &t, int id_ahead) -> bool {
        if (test(ray, extra, slot, t, id_ahead)) { flags[owner] = 1; return false; }
        return true;
    };
    auto limit = [&]() -> float { return lim; };
    bool handed_over = false;       // wave-uniform: some lane of this wave searches for another lane's query
    for (;;) {
        // another helper of this query has found a hit (nobody has a helper before the first hand-over: no flag to read)
        if (handed_over && ts.run && flags[owner] != 0) ts.run = false;
        const unsigned long long running = __builtin_amdgcn_ballot_w64(ts.run);
        if (running == 0ull) break;
        const unsigned long long donors = __builtin_amdgcn_ballot_w64(ts.run && ts.sp > 1);
        const int n_idle = 64 - (int)__popcll(running), n_donors = (int)__popcll(donors);
        if (n_idle >= SHARE_MIN_IDLE && n_donors > 0) {
            const int n_take = n_idle < n_donors ? n_idle : n_donors;
            const unsigned long long idle = ~running;
            const int d_rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(donors >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donors, 0u));
            const int i_rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            handed_over = true;
            const bool gives = ts.run && ts.sp > 1 && d_rank < n_take;
            const bool takes = !ts.run && i_rank < n_take;
            if (gives) table[d_rank] = (unsigned char)lane;
            // the entry a donor hands over: the top of its stack
            int given = 0;
            if (gives) { ts.sp--; given = ws.stack[64 * ts.sp]; }
            const int from4 = takes ? 4 * (int)table[i_rank] : 4 * lane;
            // every lane publishes its state; a taker reads its donor's
            const int t_node = __builtin_amdgcn_ds_bpermute(from4, given);
            const int t_owner = __builtin_amdgcn_ds_bpermute(from4, owner);
            Ray r2; BvhRay b2; f3 e2;
            r2.o = mk3(lane_read(ray.o.x, from4), lane_read(ray.o.y, from4), lane_read(ray.o.z, from4));
            r2.d = mk3(lane_read(ray.d.x, from4), lane_read(ray.d.y, from4), lane_read(ray.d.z, from4));
            b2.inv = mk3(lane_read(br.inv.x, from4), lane_read(br.inv.y, from4), lane_read(br.inv.z, from4));
            b2.ca = mk3(lane_read(br.ca.x, from4), lane_read(br.ca.y, from4), lane_read(br.ca.z, from4));
            b2.cb = mk3(lane_read(br.cb.x, from4), lane_read(br.cb.y, from4), lane_read(br.cb.z, from4));
            e2 = mk3(lane_read(extra.x, from4), lane_read(extra.y, from4), lane_read(extra.z, from4));
            const float l2 = lane_read(lim, from4);
            if (takes) {
                ray = r2; br.inv = b2.inv; br.ca = b2.ca; br.cb = b2.cb; extra = e2; lim = l2;
                bvh_ray_offsets<WS::RES>(br);
                owner = t_owner;
                ts.node = t_node; ts.sp = 1; ts.run = true;
            }
        }
        bvh_advance<(NODE_BUDGET > 0), (NODE_BUDGET > 0 ? RT_ANYHIT_LOOP_SHARE : 0)>(bv, br, ws, ts, SHARE_MIN_IDLE, leaf, limit, cnt, true, NODE_BUDGET > 0 ? NODE_BUDGET : 1 << 30);
    }
    return flags[lane] != 0;
}

// sampleLight's occlusion query (raytracer.glsl:372-378): is any triangle != current hit by
// intersectTriangleShadow within max_t?
template <class WS>
RT_DEV bool shadow_any_hit_bvh(const DScene &sc, const DBvh &bv, const Ray &ray, float max_t, int current, bool active,
                               const WS &ws, BvhCounters &cnt) {
    const BvhRay br = bvh_ray_setup<WS::RES>(ray, bv, ws);
    const bool ok = br.finite && ((max_t - max_t) == 0.0f);
    bool blocked = false;
    auto leaf = [&](int slot, const Tri &t, int id_ahead) -> bool {
        if (intersect_triangle_shadow(ray, t, max_t) && leaf_id(bv, ws, slot, id_ahead) != current) { blocked = true; return false; }
        return true;
    };
    const float lim = max_t * BVH_NEAR;
    auto limit = [&]() -> float { return lim; };
    bvh_traverse(bv, br, active && ok, ws, leaf, limit, cnt);
    const bool redo = active && !ok;
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
        const bool b2 = shadow_any_hit_bf(sc, ray, max_t, current, redo);
        if (redo) blocked = b2;
    }
    return blocked;
}

// Second phase of sampleLightGlossy (raytracer.glsl:406-419): does any triangle of [lights, count)
// other than `current` pass intersectTriangle against the fixed state `isec`?
template <class WS>
RT_DEV bool closer_any_hit_bvh(const DScene &sc, const DBvh &bv, const Ray &ray, const f4 &isec, int current, bool active,
                               const WS &ws, BvhCounters &cnt) {
    const BvhRay br = bvh_ray_setup<WS::RES>(ray, bv, ws);
    const bool bounded = isec.z >= 0.0f;                 // a light was hit: only triangles not farther than it matter
    const float t_light = isec.z / isec.w;
    const bool ok = br.finite && (!bounded || (t_light - t_light) == 0.0f);
    bool blocked = false;
    const int lights = sc.lights;
    auto leaf = [&](int slot, const Tri &t, int id_ahead) -> bool {
        PlaneTest pt;
        float rx, ry;
        if (plane_test_closest(ray, t, isec, pt) && inside_test(ray, t, pt, rx, ry)) {
            const int id = leaf_id(bv, ws, slot, id_ahead);
            if (id >= lights && id != current) { blocked = true; return false; }
        }
        return true;
    };
    const float lim = bounded ? t_light * BVH_NEAR : __builtin_inff();
    auto limit = [&]() -> float { return lim; };
    bvh_traverse(bv, br, active && ok, ws, leaf, limit, cnt);
    const bool redo = active && !ok;
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
        const bool b2 = closer_any_hit_range(sc.geo, lights < sc.count ? lights : sc.count, sc.count, ray, isec, current, redo);
        if (redo) blocked = b2;
    }
    return blocked;
}

// sampleLightGlossy's search (raytracer.glsl:403-419) in its two-phase form, see glossy_search_bf.
// `lbv` is a hierarchy over the light triangles [0, lights) alone; `ws` holds the LDS copy of `bv`, `lws` that of `lbv`.
template <class WS>
RT_DEV int glossy_search_bvh(const DScene &sc, const DBvh &bv, const DBvh &lbv, const Ray &ray, int current, bool active,
                             int &light_id, f4 &isec, const WS &ws, const WS &lws, BvhCounters &cnt) {
    const int count = sc.count, lights = sc.lights < sc.count ? sc.lights : sc.count;
    if (lbv.enabled) {
        light_id = closest_hit_bvh(sc, lbv, ray, active, 0, lights, current, 0.0f, lws, isec, cnt);
    } else {
        isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 0.0f;
        light_id = closest_hit_range(sc.geo, 0, lights, ray, active, current, isec, -1);
    }
    const bool blocked = (lights < count) && closer_any_hit_bvh(sc, bv, ray, isec, current, active, ws, cnt);
    if (blocked) return 0;
    return light_id < 0 ? 1 : 2;
}

// shadow_any_hit_bvh and glossy_search_bvh with the any-hit part shared among the lanes of the wave (anyhit_shared).
template <int NODE_BUDGET, class WS>
RT_DEV bool shadow_any_hit_shared(const DScene &sc, const DBvh &bv, const Ray &ray, float max_t, int current, bool active,
                                  const WS &ws, lds_u8_ptr scratch, BvhCounters &cnt) {
    const bool ok = bvh_ray_ok(ray, bv.max_abs_coord) && ((max_t - max_t) == 0.0f);
    auto test = [&](const Ray &r, const f3 &e, int slot, const Tri &t, int id_ahead) -> bool {
        return intersect_triangle_shadow(r, t, e.x) && leaf_id(bv, ws, slot, id_ahead) != __float_as_int(e.z);
    };
    bool blocked = anyhit_shared<NODE_BUDGET>(bv, ws, scratch, ray, max_t * BVH_NEAR, mk3(max_t, 0.0f, __int_as_float(current)), active && ok,
                                 test, cnt);
    const bool redo = active && !ok;
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
        const bool b2 = shadow_any_hit_bf(sc, ray, max_t, current, redo);
        if (redo) blocked = b2;
    }
    return blocked;
}

template <int NODE_BUDGET, class WS>
RT_DEV int glossy_search_shared(const DScene &sc, const DBvh &bv, const DBvh &lbv, const Ray &ray, int current, bool active,
                                int &light_id, f4 &isec, const WS &ws, const WS &lws, lds_u8_ptr scratch,
                                BvhCounters &cnt) {
    const int count = sc.count, lights = sc.lights < sc.count ? sc.lights : sc.count;
    if (lbv.enabled) {
        light_id = closest_hit_bvh(sc, lbv, ray, active, 0, lights, current, 0.0f, lws, isec, cnt);
    } else {
        isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 0.0f;
        light_id = closest_hit_range(sc.geo, 0, lights, ray, active, current, isec, -1);
    }
    bool blocked = false;
    if (lights < count) {
        const bool bounded = isec.z >= 0.0f;                 // a light was hit: only triangles not farther than it matter
        const float t_light = isec.z / isec.w;
        const bool ok = bvh_ray_ok(ray, bv.max_abs_coord) && (!bounded || (t_light - t_light) == 0.0f);
        auto test = [&](const Ray &r, const f3 &e, int slot, const Tri &t, int id_ahead) -> bool {
            f4 st;
            st.x = 0.0f; st.y = 0.0f; st.z = e.x; st.w = e.y;
            PlaneTest pt;
            float rx, ry;
            if (!(plane_test_closest(r, t, st, pt) && inside_test(r, t, pt, rx, ry))) return false;
            const int id = leaf_id(bv, ws, slot, id_ahead);
            return id >= lights && id != __float_as_int(e.z);
        };
        blocked = anyhit_shared<NODE_BUDGET>(bv, ws, scratch, ray, bounded ? t_light * BVH_NEAR : __builtin_inff(),
                                mk3(isec.z, isec.w, __int_as_float(current)), active && ok, test, cnt);
        const bool redo = active && !ok;
        if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
            const bool b2 = closer_any_hit_range(sc.geo, lights, count, ray, isec, current, redo);
            if (redo) blocked = b2;
        }
    }
    if (blocked) return 0;
    return light_id < 0 ? 1 : 2;
}

// rt_closest_hit_batch through the hierarchy.  A fixed grid of single-wave workgroups strides over the
// batch in chunks of 64 rays, so that each wave adds its counters to the shared totals once.
__global__ __launch_bounds__(64) void k_closest_hit_batch_bvh(DScene sc, DBvh bv, const float *__restrict__ o,
                                                              const float *__restrict__ d, unsigned long long n,
                                                              int *__restrict__ id, float *__restrict__ uvt,
                                                              DStats *__restrict__ stats) {
    __shared__ int stack_mem[BVH_STACK * 64];
    __shared__ v4f top_mem[128 * 2];              // single-wave workgroups: a small top only
    __shared__ float grid_mem[8];
    BvhScratch ws;
    bvh_stage_grid(bv, grid_mem);
    ws.grid = (lds_f32_ptr)grid_mem;
    ws.stack = stack_mem + threadIdx.x;
    ws.top = (lds_v4f_ptr)top_mem;
    ws.leaves = nullptr;
    ws.root = 0;
    ws.n_top = bvh_stage_top(bv, top_mem, 128);
    BvhCounters cnt;
    cnt.nodes = 0; cnt.tris = 0;
    const unsigned long long chunks = (n + 63ull) / 64ull;
    for (unsigned long long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const unsigned long long i = c * 64ull + threadIdx.x;
        const bool live = i < n;
        const unsigned long long j = live ? i : 0ull;
        Ray ray;
        ray.o = mk3(o[j * 3 + 0], o[j * 3 + 1], o[j * 3 + 2]);
        ray.d = mk3(d[j * 3 + 0], d[j * 3 + 1], d[j * 3 + 2]);
        f4 isec;
        const int hit = closest_hit_bvh(sc, bv, ray, live, 0, sc.count, -1, 1.0f, ws, isec, cnt);
        if (live) {
            const float inv_det = 1.0f / isec.w;
            id[i] = hit;
            uvt[i * 3 + 0] = isec.x * inv_det;
            uvt[i * 3 + 1] = isec.y * inv_det;
            uvt[i * 3 + 2] = isec.z * inv_det;
        }
    }
    const unsigned long long nn = wave_sum_u32(cnt.nodes), nt = wave_sum_u32(cnt.tris);
    if (threadIdx.x == 0 && (nn | nt)) { atomicAdd(&stats->ext_nodes, nn); atomicAdd(&stats->ext_tris, nt); }
}

}  // namespace rt
