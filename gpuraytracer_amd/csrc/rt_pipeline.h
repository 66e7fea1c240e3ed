// rt_pipeline.h - staged ("wavefront") form of the path tracer for gfx950.
//
// The reference walks a whole path in one shader invocation (raytracer.glsl:440-584).  Paths
// of one wave end at different depths and only some of them need a probe loop, so a wave that
// keeps its 64 pixels for the whole walk soon runs mostly idle lanes through the O(COUNT)
// loops.  Here a bounce is split into stages that each run as their own kernel over densely
// packed queues:
//
//   first    depth 0 in one kernel: camera ray, closest hit, shading, the probe in place (all lanes of a wave
//            want the same kind of probe there), rest of the bounce                   -> extend queue of depth 1
//   extend   closest hit (findIntersection) + everything of the bounce up to the probe loop;
//            paths that need a probe go to the shadow or the glossy queue, paths that need none
//            are finished (roulette) and go to the next extend queue
//   shadow   occlusion loop of sampleLight + rest of the bounce        -> next extend queue
//   glossy   search loop of sampleLightGlossy + rest of the bounce     -> next extend queue
//   resolve  adds the radiance of the batch's paths to the per-pixel sums in ascending sample order
//
// A queue entry is the path's whole state (a record), not an index into a state array: the survivors of a
// bounce are a sparse subset of the paths, and gathering 16-byte fields of every fifth path from planes
// indexed by path id made the stages of depth >= 1 wait on HBM (each 16-byte field costs a whole memory
// sector; measured: 223 cycles of the vector memory path per load instruction).  Records are written
// by the producer at consecutive positions of its own queue region and read back by the consumer with
// consecutive lanes on consecutive records.
//
// Record planes (structure of arrays, one float4/uint4 per record and plane):
//   path record   xa = (o.xyz, d.x)  xb = (d.yz, path.xy)  xc = (path.z, energy.xyz)
//                 xd = (seed, path id, kept flags, -)
//   probe record  pa, pb, pc as above (state after the bounce's direction change)
//                 pd = (seed, path id, current triangle | kept flags << 30, closest light of glossy pass 1)
//                 p0 = (probe_d.xyz, max_t)
//                 p1 = (lit.xyz, roughness), p2 = (albedo.xyz, -) for shadow probes
//                 p1 = (specular.xyz, w), p2 = (x, y, rd, det) of the closest light for glossy probes
//   Shadow records fill a region of the probe queue from its front, glossy records from its back.
// By path id:     energy = radiance of a path that has ended (written once, read by resolve)
//                 kd, ks = kept light results (see struct PathState), touched only when their flag is set
//                 h0     = (u, v, t, hit triangle id) of the unfused form at depth 0; at depth >= 1 h0 is
//                          indexed like the extend queue being read
// Path id = slot * paths_per_sample + tile * 64 + lane, where `slot` is the sample's position in
// the batch and (tile, lane) is the 8x8 pixel tile and the pixel inside it, so that the depth-0
// wave holds the same 64 pixels as a reference workgroup.
#pragma once

#include <vector>

#include "../../include/gpurt.h"
#include "rt_kernels.h"
#include "rt_bvh.h"

namespace rt {

#ifndef RT_WAVES_PER_CU
#define RT_WAVES_PER_CU 16
#endif
#ifndef RT_WG_WAVES
#define RT_WG_WAVES 8
#endif
constexpr int PIPE_WAVES = 256 * RT_WAVES_PER_CU;    // persistent grid, all waves resident
constexpr int WG_WAVES = RT_WG_WAVES;                // waves per workgroup: they share the LDS copy of the hierarchy's top
static_assert(PIPE_WAVES % WG_WAVES == 0, "the grid is a whole number of workgroups");

// Wave-uniform by construction; readfirstlane tells the compiler so, which moves everything derived from it
// (region bases, depth-0 tile coordinates) to the scalar unit.
RT_DEV uint32_t wave_index() {
    return blockIdx.x * (uint32_t)WG_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
RT_DEV uint32_t lane_index() { return threadIdx.x & 63u; }

// Queues without atomics.  Every pipeline kernel runs exactly PIPE_WAVES waves (in workgroups of WG_WAVES), and
// wave v owns region v of every queue: it is the only one that appends to it (keeping the fill count
// in a scalar register and storing it once at the end) and the only one that later consumes it.  A
// wave can never emit more entries than it consumed, so a region of ceil(chunks / PIPE_WAVES) * 64
// entries always suffices.  With one atomic per wave and queue on three shared counters the shade
// stage spent 87 % of its time waiting (2.2 ms for 8.3 M paths); the regions took that away.
struct PipeBuffers {
    float4 *xa[2], *xb[2], *xc[2];              // extend queues (ping-pong): path records
    uint4 *xd[2];
    float4 *pa, *pb, *pc, *p0, *p1, *p2;        // probe queue: shadow records from the front of a region, glossy from its back
    uint4 *pd;
    float4 *h0, *energy, *kd, *ks;
    uint32_t *n_ext[2], *n_shadow, *n_glossy;   // fill count of each region
    uint32_t region_cap;
    unsigned long long *wstats;                 // [PIPE_WAVES][WSTAT_SLOTS] per-wave work counters (see WSTAT_*)
    uint32_t *total;                            // scratch word for the early-stop read-back
};

enum { WSTAT_PATHS = 0, WSTAT_RAYS, WSTAT_SHADOW, WSTAT_GLOSSY, WSTAT_EXT_NODES, WSTAT_EXT_TRIS, WSTAT_PROBE_NODES,
       WSTAT_PROBE_TRIS, WSTAT_DEPTH1, WSTAT_SLOTS = 16 };

constexpr uint32_t KEPT_DIFF = 1u, KEPT_SPEC = 2u;      // kept flags
constexpr uint32_t CURRENT_MASK = 0x3FFFFFFFu;

RT_DEV void unpack_path(float4 a, float4 c, float4 e, PathState &ps) {
    ps.o = mk3(a.x, a.y, a.z);
    ps.d = mk3(a.w, c.x, c.y);
    ps.path = mk3(c.z, c.w, e.x);
    ps.energy = mk3(e.y, e.z, e.w);
}
RT_DEV void load_path_rec(const PipeBuffers &b, int q, size_t i, PathState &ps, uint32_t &id, uint32_t &kept_flags) {
    unpack_path(b.xa[q][i], b.xb[q][i], b.xc[q][i], ps);
    const uint4 d = b.xd[q][i];
    ps.seed = d.x;
    id = d.y;
    kept_flags = d.z;
}
RT_DEV void store_path_rec(const PipeBuffers &b, int q, size_t i, const PathState &ps, uint32_t id, uint32_t kept_flags) {
    b.xa[q][i] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    b.xb[q][i] = make_float4(ps.d.y, ps.d.z, ps.path.x, ps.path.y);
    b.xc[q][i] = make_float4(ps.path.z, ps.energy.x, ps.energy.y, ps.energy.z);
    b.xd[q][i] = make_uint4(ps.seed, id, kept_flags, 0u);
}
// The path part of a probe record (what finish_diffuse / finish_glossy go on with).
RT_DEV void load_probe_path(const PipeBuffers &b, size_t i, PathState &ps) { unpack_path(b.pa[i], b.pb[i], b.pc[i], ps); }
// A path that has ended leaves its radiance for the resolve stage.
RT_DEV void finish_path(const PipeBuffers &b, uint32_t id, const PathState &ps) {
    b.energy[id] = make_float4(ps.energy.x, ps.energy.y, ps.energy.z, 0.0f);
}

// Position in this wave's queue region for every lane with `pred`; `fill` is wave-uniform.
RT_DEV uint32_t region_slot(uint32_t &fill, bool pred) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(pred);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t slot = fill + rank;
    fill += (uint32_t)__popcll(mask);
    return slot;
}
// Hands a path on to the next extend queue, or ends it.  Whole waves call it; `live` masks lanes.
RT_DEV void route_next(const PipeBuffers &b, int q_out, uint32_t &fill_next, bool live, bool go_on, const PathState &ps,
                       uint32_t id, uint32_t kept_flags) {
    const bool cont = live && go_on;
    const uint32_t slot = region_slot(fill_next, cont);
    if (cont) store_path_rec(b, q_out, (size_t)wave_index() * b.region_cap + slot, ps, id, kept_flags);
    if (live && !go_on) finish_path(b, id, ps);
}

RT_DEV void wstat_add(const PipeBuffers &b, int which, unsigned long long v) {
    if (lane_index() == 0 && v) b.wstats[(size_t)wave_index() * WSTAT_SLOTS + which] += v;
}

struct RouteFill { uint32_t shadow, glossy, next; };

// Shading of one path after its closest hit is known (raytracer.glsl:452-556): adds the sky on a
// miss, otherwise runs shade_pre and hands the path to the shadow queue, the glossy queue or (no
// probe needed) through the rest of the bounce to the next extend queue.  Called by whole waves.
RT_DEV void shade_and_route(const DScene &sc, const DFrame &fr, const PipeBuffers &b, int depth, int q_in,
                            const uint32_t *__restrict__ coins, uint32_t paths_per_sample, int max_depth, bool live,
                            uint32_t id, PathState &ps, uint32_t kept_flags, int hit, float hu, float hv, float ht,
                            RouteFill &fill) {
    bool to_shadow = false, to_glossy = false, go_on = false;
    Bounce bn;
    bn.kind = PROBE_NONE;
    if (live) {
        if (hit < 0) {
            ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;             // :452-456
        } else {
            const uint32_t slot = id / paths_per_sample;
            const bool coin_diffuse = coins[slot * (uint32_t)max_depth + (uint32_t)depth] != 0u;
            shade_pre(sc, fr, ps, hit, hu, hv, ht, coin_diffuse, bn);
            if (bn.kind == PROBE_NONE) {
                // sampleLight returned false before its occlusion loop: the kept light enters the roulette
                ps.kept_diff = mk3(0.0f, 0.0f, 0.0f);
                if (kept_flags & KEPT_DIFF) { const float4 k = b.kd[id]; ps.kept_diff = mk3(k.x, k.y, k.z); }
                go_on = finish_diffuse(sc, ps, hit, false, bn.lit, bn.albedo, bn.roughness) && (depth + 1 < max_depth);
            } else {
                to_shadow = bn.kind == PROBE_SHADOW;
                to_glossy = bn.kind == PROBE_GLOSSY;
            }
        }
    }
    const size_t base = (size_t)wave_index() * b.region_cap;
    const uint32_t s_slot = region_slot(fill.shadow, to_shadow);
    const uint32_t g_slot = region_slot(fill.glossy, to_glossy);
    if (to_shadow | to_glossy) {
        const size_t i = to_shadow ? base + s_slot : base + (b.region_cap - 1u - g_slot);
        b.pa[i] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
        b.pb[i] = make_float4(ps.d.y, ps.d.z, ps.path.x, ps.path.y);
        b.pc[i] = make_float4(ps.path.z, ps.energy.x, ps.energy.y, ps.energy.z);
        b.pd[i] = make_uint4(ps.seed, id, (uint32_t)hit | (kept_flags << 30), 0xFFFFFFFFu);
        b.p0[i] = make_float4(bn.probe_d.x, bn.probe_d.y, bn.probe_d.z, bn.max_t);
        if (to_shadow) {
            b.p1[i] = make_float4(bn.lit.x, bn.lit.y, bn.lit.z, bn.roughness);
            b.p2[i] = make_float4(bn.albedo.x, bn.albedo.y, bn.albedo.z, 0.0f);
        } else {
            b.p1[i] = make_float4(bn.c.x, bn.c.y, bn.c.z, bn.w);
        }
    }
    route_next(b, q_in ^ 1, fill.next, live && !(to_shadow | to_glossy), go_on, ps, id, kept_flags);
}

RT_DEV void store_route_fill(const PipeBuffers &b, int q_in, const RouteFill &fill) {
    if (lane_index() == 0) {
        const uint32_t wave = wave_index();
        b.n_shadow[wave] = fill.shadow;
        b.n_glossy[wave] = fill.glossy;
        b.n_ext[q_in ^ 1][wave] = fill.next;
    }
}

// Camera path of queue position `qi` at depth 0 (raytracer.glsl:586-607).  Returns false for the padding
// lanes of ragged tiles.
RT_DEV bool first_path(const DFrame &fr, const uint32_t *__restrict__ samples, uint32_t paths_per_sample, uint32_t tiles_x,
                       uint32_t qi, bool live, PathState &ps) {
    const uint32_t lane = lane_index();
    const uint32_t slot = live ? qi / paths_per_sample : 0u;
    const uint32_t tile = (qi - slot * paths_per_sample) >> 6;
    const uint32_t tx = (tile % tiles_x) * 8u + (lane & 7u);
    const uint32_t ty = (tile / tiles_x) * 8u + (lane >> 3);
    ps.seed = pixel_seed(fr, tx, ty, samples[slot]);
    const Ray cam = camera_ray(fr, tx, ty, ps.seed);
    ps.o = cam.o; ps.d = cam.d;
    ps.path = mk3(1.0f, 1.0f, 1.0f);
    ps.energy = mk3(0.0f, 0.0f, 0.0f);
    return live && (tx < fr.width) && (ty < fr.height);
}

// Extend stage: the closest-hit search.  FIRST: depth 0, the wave generates the camera rays of its
// pixels itself instead of reading a queue.  FUSED: the wave goes on to shade its paths; otherwise it
// writes the hit record h0 = (u, v, t, triangle id; -1 = miss) and k_shade does the shading.  BVH:
// hierarchy traversal instead of the loop over all triangles.
template <bool FIRST, bool FUSED, bool BVH, int SHARE = 0>
__global__ __launch_bounds__(64 * WG_WAVES) void k_extend(DScene sc, DBvh bv, DFrame fr, PipeBuffers b, int depth, int q_in,
                                                          const uint32_t *__restrict__ samples,
                                                          const uint32_t *__restrict__ coins, uint32_t paths_per_sample,
                                                          uint32_t n_slots, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = FIRST ? paths_per_sample * n_slots : b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    __shared__ int stack_mem[BVH ? BVH_STACK * 64 * WG_WAVES : 64];
    __shared__ v4f top_mem[BVH ? BVH_TOP_NODES * 4 : 1];
    BvhScratch ws;
    ws.stack = stack_mem + (BVH ? (threadIdx.x >> 6) * (BVH_STACK * 64) + lane : 0);
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = BVH ? bvh_stage_top(bv, top_mem) : 0;
    __shared__ unsigned char share_mem[SHARE ? 128 * WG_WAVES : 1];
    lds_u8_ptr share = (lds_u8_ptr)share_mem + (SHARE ? 128 * (threadIdx.x >> 6) : 0);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    uint32_t n_paths = 0, n_rays = 0;
    RouteFill fill;
    fill.shadow = fill.glossy = fill.next = 0;

    // FIRST: chunks wave, wave + PIPE_WAVES, ... of the identity queue; else this wave's region
    for (uint32_t base = FIRST ? wave * 64u : 0u; base < total; base += FIRST ? (uint32_t)PIPE_WAVES * 64u : 64u) {
        const uint32_t qi = base + lane;
        bool live = qi < total;
        uint32_t id = 0, kept_flags = 0;
        PathState ps;
        if (FIRST) {
            id = qi;
            live = first_path(fr, samples, paths_per_sample, tiles_x, qi, live, ps);
            n_paths += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        } else if (FUSED) {
            load_path_rec(b, q_in, region + (live ? qi : 0u), ps, id, kept_flags);
        } else {
            const size_t i = region + (live ? qi : 0u);
            const float4 a = b.xa[q_in][i], c = b.xb[q_in][i];
            ps.o = mk3(a.x, a.y, a.z);
            ps.d = mk3(a.w, c.x, c.y);
        }
        n_rays += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        const int hit = SHARE ? closest_hit_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, ray, live, 0, sc.count, -1, 1.0f, ws, share, isec, bc)
                        : BVH ? closest_hit_bvh(sc, bv, ray, live, 0, sc.count, -1, 1.0f, ws, isec, bc)
                            : closest_hit_bf(sc, ray, isec);
        const float inv_det = 1.0f / isec.w;
        if (FUSED) {
            shade_and_route(sc, fr, b, depth, q_in, coins, paths_per_sample, max_depth, live, id, ps, kept_flags, hit,
                            isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, fill);
        } else if (live) {
            b.h0[FIRST ? (size_t)qi : region + qi] =
                make_float4(isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, __int_as_float(hit));
        }
    }
    if (FUSED) store_route_fill(b, q_in, fill);
    if (FIRST) wstat_add(b, WSTAT_PATHS, n_paths);
    wstat_add(b, WSTAT_RAYS, n_rays);
    if (BVH) {
        wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
    }
}

// Depth 0 in one kernel: camera ray, closest hit, shading, the bounce's light probe and the rest of the bounce.
// At depth 0 the 64 lanes of a wave are the pixels of one 8x8 tile of one sample, so they all draw the same
// coin (raytracer.glsl:553: light_seed depends on SAMPLE and depth only) and run the same kind of probe, from
// neighbouring surface points: the probe can run in place with full lanes, and the probe records (112 bytes
// per path written and read back) are never made.  Deeper bounces mix samples in a wave and keep the queues.
template <bool BVH, int SHARE = 0>
__global__ __launch_bounds__(64 * WG_WAVES, SHARE ? 4 : 1) void k_first(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b,
                                                         const uint32_t *__restrict__ samples,
                                                         const uint32_t *__restrict__ coins, uint32_t paths_per_sample,
                                                         uint32_t n_slots, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = paths_per_sample * n_slots;
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    __shared__ int stack_mem[BVH ? BVH_STACK * 64 * WG_WAVES : 64];
    __shared__ v4f top_mem[BVH ? BVH_TOP_NODES * 4 : 1];
    __shared__ v4f light_top_mem[BVH ? BVH_LIGHT_TOP_NODES * 4 : 1];
    BvhScratch ws;
    ws.stack = stack_mem + (BVH ? (threadIdx.x >> 6) * (BVH_STACK * 64) + lane : 0);
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = BVH ? bvh_stage_top(bv, top_mem) : 0;
    BvhScratch lws = ws;
    lws.top = (lds_v4f_ptr)light_top_mem;
    // SHARE: the last 1 KB of the light hierarchy's cache holds 128 bytes per wave for anyhit_shared
    lws.n_top = BVH ? bvh_stage_top(lbv, light_top_mem, SHARE ? BVH_LIGHT_TOP_NODES - 16 : BVH_LIGHT_TOP_NODES) : 0;
    lds_u8_ptr share = (lds_u8_ptr)&light_top_mem[(BVH_LIGHT_TOP_NODES - 16) * 4] + 128 * (threadIdx.x >> 6);
    BvhCounters bc, pc;
    bc.nodes = 0; bc.tris = 0; pc.nodes = 0; pc.tris = 0;
    uint32_t n_paths = 0, n_shadow = 0, n_glossy = 0, fill_next = 0;
    const bool deeper = 1 < max_depth;

#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS > 0
    unsigned long long clk[4] = {0, 0, 0, 0};
#define SECTION_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); clk[k] += now_ - mark_; mark_ = now_; } while (0)
#else
#define SECTION_MARK(k) do { } while (0)
#define RT_SECTION_CLOCKS 0
#endif
    for (uint32_t base = wave * 64u; base < total; base += (uint32_t)PIPE_WAVES * 64u) {
#if RT_SECTION_CLOCKS > 0
        unsigned long long mark_ = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t id = base + lane;
        PathState ps;
        const bool live = first_path(fr, samples, paths_per_sample, tiles_x, id, id < total, ps);
        n_paths += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        const int hit = BVH ? closest_hit_bvh(sc, bv, ray, live, 0, sc.count, -1, 1.0f, ws, isec, bc)
                            : closest_hit_bf(sc, ray, isec);
        SECTION_MARK(0);
        const float inv_det = 1.0f / isec.w;
        // one coin for the whole wave: `base` lies in one sample's range of path ids
        const bool coin_diffuse = coins[(base / paths_per_sample) * (uint32_t)max_depth] != 0u;
        Bounce bn;
        bn.kind = PROBE_NONE; bn.max_t = 0.0f; bn.w = 0.0f; bn.roughness = 0.0f;
        bn.c = bn.lit = bn.albedo = bn.probe_d = mk3(0.0f, 0.0f, 0.0f);
        bool shaded = false;
        if (live) {
            if (hit < 0) {
                ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;             // :452-456
            } else {
                shade_pre(sc, fr, ps, hit, isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, coin_diffuse, bn);
                shaded = true;
            }
        }
        SECTION_MARK(1);
        Ray probe; probe.o = ps.o; probe.d = bn.probe_d;
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (coin_diffuse) {
            const bool run = shaded && bn.kind == PROBE_SHADOW;
            bool blocked = true;
            if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                blocked = SHARE ? shadow_any_hit_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, probe, bn.max_t, hit, run, ws, share, pc)
                          : BVH ? shadow_any_hit_bvh(sc, bv, probe, bn.max_t, hit, run, ws, pc)
                              : shadow_any_hit_bf(sc, probe, bn.max_t, hit, run);
                n_shadow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run));
            }
            SECTION_MARK(2);
            if (shaded) {
                const bool written = run && !blocked;
                if (written) { b.kd[id] = make_float4(bn.lit.x, bn.lit.y, bn.lit.z, 0.0f); kept_flags |= KEPT_DIFF; }
                go_on = finish_diffuse(sc, ps, hit, written, bn.lit, bn.albedo, bn.roughness) && deeper;
            }
        } else {
            const bool run = shaded && bn.kind == PROBE_GLOSSY;
            int outcome = 0, light_id = -1;
            f4 lsec;
            lsec.x = lsec.y = lsec.z = 0.0f; lsec.w = 1.0f;
            if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                outcome = SHARE ? glossy_search_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, lbv, probe, hit, run, light_id, lsec, ws, lws, share, pc)
                          : BVH ? glossy_search_bvh(sc, bv, lbv, probe, hit, run, light_id, lsec, ws, lws, pc)
                              : glossy_search_bf(sc, probe, hit, run, light_id, lsec);
                n_glossy += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run));
            }
            SECTION_MARK(RT_SECTION_CLOCKS == 2 ? 3 : 2);     // 2: the glossy probe is booked on the last slot
            if (shaded) {
                go_on = finish_glossy(sc, fr, ps, hit, run ? outcome : 0, bn.probe_d, light_id, lsec, bn.c, bn.w) && deeper;
                if (run && outcome != 0) {
                    b.ks[id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
                    kept_flags |= KEPT_SPEC;
                }
            }
        }
        route_next(b, 1, fill_next, live, go_on, ps, id, kept_flags);
        SECTION_MARK(3);
    }
    if (lane == 0) {
        b.n_shadow[wave] = 0;
        b.n_glossy[wave] = 0;
        b.n_ext[1][wave] = fill_next;
    }
    wstat_add(b, WSTAT_PATHS, n_paths);
    wstat_add(b, WSTAT_RAYS, n_paths);
    wstat_add(b, WSTAT_DEPTH1, fill_next);
    wstat_add(b, WSTAT_SHADOW, n_shadow);
    wstat_add(b, WSTAT_GLOSSY, n_glossy);
#if RT_SECTION_CLOCKS > 0
    // measurement build: the four hierarchy counters carry the cycles of the four sections instead
    wstat_add(b, WSTAT_EXT_NODES, clk[0]);
    wstat_add(b, WSTAT_EXT_TRIS, clk[1]);
    wstat_add(b, WSTAT_PROBE_NODES, clk[2]);
    wstat_add(b, WSTAT_PROBE_TRIS, clk[3]);
#else
    if (BVH) {
        wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(pc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(pc.tris));
    }
#endif
#undef SECTION_MARK
}

// Depth 0 with the light probes of a workgroup pooled.  In k_first a probe runs on the lane that shaded the
// path: a third of the lanes has no probe (sky, emissive hits), and any-hit searches end at very different
// times, so the probe loops run with 0.2-0.45 of their lanes doing work.  Here the 8 waves of a workgroup
// put their probe rays (32 bytes each; the path state stays in the owner's registers) into a pool, meet at a
// barrier, and then every wave draws rays from the pool: 64 at first, and as soon as POOL_REFILL_MIN lanes
// have finished, that many new ones (bvh_advance keeps the others' traversal state).  A finished lane leaves a
// one-word result in the pool record; after a second barrier the owners pick theirs up and go on with the bounce.
// The pool lives in this workgroup's part of the probe-record arrays (unused at depth 0 otherwise); it is
// written and read within microseconds by one CU, i.e. it lives in L2.  Two halves alternate from round to round, so
// that a wave that is already appending for the next round cannot overwrite results a slower wave has yet to read.
// A probe's result is a pure function of its ray, so the images are bit-identical to k_first's.
// Measured on MI355X (option "pool", off by default): the probe loops' lane utilisation rises from 0.43 to 0.58
// on a 424-triangle scene, but the kernel takes 9-20 % longer: a round's pool (at most 512 rays for 512 lanes)
// is drawn empty by the first packets, so there is nothing to refill from, and the two barriers per round make
// every wave wait for the workgroup's slowest tile twice.
constexpr int POOL_REFILL_MIN = 24;
constexpr uint32_t POOL_MIN_REGION = 128;      // two halves of 8 * region_cap / 2 >= 512 records

RT_DEV bool ray_is_finite(const Ray &ray) {
    const float s = ((ray.o.x + ray.o.y) + ray.o.z) + ((ray.d.x + ray.d.y) + ray.d.z);
    return (s - s) == 0.0f;
}

// One pass over a pool: `n_total` records at pool positions slot_of(0 .. n_total-1); `cursor` (LDS) hands them out.
// load(slot) sets up the lane's probe; leaf/limit as for bvh_advance; store(slot) leaves the result.
template <typename LoadFn, typename LeafFn, typename LimitFn, typename StoreFn>
RT_DEV void pool_consume(const DBvh &bv, const BvhScratch &ws, uint32_t *cursor, uint32_t n_total, BvhRay &br, LoadFn load,
                         LeafFn leaf, LimitFn limit, StoreFn store, BvhCounters &cnt) {
    bool busy = false, exhausted = false;
    uint32_t slot = 0;
    TravState ts;
    ts.node = 0; ts.sp = 0; ts.run = false;
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!exhausted && n_idle >= (uint32_t)POOL_REFILL_MIN) {
            uint32_t start = 0;
            if (lane_index() == 0) start = atomicAdd(cursor, n_idle);
            start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
            const uint32_t avail = start < n_total ? n_total - start : 0u;
            const uint32_t take = n_idle < avail ? n_idle : avail;
            exhausted = take < n_idle;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!busy && rank < take) {
                slot = start + rank;
                load(slot);
                busy = true;
                ts.node = 0; ts.sp = 0; ts.run = true;
            }
        }
        if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
        bvh_advance(bv, br, ws, ts, exhausted ? 65 : POOL_REFILL_MIN, leaf, limit, cnt);
        if (busy && !ts.run) store(slot);
        busy = busy && ts.run;
    }
}

__global__ __launch_bounds__(64 * WG_WAVES, 4) void k_first_pool(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b,
                                                              const uint32_t *__restrict__ samples,
                                                              const uint32_t *__restrict__ coins, uint32_t paths_per_sample,
                                                              uint32_t n_slots, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = paths_per_sample * n_slots;
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    __shared__ int stack_mem[BVH_STACK * 64 * WG_WAVES];
    __shared__ v4f top_mem[BVH_TOP_NODES * 4];
    __shared__ v4f light_top_mem[BVH_LIGHT_TOP_NODES * 4];
    // per half of the pool: shadow records, glossy records, shadow cursor, glossy cursor.  The 80 KB of the arrays
    // above are exactly half of a CU's LDS, so these 32 bytes take the place of the light hierarchy's last cached node.
    uint32_t (*pool_ctl)[4] = (uint32_t (*)[4])&light_top_mem[(BVH_LIGHT_TOP_NODES - 1) * 4];
    if (threadIdx.x < 8) (&pool_ctl[0][0])[threadIdx.x] = 0u;
    BvhScratch ws;
    ws.stack = stack_mem + (threadIdx.x >> 6) * (BVH_STACK * 64) + lane;
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = bvh_stage_top(bv, top_mem);
    BvhScratch lws = ws;
    lws.top = (lds_v4f_ptr)light_top_mem;
    lws.n_top = bvh_stage_top(lbv, light_top_mem, BVH_LIGHT_TOP_NODES - 1);
    BvhCounters bc, pc;
    bc.nodes = 0; bc.tris = 0; pc.nodes = 0; pc.tris = 0;
    uint32_t n_paths = 0, n_shadow = 0, n_glossy = 0, fill_next = 0;
    const bool deeper = 1 < max_depth;
    const int count = sc.count, lights = sc.lights < sc.count ? sc.lights : sc.count;
    const uint32_t half = (uint32_t)WG_WAVES * b.region_cap / 2u;
    const size_t pool0 = (size_t)blockIdx.x * WG_WAVES * b.region_cap;

    uint32_t parity = 0;
    for (uint32_t round = 0; (unsigned long long)round * PIPE_WAVES * 64ull < total; round++, parity ^= 1u) {
        const unsigned long long base64 = ((unsigned long long)round * PIPE_WAVES + wave) * 64ull;
        const bool in_range = base64 < total;            // wave-uniform; `total` is a multiple of 64
        const uint32_t base = in_range ? (uint32_t)base64 : 0u;
        const uint32_t id = base + lane;
        // ---- camera ray, closest hit, shading up to the probe
        PathState ps;
        const bool live = first_path(fr, samples, paths_per_sample, tiles_x, id, in_range, ps);
        n_paths += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        const int hit = closest_hit_bvh(sc, bv, ray, live, 0, count, -1, 1.0f, ws, isec, bc);
        const float inv_det = 1.0f / isec.w;
        const bool coin_diffuse = coins[(base / paths_per_sample) * (uint32_t)max_depth] != 0u;
        Bounce bn;
        bn.kind = PROBE_NONE; bn.max_t = 0.0f; bn.w = 0.0f; bn.roughness = 0.0f;
        bn.c = bn.lit = bn.albedo = bn.probe_d = mk3(0.0f, 0.0f, 0.0f);
        bool shaded = false;
        if (live) {
            if (hit < 0) {
                ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;             // :452-456
            } else {
                shade_pre(sc, fr, ps, hit, isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, coin_diffuse, bn);
                shaded = true;
            }
        }
        Ray probe; probe.o = ps.o; probe.d = bn.probe_d;
        const bool finite = ray_is_finite(probe);
        const bool run_s = shaded && coin_diffuse && bn.kind == PROBE_SHADOW;
        const bool run_g = shaded && !coin_diffuse && bn.kind == PROBE_GLOSSY;
        n_shadow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_s));
        n_glossy += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_g));
        bool blocked = false;          // result of the any-hit search of either kind
        // shadow probes the hierarchy cannot take (non-finite ray or range) go through the index-order loop here
        const bool pool_s = run_s && finite && ((bn.max_t - bn.max_t) == 0.0f);
        if (__builtin_amdgcn_ballot_w64(run_s && !pool_s) != 0ull) {
            const bool b2 = shadow_any_hit_bf(sc, probe, bn.max_t, hit, run_s && !pool_s);
            if (run_s && !pool_s) blocked = b2;
        }
        // glossy probes: the closest emissive triangle in place (a small hierarchy), the any-hit search pooled
        int light_id = -1;
        f4 lsec;
        lsec.x = lsec.y = 0.0f; lsec.z = -1.0f; lsec.w = 0.0f;
        bool pool_g = false;
        if (__builtin_amdgcn_ballot_w64(run_g) != 0ull) {
            if (lbv.enabled) {
                light_id = closest_hit_bvh(sc, lbv, probe, run_g, 0, lights, hit, 0.0f, lws, lsec, pc);
            } else {
                light_id = closest_hit_range(sc.geo, 0, lights, probe, run_g, hit, lsec, -1);
            }
            const bool bounded = lsec.z >= 0.0f;
            const float t_light = lsec.z / lsec.w;
            const bool ok = finite && (!bounded || (t_light - t_light) == 0.0f);
            const bool others = lights < count;
            pool_g = run_g && ok && others;
            const bool redo = run_g && !ok && others;
            if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
                const bool b2 = closer_any_hit_range(sc.geo, lights, count, probe, lsec, hit, redo);
                if (redo) blocked = b2;
            }
        }
        // ---- append to the pool: shadow records from the front of this round's half, glossy records from its back
        const size_t pool = pool0 + (size_t)parity * half;
        uint32_t my_slot = 0;
        {
            const unsigned long long ms = __builtin_amdgcn_ballot_w64(pool_s), mg = __builtin_amdgcn_ballot_w64(pool_g);
            uint32_t start_s = 0, start_g = 0;
            if (lane == 0) {
                if (ms) start_s = atomicAdd(&pool_ctl[parity][0], (uint32_t)__popcll(ms));
                if (mg) start_g = atomicAdd(&pool_ctl[parity][1], (uint32_t)__popcll(mg));
            }
            start_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)start_s);
            start_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)start_g);
            if (pool_s) {
                my_slot = start_s + __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
                b.p0[pool + my_slot] = make_float4(probe.o.x, probe.o.y, probe.o.z, bn.max_t);
                b.p1[pool + my_slot] = make_float4(probe.d.x, probe.d.y, probe.d.z, __int_as_float(hit));
            }
            if (pool_g) {
                my_slot = half - 1u - (start_g + __builtin_amdgcn_mbcnt_hi((uint32_t)(mg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mg, 0u)));
                b.p0[pool + my_slot] = make_float4(probe.o.x, probe.o.y, probe.o.z, lsec.z);
                b.p1[pool + my_slot] = make_float4(probe.d.x, probe.d.y, probe.d.z, __int_as_float(hit));
                b.pd[pool + my_slot] = make_uint4(0u, __float_as_uint(lsec.w), 0u, 0u);
            }
        }
        __syncthreads();
        // the other half's counters were last read before the barrier every wave has just passed
        if (threadIdx.x < 4) pool_ctl[parity ^ 1u][threadIdx.x] = 0u;
        const uint32_t pool_ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool_ctl[parity][0]);
        const uint32_t pool_ng = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool_ctl[parity][1]);
        // ---- every wave draws from the pool
        {
            Ray pr;
            pr.o = pr.d = mk3(0.0f, 0.0f, 0.0f);
            float pmax = 0.0f, lim = 0.0f;
            int pcur = -1;
            bool pblocked = false;
            f4 psec;
            psec.x = psec.y = 0.0f; psec.z = -1.0f; psec.w = 0.0f;
            BvhRay br = bvh_ray_setup(pr, bv.max_abs_coord);
            if (pool_ns > 0u) {
                auto load = [&](uint32_t slot) {
                    const float4 a = b.p0[pool + slot], d = b.p1[pool + slot];
                    pr.o = mk3(a.x, a.y, a.z); pr.d = mk3(d.x, d.y, d.z);
                    pmax = a.w; pcur = __float_as_int(d.w);
                    lim = pmax * BVH_NEAR;
                    br = bvh_ray_setup(pr, bv.max_abs_coord);
                    pblocked = false;
                };
                auto leaf = [&](int s) -> bool {
                    const Tri t = load_leaf_tri(bv, s);
                    if (intersect_triangle_shadow(pr, t, pmax) && leaf_tri_id(bv, s) != pcur) { pblocked = true; return false; }
                    return true;
                };
                auto limit = [&]() -> float { return lim; };
                auto store = [&](uint32_t slot) { ((uint32_t *)&b.pd[pool + slot])[0] = pblocked ? 1u : 0u; };
                pool_consume(bv, ws, &pool_ctl[parity][2], pool_ns, br, load, leaf, limit, store, pc);
            }
            if (pool_ng > 0u) {
                auto load = [&](uint32_t k) {
                    const size_t i = pool + (half - 1u - k);
                    const float4 a = b.p0[i], d = b.p1[i];
                    pr.o = mk3(a.x, a.y, a.z); pr.d = mk3(d.x, d.y, d.z);
                    pcur = __float_as_int(d.w);
                    psec.z = a.w; psec.w = __uint_as_float(b.pd[i].y);
                    const bool bounded = psec.z >= 0.0f;          // a light was hit: only triangles not farther than it matter
                    lim = bounded ? (psec.z / psec.w) * BVH_NEAR : __builtin_inff();
                    br = bvh_ray_setup(pr, bv.max_abs_coord);
                    pblocked = false;
                };
                auto leaf = [&](int s) -> bool {
                    const Tri t = load_leaf_tri(bv, s);
                    PlaneTest pt;
                    float rx, ry;
                    if (plane_test_closest(pr, t, psec, pt) && inside_test(pr, t, pt, rx, ry)) {
                        const int tid = leaf_tri_id(bv, s);
                        if (tid >= lights && tid != pcur) { pblocked = true; return false; }
                    }
                    return true;
                };
                auto limit = [&]() -> float { return lim; };
                auto store = [&](uint32_t k) { ((uint32_t *)&b.pd[pool + (half - 1u - k)])[0] = pblocked ? 1u : 0u; };
                pool_consume(bv, ws, &pool_ctl[parity][3], pool_ng, br, load, leaf, limit, store, pc);
            }
        }
        __syncthreads();
        // ---- the owners pick up their results and finish the bounce
        if (pool_s | pool_g) blocked = b.pd[pool + my_slot].x != 0u;
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (shaded) {
            if (coin_diffuse) {
                const bool written = run_s && !blocked;
                if (written) { b.kd[id] = make_float4(bn.lit.x, bn.lit.y, bn.lit.z, 0.0f); kept_flags |= KEPT_DIFF; }
                go_on = finish_diffuse(sc, ps, hit, written, bn.lit, bn.albedo, bn.roughness) && deeper;
            } else {
                const int outcome = run_g ? (blocked ? 0 : (light_id < 0 ? 1 : 2)) : 0;
                go_on = finish_glossy(sc, fr, ps, hit, outcome, bn.probe_d, light_id, lsec, bn.c, bn.w) && deeper;
                if (run_g && outcome != 0) {
                    b.ks[id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
                    kept_flags |= KEPT_SPEC;
                }
            }
        }
        route_next(b, 1, fill_next, live, go_on, ps, id, kept_flags);
    }
    if (lane == 0) {
        b.n_shadow[wave] = 0;
        b.n_glossy[wave] = 0;
        b.n_ext[1][wave] = fill_next;
    }
    wstat_add(b, WSTAT_PATHS, n_paths);
    wstat_add(b, WSTAT_RAYS, n_paths);
    wstat_add(b, WSTAT_DEPTH1, fill_next);
    wstat_add(b, WSTAT_SHADOW, n_shadow);
    wstat_add(b, WSTAT_GLOSSY, n_glossy);
    wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
    wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
    wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(pc.nodes));
    wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(pc.tris));
}

// The tail of the batch in one launch.  After a few bounces a small fraction of the paths is still under way, and
// a stage of the pipeline over so few records takes as long as its chain of dependent memory accesses, whatever
// the record count: 45-60 us per kernel plus launch gaps and the early-stop read-back, three kernels per depth
// (12 depths of that are a fifth of a RECURSION 16 batch).  Here every lane walks the rest of its path: closest
// hit, shading, probe in place, rest of the bounce, next bounce, as in the single-kernel form but starting from
// the records of the extend queue and with the hierarchy.  Lanes of a wave end at different depths; a lane whose
// path has ended takes the next record of the wave's region, so every bounce runs with a full wave until the
// region is used up.  The kept light results stay in registers.
#ifndef RT_TAIL_FIRST_CHECK
#define RT_TAIL_FIRST_CHECK 2     // the host first looks at the queue size after this depth
#endif
// ONE: a single bounce of every path with its probe in place; the survivors go to the next extend queue.
template <bool ONE, int SHARE>
__global__ __launch_bounds__(64 * WG_WAVES, 4) void k_tail(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b, int depth0,
                                                           int q_in, const uint32_t *__restrict__ coins,
                                                           uint32_t paths_per_sample, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    __shared__ int stack_mem[BVH_STACK * 64 * WG_WAVES];
    __shared__ v4f top_mem[BVH_TOP_NODES * 4];
    __shared__ v4f light_top_mem[BVH_LIGHT_TOP_NODES * 4];
    BvhScratch ws;
    ws.stack = stack_mem + (threadIdx.x >> 6) * (BVH_STACK * 64) + lane;
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = bvh_stage_top(bv, top_mem);
    BvhScratch lws = ws;
    lws.top = (lds_v4f_ptr)light_top_mem;
    // SHARE: the last 1 KB of the light hierarchy's cache holds 128 bytes per wave for anyhit_shared
    lws.n_top = bvh_stage_top(lbv, light_top_mem, SHARE ? BVH_LIGHT_TOP_NODES - 16 : BVH_LIGHT_TOP_NODES);
    lds_u8_ptr share = (lds_u8_ptr)&light_top_mem[(BVH_LIGHT_TOP_NODES - 16) * 4] + 128 * (threadIdx.x >> 6);
    BvhCounters bc, pc;
    bc.nodes = 0; bc.tris = 0; pc.nodes = 0; pc.tris = 0;
    uint32_t n_rays = 0, n_shadow = 0, n_glossy = 0, fill_next = 0;
    const int depth_end = ONE ? (depth0 + 1 < max_depth ? depth0 + 1 : max_depth) : max_depth;
    const uint32_t total = b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    // per-lane state of the path a lane is walking
    PathState ps;
    uint32_t id = 0, kept_flags = 0, coin_row = 0;
    bool alive = false;
    ps.o = ps.d = ps.path = ps.energy = ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
    ps.seed = 0;
    auto take_record = [&](uint32_t pos) {
        load_path_rec(b, q_in, region + pos, ps, id, kept_flags);
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        if (kept_flags & KEPT_DIFF) { const float4 k = b.kd[id]; ps.kept_diff = mk3(k.x, k.y, k.z); }
        if (kept_flags & KEPT_SPEC) { const float4 k = b.ks[id]; ps.kept_spec = mk3(k.x, k.y, k.z); }
        coin_row = (id / paths_per_sample) * (uint32_t)max_depth;
    };
    // one bounce of every lane with `alive`, each at its own `depth`
    auto bounce = [&](int depth) {
        n_rays += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(alive));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        const int hit = closest_hit_bvh(sc, bv, ray, alive, 0, sc.count, -1, 1.0f, ws, isec, bc);
        const float inv_det = 1.0f / isec.w;
        const bool coin_diffuse = coins[alive ? coin_row + (uint32_t)depth : 0u] != 0u;     // idle lanes: any valid entry
        Bounce bn;
        bn.kind = PROBE_NONE; bn.max_t = 0.0f; bn.w = 0.0f; bn.roughness = 0.0f;
        bn.c = bn.lit = bn.albedo = bn.probe_d = mk3(0.0f, 0.0f, 0.0f);
        bool shaded = false;
        if (alive) {
            if (hit < 0) {
                ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;             // :452-456
            } else {
                shade_pre(sc, fr, ps, hit, isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, coin_diffuse, bn);
                shaded = true;
            }
        }
        Ray probe; probe.o = ps.o; probe.d = bn.probe_d;
        const bool run_s = shaded && coin_diffuse && bn.kind == PROBE_SHADOW;
        const bool run_g = shaded && !coin_diffuse && bn.kind == PROBE_GLOSSY;
        bool blocked = true;
        if (__builtin_amdgcn_ballot_w64(run_s) != 0ull) {
            blocked = SHARE ? shadow_any_hit_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, probe, bn.max_t, hit, run_s, ws, share, pc)
                            : shadow_any_hit_bvh(sc, bv, probe, bn.max_t, hit, run_s, ws, pc);
            n_shadow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_s));
        }
        int outcome = 0, light_id = -1;
        f4 lsec;
        lsec.x = lsec.y = lsec.z = 0.0f; lsec.w = 1.0f;
        if (__builtin_amdgcn_ballot_w64(run_g) != 0ull) {
            outcome = SHARE ? glossy_search_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, lbv, probe, hit, run_g, light_id, lsec, ws, lws, share, pc)
                            : glossy_search_bvh(sc, bv, lbv, probe, hit, run_g, light_id, lsec, ws, lws, pc);
            n_glossy += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_g));
        }
        bool go_on = false;
        if (shaded) {
            if (coin_diffuse) {
                const bool written = run_s && !blocked;
                go_on = finish_diffuse(sc, ps, hit, written, bn.lit, bn.albedo, bn.roughness);
                if (ONE && written) { b.kd[id] = make_float4(bn.lit.x, bn.lit.y, bn.lit.z, 0.0f); kept_flags |= KEPT_DIFF; }
            } else {
                go_on = finish_glossy(sc, fr, ps, hit, run_g ? outcome : 0, bn.probe_d, light_id, lsec, bn.c, bn.w);
                if (ONE && run_g && outcome != 0) {
                    b.ks[id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
                    kept_flags |= KEPT_SPEC;
                }
            }
        }
        go_on = go_on && (depth + 1 < max_depth);
        if (alive && !go_on) finish_path(b, id, ps);
        alive = alive && go_on;
    };
    if (ONE) {
        for (uint32_t base = 0; base < total; base += 64u) {
            alive = base + lane < total;
            if (alive) take_record(base + lane);
            if (depth0 < depth_end) bounce(depth0);
            // whole waves: the queue position comes from a ballot
            const uint32_t slot = region_slot(fill_next, alive);
            if (alive) store_path_rec(b, q_in ^ 1, region + slot, ps, id, kept_flags);
        }
    } else {
        // Lanes whose path has ended take the next record of the region, so the wave stays full while lanes are at
        // different depths of different paths.
        uint32_t next = 0;
        int depth = depth0;
        for (;;) {
            const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
            if (next < total && idle != 0ull) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t avail = total - next, n_idle = (uint32_t)__popcll(idle);
                const uint32_t take = n_idle < avail ? n_idle : avail;
                if (!alive && rank < take) { take_record(next + rank); depth = depth0; alive = depth0 < max_depth; }
                next += take;
            }
            if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
            bounce(depth);
            depth++;
        }
    }
    if (lane == 0) {
        b.n_shadow[wave] = 0;
        b.n_glossy[wave] = 0;
        b.n_ext[q_in ^ 1][wave] = fill_next;
    }
    wstat_add(b, WSTAT_RAYS, n_rays);
    wstat_add(b, WSTAT_SHADOW, n_shadow);
    wstat_add(b, WSTAT_GLOSSY, n_glossy);
    wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
    wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
    wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(pc.nodes));
    wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(pc.tris));
}

// Shade stage of the unfused form.
template <bool FIRST>
__global__ __launch_bounds__(64 * WG_WAVES) void k_shade(DScene sc, DFrame fr, PipeBuffers b, int depth, int q_in,
                                                         const uint32_t *__restrict__ samples,
                                                         const uint32_t *__restrict__ coins, uint32_t paths_per_sample,
                                                         uint32_t n_slots, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = FIRST ? paths_per_sample * n_slots : b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    RouteFill fill;
    fill.shadow = fill.glossy = fill.next = 0;

    for (uint32_t base = FIRST ? wave * 64u : 0u; base < total; base += FIRST ? (uint32_t)PIPE_WAVES * 64u : 64u) {
        const uint32_t qi = base + lane;
        bool live = qi < total;
        uint32_t id = 0, kept_flags = 0;
        PathState ps;
        size_t hi;
        if (FIRST) {
            // the camera ray is cheap to regenerate (raytracer.glsl:594-606): no state was stored for depth 0
            id = live ? qi : 0u;
            live = first_path(fr, samples, paths_per_sample, tiles_x, id, live, ps);
            hi = id;
        } else {
            hi = region + (live ? qi : 0u);
            load_path_rec(b, q_in, hi, ps, id, kept_flags);
        }
        const float4 h = b.h0[hi];
        shade_and_route(sc, fr, b, depth, q_in, coins, paths_per_sample, max_depth, live, id, ps, kept_flags,
                        __float_as_int(h.w), h.x, h.y, h.z, fill);
    }
    store_route_fill(b, q_in, fill);
}

// What the probe stages read of a probe record before the search.
struct ProbeHead {
    Ray probe;
    float max_t;
    uint32_t seed, id, kept_flags;
    int current, light_id;
};
RT_DEV void load_probe_head(const PipeBuffers &b, size_t i, ProbeHead &h) {
    const float4 a = b.pa[i], pr = b.p0[i];
    const uint4 d = b.pd[i];
    h.probe.o = mk3(a.x, a.y, a.z);
    h.probe.d = mk3(pr.x, pr.y, pr.z);
    h.max_t = pr.w;
    h.seed = d.x;
    h.id = d.y;
    h.current = (int)(d.z & CURRENT_MASK);
    h.kept_flags = d.z >> 30;
    h.light_id = (int)d.w;
}
// Rest of a bounce after its shadow probe (raytracer.glsl:557-579).  Returns whether the path goes on.
RT_DEV bool retire_shadow(const DScene &sc, const PipeBuffers &b, size_t i, const ProbeHead &h, bool blocked, PathState &ps,
                          uint32_t &kept_flags) {
    load_probe_path(b, i, ps);
    ps.seed = h.seed;
    const float4 c = b.p1[i], al = b.p2[i];
    kept_flags = h.kept_flags;
    ps.kept_diff = mk3(0.0f, 0.0f, 0.0f);
    if (blocked) {
        if (kept_flags & KEPT_DIFF) { const float4 k = b.kd[h.id]; ps.kept_diff = mk3(k.x, k.y, k.z); }
    } else {
        b.kd[h.id] = make_float4(c.x, c.y, c.z, 0.0f);
        kept_flags |= KEPT_DIFF;
    }
    return finish_diffuse(sc, ps, h.current, !blocked, mk3(c.x, c.y, c.z), mk3(al.x, al.y, al.z), c.w);
}
// Rest of a bounce after its glossy probe (raytracer.glsl:420-437, :557-579).
RT_DEV bool retire_glossy(const DScene &sc, const DFrame &fr, const PipeBuffers &b, size_t i, const ProbeHead &h, int outcome,
                          int light_id, f4 lsec, PathState &ps, uint32_t &kept_flags) {
    load_probe_path(b, i, ps);
    ps.seed = h.seed;
    const float4 c = b.p1[i];
    kept_flags = h.kept_flags;
    ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
    if (outcome == 0 && (kept_flags & KEPT_SPEC)) { const float4 k = b.ks[h.id]; ps.kept_spec = mk3(k.x, k.y, k.z); }
    const bool go_on = finish_glossy(sc, fr, ps, h.current, outcome, h.probe.d, light_id, lsec, mk3(c.x, c.y, c.z), c.w);
    if (outcome != 0) {
        b.ks[h.id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
        kept_flags |= KEPT_SPEC;
    }
    return go_on;
}

// Shadow-probe stage: occlusion query of sampleLight (raytracer.glsl:372-378), then :557-579.
template <bool BVH, int SHARE = 0>
__global__ __launch_bounds__(64 * WG_WAVES) void k_probe_shadow(DScene sc, DBvh bv, PipeBuffers b, int depth, int q_in,
                                                                int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    __shared__ int stack_mem[BVH ? BVH_STACK * 64 * WG_WAVES : 64];
    __shared__ v4f top_mem[BVH ? BVH_TOP_NODES * 4 : 1];
    BvhScratch ws;
    ws.stack = stack_mem + (BVH ? (threadIdx.x >> 6) * (BVH_STACK * 64) + lane : 0);
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = BVH ? bvh_stage_top(bv, top_mem) : 0;
    __shared__ unsigned char share_mem[SHARE ? 128 * WG_WAVES : 1];
    lds_u8_ptr share = (lds_u8_ptr)share_mem + (SHARE ? 128 * (threadIdx.x >> 6) : 0);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    const size_t region = (size_t)wave * b.region_cap;
    const uint32_t total = b.n_shadow[wave];
    uint32_t fill_next = b.n_ext[q_in ^ 1][wave];
    const bool deeper = depth + 1 < max_depth;
    for (uint32_t base = 0; base < total; base += 64u) {
        const bool live = base + lane < total;
        const size_t i = region + (live ? base + lane : 0u);
        ProbeHead h;
        load_probe_head(b, i, h);
        const bool blocked = SHARE ? shadow_any_hit_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, h.probe, h.max_t, h.current, live, ws, share, bc)
                             : BVH ? shadow_any_hit_bvh(sc, bv, h.probe, h.max_t, h.current, live, ws, bc)
                                 : shadow_any_hit_bf(sc, h.probe, h.max_t, h.current, live);
        PathState ps;
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (live) go_on = retire_shadow(sc, b, i, h, blocked, ps, kept_flags) && deeper;
        route_next(b, q_in ^ 1, fill_next, live, go_on, ps, h.id, kept_flags);
    }
    if (lane == 0) b.n_ext[q_in ^ 1][wave] = fill_next;
    wstat_add(b, WSTAT_SHADOW, total);
    if (BVH) {
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(bc.tris));
    }
}

// Glossy-probe stage: search of sampleLightGlossy (raytracer.glsl:403-437), then :557-579.
template <bool BVH, int SHARE = 0>
__global__ __launch_bounds__(64 * WG_WAVES) void k_probe_glossy(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b,
                                                                int depth, int q_in, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    __shared__ int stack_mem[BVH ? BVH_STACK * 64 * WG_WAVES : 64];
    __shared__ v4f top_mem[BVH ? BVH_TOP_NODES * 4 : 1];
    BvhScratch ws;
    ws.stack = stack_mem + (BVH ? (threadIdx.x >> 6) * (BVH_STACK * 64) + lane : 0);
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = BVH ? bvh_stage_top(bv, top_mem) : 0;
    __shared__ v4f light_top_mem[BVH ? BVH_LIGHT_TOP_NODES * 4 : 1];
    BvhScratch lws = ws;
    lws.top = (lds_v4f_ptr)light_top_mem;
    // SHARE: the last 1 KB of the light hierarchy's cache holds 128 bytes per wave for anyhit_shared
    lws.n_top = BVH ? bvh_stage_top(lbv, light_top_mem, SHARE ? BVH_LIGHT_TOP_NODES - 16 : BVH_LIGHT_TOP_NODES) : 0;
    lds_u8_ptr share = (lds_u8_ptr)&light_top_mem[BVH ? (BVH_LIGHT_TOP_NODES - 16) * 4 : 0] + (SHARE ? 128 * (threadIdx.x >> 6) : 0);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    const size_t region = (size_t)wave * b.region_cap;
    const size_t last = region + b.region_cap - 1u;
    const uint32_t total = b.n_glossy[wave];
    uint32_t fill_next = b.n_ext[q_in ^ 1][wave];
    const bool deeper = depth + 1 < max_depth;
    for (uint32_t base = 0; base < total; base += 64u) {
        const bool live = base + lane < total;
        const size_t i = last - (live ? base + lane : 0u);
        ProbeHead h;
        load_probe_head(b, i, h);
        int light_id;
        f4 lsec;
        const int outcome = SHARE ? glossy_search_shared<SHARE == 2 ? SHARE_NODE_BUDGET : 0>(sc, bv, lbv, h.probe, h.current, live, light_id, lsec, ws, lws, share, bc)
                            : BVH ? glossy_search_bvh(sc, bv, lbv, h.probe, h.current, live, light_id, lsec, ws, lws, bc)
                                : glossy_search_bf(sc, h.probe, h.current, live, light_id, lsec);
        PathState ps;
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (live) go_on = retire_glossy(sc, fr, b, i, h, outcome, light_id, lsec, ps, kept_flags) && deeper;
        route_next(b, q_in ^ 1, fill_next, live, go_on, ps, h.id, kept_flags);
    }
    if (lane == 0) b.n_ext[q_in ^ 1][wave] = fill_next;
    wstat_add(b, WSTAT_GLOSSY, total);
    if (BVH) {
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(bc.tris));
    }
}

// ---------------------------------------------------------------------------------------------------
// Traversal stages with lane refill.  Rays that start on surfaces are incoherent: in a wave that takes 64
// queue entries and waits for the slowest, under a third of the lanes do useful work in the node loop.
// Here a wave keeps its lanes busy instead: as soon as REFILL_MIN lanes have finished their ray, their
// results are retired and the free lanes take the next records of the wave's queue region while the other
// lanes carry on where they were (bvh_advance keeps per-lane traversal state).  A path's arithmetic does
// not depend on which lane or in which order it is processed, and the order of a queue region only
// decides the order of the next region, so results stay bit-identical.
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 24
#endif
constexpr int REFILL_MIN = RT_REFILL_MIN;

// Hands queue positions to the lanes that are not busy, once at least REFILL_MIN lanes are free.  Returns
// true for the lanes that received one (in `k`).
RT_DEV bool refill_take(uint32_t total, uint32_t &next, bool busy, uint32_t &k) {
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    bool got = false;
    if (next < total && n_idle >= (uint32_t)REFILL_MIN) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const uint32_t avail = total - next;
        const uint32_t take = n_idle < avail ? n_idle : avail;
        if (!busy && rank < take) { k = next + rank; got = true; }
        next += take;
    }
    return got;
}

// All light probes of one depth in one launch: the shadow records, then the glossy records in two passes
// (closest light through the hierarchy over the emissive triangles; then the any-hit search over the
// others, raytracer.glsl:403-419 in the two-phase form of glossy_search_bf).
__global__ __launch_bounds__(64 * WG_WAVES) void k_probe_refill(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b, int depth,
                                                                int q_in, int max_depth) {
    const uint32_t lane = lane_index(), wave = wave_index();
    __shared__ int stack_mem[BVH_STACK * 64 * WG_WAVES];
    __shared__ v4f top_mem[BVH_TOP_NODES * 4];
    BvhScratch ws;
    ws.stack = stack_mem + (threadIdx.x >> 6) * (BVH_STACK * 64) + lane;
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = bvh_stage_top(bv, top_mem);
    __shared__ v4f light_top_mem[BVH_LIGHT_TOP_NODES * 4];
    BvhScratch lws = ws;
    lws.top = (lds_v4f_ptr)light_top_mem;
    lws.n_top = bvh_stage_top(lbv, light_top_mem, BVH_LIGHT_TOP_NODES);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    const size_t region = (size_t)wave * b.region_cap;
    const size_t last = region + b.region_cap - 1u;
    uint32_t fill_next = b.n_ext[q_in ^ 1][wave];
    const bool deeper = depth + 1 < max_depth;
    const int count = sc.count, lights = sc.lights < sc.count ? sc.lights : sc.count;
    ProbeHead h;
    h.probe.o = h.probe.d = mk3(0.0f, 0.0f, 0.0f);
    h.max_t = 0.0f; h.seed = 0; h.id = 0; h.kept_flags = 0; h.current = -1; h.light_id = -1;

    // ---- shadow records: occlusion query of sampleLight (raytracer.glsl:372-378), then :557-579
    {
        const uint32_t total = b.n_shadow[wave];
        uint32_t next = 0;
        size_t i = region;
        bool busy = false, ok = true, blocked = false;
        TravState ts;
        ts.node = 0; ts.sp = 0; ts.run = false;
        BvhRay br = bvh_ray_setup(h.probe, bv.max_abs_coord);
        for (;;) {
            uint32_t k = 0;
            if (refill_take(total, next, busy, k)) {
                i = region + k;
                load_probe_head(b, i, h);
                br = bvh_ray_setup(h.probe, bv.max_abs_coord);
                ok = br.finite && ((h.max_t - h.max_t) == 0.0f);
                blocked = false;
                busy = true;
                ts.node = 0; ts.sp = 0; ts.run = ok;
            }
            if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
            auto leaf = [&](int slot) -> bool {
                const Tri t = load_leaf_tri(bv, slot);
                if (intersect_triangle_shadow(h.probe, t, h.max_t) && leaf_tri_id(bv, slot) != h.current) { blocked = true; return false; }
                return true;
            };
            const float lim = h.max_t * BVH_NEAR;
            auto limit = [&]() -> float { return lim; };
            bvh_advance(bv, br, ws, ts, next < total ? REFILL_MIN : 65, leaf, limit, bc);
            const bool fin = busy && !ts.run;
            const bool redo = fin && !ok;
            if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
                const bool b2 = shadow_any_hit_bf(sc, h.probe, h.max_t, h.current, redo);
                if (redo) blocked = b2;
            }
            PathState ps;
            uint32_t kept_flags = 0;
            bool go_on = false;
            if (fin) go_on = retire_shadow(sc, b, i, h, blocked, ps, kept_flags) && deeper;
            route_next(b, q_in ^ 1, fill_next, fin, go_on, ps, h.id, kept_flags);
            busy = busy && ts.run;
        }
        wstat_add(b, WSTAT_SHADOW, total);
    }

    // ---- glossy records, pass 1: closest emissive triangle -> p2 = its (x, y, rd, det), pd.w = its index
    const uint32_t g_total = b.n_glossy[wave];
    if (lbv.enabled) {
        uint32_t next = 0;
        size_t i = last;
        bool busy = false;
        TravState ts;
        ts.node = 0; ts.sp = 0; ts.run = false;
        BvhRay br = bvh_ray_setup(h.probe, lbv.max_abs_coord);
        ClosestFold cf;
        closest_fold_reset(cf, 0.0f);
        for (;;) {
            uint32_t k = 0;
            if (refill_take(g_total, next, busy, k)) {
                i = last - k;
                load_probe_head(b, i, h);
                br = bvh_ray_setup(h.probe, lbv.max_abs_coord);
                closest_fold_reset(cf, 0.0f);
                busy = true;
                ts.node = 0; ts.sp = 0; ts.run = br.finite;
            }
            if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
            auto leaf = [&](int slot) -> bool { closest_fold_leaf(lbv, h.probe, slot, 0, lights, h.current, cf); return true; };
            auto limit = [&]() -> float { return cf.t_min * BVH_NEAR; };
            bvh_advance(lbv, br, lws, ts, next < g_total ? REFILL_MIN : 65, leaf, limit, bc);
            const bool fin = busy && !ts.run;
            closest_fold_redo(sc, h.probe, fin && (!br.finite || cf.n_near > 2), 0, lights, h.current, 0.0f, cf);
            if (fin) {
                b.p2[i] = make_float4(cf.st.x, cf.st.y, cf.st.z, cf.st.w);
                ((uint32_t *)&b.pd[i])[3] = (uint32_t)cf.best;
            }
            busy = busy && ts.run;
        }
    } else {
        for (uint32_t base = 0; base < g_total; base += 64u) {
            const bool live = base + lane < g_total;
            const size_t i = last - (live ? base + lane : 0u);
            load_probe_head(b, i, h);
            f4 isec;
            isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 0.0f;
            const int light_id = closest_hit_range(sc.geo, 0, lights, h.probe, live, h.current, isec, -1);
            if (live) {
                b.p2[i] = make_float4(isec.x, isec.y, isec.z, isec.w);
                ((uint32_t *)&b.pd[i])[3] = (uint32_t)light_id;
            }
        }
    }
    // the records of pass 1 are read back by other lanes of this wave: make them visible
    __threadfence();

    // ---- glossy records, pass 2: is any non-emissive triangle not farther than that light?  then :420-437, :557-579
    {
        uint32_t next = 0;
        size_t i = last;
        bool busy = false, ok = true, blocked = false;
        TravState ts;
        ts.node = 0; ts.sp = 0; ts.run = false;
        f4 isec;
        isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 0.0f;
        float lim = 0.0f;
        BvhRay br = bvh_ray_setup(h.probe, bv.max_abs_coord);
        for (;;) {
            uint32_t k = 0;
            if (refill_take(g_total, next, busy, k)) {
                i = last - k;
                load_probe_head(b, i, h);
                const float4 ls = b.p2[i];
                isec.x = ls.x; isec.y = ls.y; isec.z = ls.z; isec.w = ls.w;
                br = bvh_ray_setup(h.probe, bv.max_abs_coord);
                const bool bounded = isec.z >= 0.0f;       // a light was hit: only triangles not farther than it matter
                const float t_light = isec.z / isec.w;
                ok = br.finite && (!bounded || (t_light - t_light) == 0.0f);
                lim = bounded ? t_light * BVH_NEAR : __builtin_inff();
                blocked = false;
                busy = true;
                ts.node = 0; ts.sp = 0; ts.run = ok && (lights < count);
            }
            if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
            auto leaf = [&](int slot) -> bool {
                const Tri t = load_leaf_tri(bv, slot);
                PlaneTest pt;
                float rx, ry;
                if (plane_test_closest(h.probe, t, isec, pt) && inside_test(h.probe, t, pt, rx, ry)) {
                    const int tid = leaf_tri_id(bv, slot);
                    if (tid >= lights && tid != h.current) { blocked = true; return false; }
                }
                return true;
            };
            auto limit = [&]() -> float { return lim; };
            bvh_advance(bv, br, ws, ts, next < g_total ? REFILL_MIN : 65, leaf, limit, bc);
            const bool fin = busy && !ts.run;
            const bool redo = fin && !ok && (lights < count);
            if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
                const bool b2 = closer_any_hit_range(sc.geo, lights, count, h.probe, isec, h.current, redo);
                if (redo) blocked = b2;
            }
            PathState ps;
            uint32_t kept_flags = 0;
            bool go_on = false;
            if (fin) {
                const int outcome = blocked ? 0 : (h.light_id < 0 ? 1 : 2);
                go_on = retire_glossy(sc, fr, b, i, h, outcome, h.light_id, isec, ps, kept_flags) && deeper;
            }
            route_next(b, q_in ^ 1, fill_next, fin, go_on, ps, h.id, kept_flags);
            busy = busy && ts.run;
        }
        wstat_add(b, WSTAT_GLOSSY, g_total);
    }
    if (lane == 0) b.n_ext[q_in ^ 1][wave] = fill_next;
    wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(bc.nodes));
    wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(bc.tris));
}

// Extend stage for depth >= 1 with lane refill: the closest-hit search alone; it leaves the hit record
// h0 = (u, v, t, triangle id; -1 = miss) beside each path record and k_shade does the shading with full waves.
__global__ __launch_bounds__(64 * WG_WAVES) void k_extend_refill(DScene sc, DBvh bv, PipeBuffers b, int q_in) {
    const uint32_t lane = lane_index(), wave = wave_index();
    __shared__ int stack_mem[BVH_STACK * 64 * WG_WAVES];
    __shared__ v4f top_mem[BVH_TOP_NODES * 4];
    BvhScratch ws;
    ws.stack = stack_mem + (threadIdx.x >> 6) * (BVH_STACK * 64) + lane;
    ws.top = (lds_v4f_ptr)top_mem;
    ws.n_top = bvh_stage_top(bv, top_mem);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    const uint32_t total = b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    uint32_t next = 0;
    size_t i = region;
    bool busy = false;
    TravState ts;
    ts.node = 0; ts.sp = 0; ts.run = false;
    Ray ray;
    ray.o = ray.d = mk3(0.0f, 0.0f, 0.0f);
    BvhRay br = bvh_ray_setup(ray, bv.max_abs_coord);
    ClosestFold cf;
    closest_fold_reset(cf, 1.0f);
    for (;;) {
        uint32_t k = 0;
        if (refill_take(total, next, busy, k)) {
            i = region + k;
            const float4 a = b.xa[q_in][i], c = b.xb[q_in][i];
            ray.o = mk3(a.x, a.y, a.z);
            ray.d = mk3(a.w, c.x, c.y);
            br = bvh_ray_setup(ray, bv.max_abs_coord);
            closest_fold_reset(cf, 1.0f);
            busy = true;
            ts.node = 0; ts.sp = 0; ts.run = br.finite;
        }
        if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
        auto leaf = [&](int slot) -> bool { closest_fold_leaf(bv, ray, slot, 0, sc.count, -1, cf); return true; };
        auto limit = [&]() -> float { return cf.t_min * BVH_NEAR; };
        bvh_advance(bv, br, ws, ts, next < total ? REFILL_MIN : 65, leaf, limit, bc);
        const bool fin = busy && !ts.run;
        closest_fold_redo(sc, ray, fin && (!br.finite || cf.n_near > 2), 0, sc.count, -1, 1.0f, cf);
        if (fin) {
            const float inv_det = 1.0f / cf.st.w;
            b.h0[i] = make_float4(cf.st.x * inv_det, cf.st.y * inv_det, cf.st.z * inv_det, __int_as_float(cf.best));
        }
        busy = busy && ts.run;
    }
    wstat_add(b, WSTAT_RAYS, total);
    wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
    wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
}

// Sum of one array of region fill counts (for the host's early stop).
__global__ __launch_bounds__(256) void k_sum_counts(const uint32_t *__restrict__ counts, int n, uint32_t *__restrict__ out) {
    __shared__ uint32_t part[4];
    uint32_t v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += counts[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = part[0] + part[1] + part[2] + part[3];
}

// Resolve stage: sum[pixel] += radiance of the batch's samples, ascending slot order.
__global__ __launch_bounds__(256) void k_resolve(PipeBuffers b, float4 *__restrict__ sum, uint32_t width,
                                                 uint32_t height, uint32_t paths_per_sample, uint32_t n_slots) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= width * height) return;
    const uint32_t tx = pix % width, ty = pix / width;
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t local = ((ty >> 3) * tiles_x + (tx >> 3)) * 64u + (ty & 7u) * 8u + (tx & 7u);
    float4 s = sum[pix];
    for (uint32_t k = 0; k < n_slots; k++) {
        const float4 e = b.energy[(size_t)k * paths_per_sample + local];
        s.x += e.x; s.y += e.y; s.z += e.z;
    }
    sum[pix] = s;
}

// Host side of the pipeline: buffers sized for one batch, and the launch sequence.
struct Pipeline {
    PipeBuffers buf = {};
    size_t cap_paths = 0;
    uint32_t *d_coins = nullptr;
    size_t cap_coins = 0;
    // pinned staging for the coin table, two slots used in turn: the upload of a pass is queued behind the previous
    // pass's kernels, so the host must not overwrite a slot before its copy has run (the slot's event tells)
    uint32_t *h_coins[2] = {nullptr, nullptr};
    hipEvent_t coins_copied[2] = {nullptr, nullptr};
    int coin_slot = 0;
    uint32_t *h_count = nullptr;   // pinned, for the early-stop read-back
    // Profiling: events are recorded around every stage launch and turned into times lazily
    // (collect_times), so that the timed region carries no extra synchronisation.
    std::vector<hipEvent_t> events;
    std::vector<int> kinds;        // interval i = [event i, event i+1]: 0 extend, 3 extend at depth 0, 1 probes, 2 shade, -1 not counted
    size_t ev_used = 0;
    // light seeds are also needed on the host to tabulate the per-depth coin; set before run()
    std::vector<uint32_t> h_light_seeds;

    std::vector<void **> path_buffers() {
        return {(void **)&buf.xa[0], (void **)&buf.xa[1], (void **)&buf.xb[0], (void **)&buf.xb[1], (void **)&buf.xc[0],
                (void **)&buf.xc[1], (void **)&buf.xd[0], (void **)&buf.xd[1], (void **)&buf.pa, (void **)&buf.pb,
                (void **)&buf.pc, (void **)&buf.pd, (void **)&buf.p0, (void **)&buf.p1, (void **)&buf.p2, (void **)&buf.h0,
                (void **)&buf.energy, (void **)&buf.kd, (void **)&buf.ks};
    }

    void free_paths() {
        for (void **p : path_buffers()) { (void)hipFree(*p); *p = nullptr; }
        cap_paths = 0;
    }

    void release() {
        free_paths();
        (void)hipFree(buf.n_ext[0]); (void)hipFree(buf.n_ext[1]); (void)hipFree(buf.n_shadow); (void)hipFree(buf.n_glossy);
        (void)hipFree(buf.wstats); (void)hipFree(buf.total);
        (void)hipFree(d_coins);
        for (int k = 0; k < 2; k++) {
            if (h_coins[k]) (void)hipHostFree(h_coins[k]);
            if (coins_copied[k]) (void)hipEventDestroy(coins_copied[k]);
            h_coins[k] = nullptr;
            coins_copied[k] = nullptr;
        }
        if (h_count) (void)hipHostFree(h_count);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        buf = PipeBuffers{};
        d_coins = nullptr;
        h_count = nullptr;
        cap_coins = 0;
    }

    static uint32_t region_cap_for(size_t paths) {
        const size_t chunks = (paths + 63) / 64;
        const uint32_t cap = (uint32_t)(((chunks + PIPE_WAVES - 1) / PIPE_WAVES) * 64);
        return cap < POOL_MIN_REGION ? POOL_MIN_REGION : cap;     // k_first_pool keeps a workgroup's probe pool in 8 regions
    }

    hipError_t ensure(size_t paths, size_t coins, const char **what) {
        hipError_t e = hipSuccess;
#define PIPE_ALLOC(ptr, bytes) do { *what = #ptr; e = hipMalloc((void **)&(ptr), (bytes)); if (e != hipSuccess) return e; } while (0)
        if (!buf.wstats) {
            PIPE_ALLOC(buf.wstats, sizeof(unsigned long long) * WSTAT_SLOTS * PIPE_WAVES);
            *what = "memset wstats";
            e = hipMemset(buf.wstats, 0, sizeof(unsigned long long) * WSTAT_SLOTS * PIPE_WAVES);
            if (e != hipSuccess) return e;
            PIPE_ALLOC(buf.n_ext[0], sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_ext[1], sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_shadow, sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_glossy, sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.total, sizeof(uint32_t) * 4);
        }
        if (paths > cap_paths) {
            free_paths();
            // queue planes hold PIPE_WAVES regions; h0 is indexed by path id at depth 0 and by queue position later
            const size_t qn = (size_t)region_cap_for(paths) * PIPE_WAVES;
            for (int k = 0; k < 2; k++) {
                PIPE_ALLOC(buf.xa[k], sizeof(float4) * qn);
                PIPE_ALLOC(buf.xb[k], sizeof(float4) * qn);
                PIPE_ALLOC(buf.xc[k], sizeof(float4) * qn);
                PIPE_ALLOC(buf.xd[k], sizeof(uint4) * qn);
            }
            PIPE_ALLOC(buf.pa, sizeof(float4) * qn);
            PIPE_ALLOC(buf.pb, sizeof(float4) * qn);
            PIPE_ALLOC(buf.pc, sizeof(float4) * qn);
            PIPE_ALLOC(buf.pd, sizeof(uint4) * qn);
            PIPE_ALLOC(buf.p0, sizeof(float4) * qn);
            PIPE_ALLOC(buf.p1, sizeof(float4) * qn);
            PIPE_ALLOC(buf.p2, sizeof(float4) * qn);
            PIPE_ALLOC(buf.h0, sizeof(float4) * qn);
            PIPE_ALLOC(buf.energy, sizeof(float4) * paths);
            PIPE_ALLOC(buf.kd, sizeof(float4) * paths);
            PIPE_ALLOC(buf.ks, sizeof(float4) * paths);
            cap_paths = paths;
        }
        if (coins > cap_coins) {
            (void)hipFree(d_coins);
            d_coins = nullptr;
            cap_coins = 0;
            PIPE_ALLOC(d_coins, sizeof(uint32_t) * coins);
            for (int k = 0; k < 2; k++) {
                if (h_coins[k]) (void)hipHostFree(h_coins[k]);      // hipFree above has drained the device
                h_coins[k] = nullptr;
                *what = "h_coins";
                e = hipHostMalloc((void **)&h_coins[k], sizeof(uint32_t) * coins);
                if (e != hipSuccess) return e;
                if (!coins_copied[k]) {
                    *what = "coins_copied";
                    e = hipEventCreateWithFlags(&coins_copied[k], hipEventDisableTiming);
                    if (e != hipSuccess) return e;
                }
            }
            cap_coins = coins;
        }
        if (!h_count) {
            *what = "h_count";
            e = hipHostMalloc((void **)&h_count, sizeof(uint32_t) * 4);
            if (e != hipSuccess) return e;
        }
#undef PIPE_ALLOC
        return hipSuccess;
    }

    // Adds the time of every recorded interval to host_stats and recycles the events.
    hipError_t collect_times(rt_stats *host_stats) {
        hipError_t e = hipSuccess;
        if (ev_used >= 2) {
            e = hipEventSynchronize(events[ev_used - 1]);
            for (size_t i = 0; e == hipSuccess && i + 1 < ev_used; i++) {
                if (kinds[i] < 0) continue;
                float ms = 0.0f;
                e = hipEventElapsedTime(&ms, events[i], events[i + 1]);
                if (e != hipSuccess) break;
                if (kinds[i] == 0 || kinds[i] == 3) {
                    host_stats->extend_ms += ms;
                    host_stats->extend_launches++;
                    if (kinds[i] == 3) { host_stats->extend0_ms += ms; host_stats->extend0_launches++; }
                } else if (kinds[i] == 2) {
                    host_stats->shade_ms += ms;
                } else {
                    host_stats->probe_ms += ms;
                    host_stats->probe_launches += 2;
                }
            }
        }
        ev_used = 0;
        kinds.clear();
        return e;
    }

    // Sums the per-wave work counters into `out` (WSTAT_SLOTS values, WSTAT_* order); synchronous.
    hipError_t read_wstats(unsigned long long out[WSTAT_SLOTS]) {
        for (int k = 0; k < WSTAT_SLOTS; k++) out[k] = 0;
        if (!buf.wstats) return hipSuccess;
        std::vector<unsigned long long> h((size_t)WSTAT_SLOTS * PIPE_WAVES);
        hipError_t e = hipMemcpy(h.data(), buf.wstats, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        for (size_t i = 0; i < h.size(); i++) out[i % WSTAT_SLOTS] += h[i];
        return hipSuccess;
    }
    hipError_t reset_wstats(hipStream_t stream) {
        if (!buf.wstats) return hipSuccess;
        return hipMemsetAsync(buf.wstats, 0, sizeof(unsigned long long) * WSTAT_SLOTS * PIPE_WAVES, stream);
    }

    // Renders the `n` samples listed in d_samples and adds them to `sum`.
    hipError_t run(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum,
                   const uint32_t *d_samples, int n, hipStream_t stream, bool profiling, bool fused, bool refill,
                   bool pool, bool in_place, int tail_div, int inplace_depths, bool share, bool share_closest, rt_stats *host_stats,
                   const char **what);
};

inline hipError_t Pipeline::run(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum,
                                const uint32_t *d_samples, int n, hipStream_t stream, bool profiling, bool fused, bool refill,
                                bool pool, bool in_place, int tail_div, int inplace_depths, bool share, bool share_closest, rt_stats *host_stats,
                                const char **what) {
    const bool use_bvh = bvh.enabled != 0;
    // lane refill needs the hierarchy; it shades depth >= 1 in a kernel of its own
    const bool use_refill = use_bvh && refill && fused;
    // shared any-hit searches: with a node budget per round where traversals are long
    const int share_mode = !(use_bvh && share) ? 0 : (sc.count >= SHARE_BUDGET_MIN_TRIANGLES ? 2 : 1);
    const int max_depth = fr.recursion > 1 ? fr.recursion : 1;
    const uint32_t tiles = ((fr.width + 7u) >> 3) * ((fr.height + 7u) >> 3);
    const uint32_t pps = tiles * 64u;
    const size_t paths = (size_t)pps * (size_t)n;
    if (paths > 0xFFFFFFF0ull) { *what = "batch has more than 2^32 paths"; return hipErrorInvalidValue; }
    hipError_t e = ensure(paths, (size_t)n * max_depth, what);
    if (e != hipSuccess) return e;
    buf.region_cap = region_cap_for(paths);

    // coin[slot][depth] = unitFloat(light_seed) < 0.5 after `depth` earlier draws (raytracer.glsl:553)
#define PIPE_TRY(expr) do { *what = #expr; e = (expr); if (e != hipSuccess) return e; } while (0)
    const int cs = coin_slot;
    coin_slot ^= 1;
    PIPE_TRY(hipEventSynchronize(coins_copied[cs]));
    uint32_t *coins = h_coins[cs];
    for (int k = 0; k < n; k++) {
        uint32_t ls = h_light_seeds[k];
        for (int d = 0; d < max_depth; d++) {
            const uint32_t state = ls * 747796405u + 2891336453u;
            const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
            ls = (word >> 22u) ^ word;
            coins[(size_t)k * max_depth + d] = ((float)ls * (1.0f / 4294967296.0f) < 0.5f) ? 1u : 0u;
        }
    }
    PIPE_TRY(hipMemcpyAsync(d_coins, coins, sizeof(uint32_t) * (size_t)n * max_depth, hipMemcpyHostToDevice, stream));
    PIPE_TRY(hipEventRecord(coins_copied[cs], stream));

    if (profiling && ev_used > 8192) PIPE_TRY(collect_times(host_stats));
    auto stamp = [&](int kind_of_next_interval) -> hipError_t {
        if (!profiling) return hipSuccess;
        if (ev_used == events.size()) {
            hipEvent_t ev;
            hipError_t ee = hipEventCreate(&ev);
            if (ee != hipSuccess) return ee;
            events.push_back(ev);
        }
        kinds.push_back(kind_of_next_interval);
        return hipEventRecord(events[ev_used++], stream);
    };

    const dim3 grid(PIPE_WAVES / WG_WAVES), block(64 * WG_WAVES);
    int q_in = 0;
    for (int depth = 0; depth < max_depth; depth++) {
        PIPE_TRY(stamp(depth == 0 ? 3 : 0));
#define LAUNCH_EXTEND(F, FU, BV)                                                                                        \
    hipLaunchKernelGGL((k_extend<F, FU, BV>), grid, block, 0, stream, sc, bvh, fr, buf, depth, q_in, d_samples, d_coins, \
                       pps, (uint32_t)n, max_depth)
        if (fused && in_place && depth == 0) {
            // depth 0 runs its probes in place (k_first), so q_in = 0 is never read and the next queue is 1
            if (use_bvh && pool && buf.region_cap >= POOL_MIN_REGION)
                hipLaunchKernelGGL(k_first_pool, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, d_samples, d_coins, pps,
                                   (uint32_t)n, max_depth);
            else if (use_bvh && share_mode == 2) hipLaunchKernelGGL((k_first<true, 2>), grid, block, 0, stream, sc, bvh, light_bvh, fr,
                                                                    buf, d_samples, d_coins, pps, (uint32_t)n, max_depth);
            else if (use_bvh && share_mode == 1) hipLaunchKernelGGL((k_first<true, 1>), grid, block, 0, stream, sc, bvh, light_bvh, fr,
                                                                    buf, d_samples, d_coins, pps, (uint32_t)n, max_depth);
            else if (use_bvh) hipLaunchKernelGGL(k_first<true>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, d_samples, d_coins,
                                                 pps, (uint32_t)n, max_depth);
            else hipLaunchKernelGGL(k_first<false>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, d_samples, d_coins,
                                    pps, (uint32_t)n, max_depth);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(-1));
            q_in ^= 1;
            continue;
        }
        if (use_bvh && fused && depth > 0 && depth <= inplace_depths) {
            // this bounce with its probes in place, as at depth 0
#define LAUNCH_TAIL(ONE_, DEPTH_)                                                                                            \
    do {                                                                                                                     \
        if (share_mode == 2) hipLaunchKernelGGL((k_tail<ONE_, 2>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, DEPTH_, q_in, \
                                                d_coins, pps, max_depth);                                                    \
        else if (share_mode == 1) hipLaunchKernelGGL((k_tail<ONE_, 1>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, DEPTH_, \
                                                     q_in, d_coins, pps, max_depth);                                         \
        else hipLaunchKernelGGL((k_tail<ONE_, 0>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, DEPTH_, q_in, d_coins, pps,  \
                                max_depth);                                                                                  \
    } while (0)
            LAUNCH_TAIL(true, depth);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(-1));
            q_in ^= 1;
            continue;
        }
        if (use_refill && depth > 0) {
            hipLaunchKernelGGL(k_extend_refill, grid, block, 0, stream, sc, bvh, buf, q_in);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(2));
            hipLaunchKernelGGL(k_shade<false>, grid, block, 0, stream, sc, fr, buf, depth, q_in, d_samples, d_coins, pps,
                               (uint32_t)n, max_depth);
            PIPE_TRY(hipGetLastError());
        } else if (fused) {
            if (depth == 0) { if (use_bvh) LAUNCH_EXTEND(true, true, true); else LAUNCH_EXTEND(true, true, false); }
            else if (use_bvh && share_closest && share_mode == 2)
                hipLaunchKernelGGL((k_extend<false, true, true, 2>), grid, block, 0, stream, sc, bvh, fr, buf, depth, q_in, d_samples,
                                   d_coins, pps, (uint32_t)n, max_depth);
            else if (use_bvh && share_closest && share_mode == 1)
                hipLaunchKernelGGL((k_extend<false, true, true, 1>), grid, block, 0, stream, sc, bvh, fr, buf, depth, q_in, d_samples,
                                   d_coins, pps, (uint32_t)n, max_depth);
            else { if (use_bvh) LAUNCH_EXTEND(false, true, true); else LAUNCH_EXTEND(false, true, false); }
            PIPE_TRY(hipGetLastError());
        } else {
            if (depth == 0) { if (use_bvh) LAUNCH_EXTEND(true, false, true); else LAUNCH_EXTEND(true, false, false); }
            else { if (use_bvh) LAUNCH_EXTEND(false, false, true); else LAUNCH_EXTEND(false, false, false); }
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(2));
            if (depth == 0)
                hipLaunchKernelGGL(k_shade<true>, grid, block, 0, stream, sc, fr, buf, depth, q_in, d_samples, d_coins,
                                   pps, (uint32_t)n, max_depth);
            else
                hipLaunchKernelGGL(k_shade<false>, grid, block, 0, stream, sc, fr, buf, depth, q_in, d_samples, d_coins,
                                   pps, (uint32_t)n, max_depth);
            PIPE_TRY(hipGetLastError());
        }
#undef LAUNCH_EXTEND
        PIPE_TRY(stamp(1));
        if (use_refill && !share) {      // shared any-hit searches beat refilled ones (1 M triangles: 8.3 vs 11.1 ms)
            hipLaunchKernelGGL(k_probe_refill, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
        } else if (use_bvh && share_mode == 2) {
            hipLaunchKernelGGL((k_probe_shadow<true, 2>), grid, block, 0, stream, sc, bvh, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL((k_probe_glossy<true, 2>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
        } else if (use_bvh && share_mode == 1) {
            hipLaunchKernelGGL((k_probe_shadow<true, 1>), grid, block, 0, stream, sc, bvh, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL((k_probe_glossy<true, 1>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
        } else if (use_bvh) {
            hipLaunchKernelGGL(k_probe_shadow<true>, grid, block, 0, stream, sc, bvh, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_probe_glossy<true>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
        } else {
            hipLaunchKernelGGL(k_probe_shadow<false>, grid, block, 0, stream, sc, bvh, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_probe_glossy<false>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
        }
        PIPE_TRY(hipGetLastError());
        PIPE_TRY(stamp(-1));
        q_in ^= 1;
        // Deep paths are rare.  From depth 3 on the host looks at the size of the next extend queue: it stops when
        // the queue is empty, and hands what is left to the tail kernel once that is a small fraction of the batch.
        const bool use_tail = use_bvh && fused && tail_div > 0;
        if (depth >= (use_tail ? RT_TAIL_FIRST_CHECK : 3) && depth + 1 < max_depth) {
            hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(256), 0, stream, buf.n_ext[q_in], PIPE_WAVES, buf.total);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(hipMemcpyAsync(h_count, buf.total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            PIPE_TRY(hipStreamSynchronize(stream));
            if (h_count[0] == 0) break;
            if (use_tail && (size_t)h_count[0] * (size_t)tail_div <= paths) {
                PIPE_TRY(stamp(0));
                LAUNCH_TAIL(false, depth + 1);
                PIPE_TRY(hipGetLastError());
                PIPE_TRY(stamp(-1));
                break;
            }
        }
    }
    hipLaunchKernelGGL(k_resolve, dim3((fr.width * fr.height + 255u) / 256u), dim3(256), 0, stream, buf, sum, fr.width,
                       fr.height, pps, (uint32_t)n);
    PIPE_TRY(hipGetLastError());
#undef PIPE_TRY
    return hipSuccess;
}

}  // namespace rt
