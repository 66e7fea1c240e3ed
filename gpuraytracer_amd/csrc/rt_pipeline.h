// rt_pipeline.h - staged ("wavefront") form of the path tracer for gfx950.
//
// The reference walks a whole path in one shader invocation (raytracer.glsl:440-584).  Paths
// of one wave end at different depths and only some of them need a probe loop, so a wave that
// keeps its 64 pixels for the whole walk soon runs mostly idle lanes through the O(COUNT)
// loops.  Here a bounce is split into stages that each run as their own kernel over densely
// packed queues:
//
//   first    depth 0 in one kernel: camera ray, closest hit (over the candidate list of the wave's pixel tile, else by
//            the hierarchy), shading, the probe in place (all lanes of a wave want the same kind of probe there; chunks
//            with few probes park them as records, idle lanes of later chunks take them up), rest of the bounce
//                                                                                     -> extend queue of depth 1
//   extend   closest hit (findIntersection) + everything of the bounce up to the probe loop;
//            paths that need a probe go to the shadow or the glossy queue, paths that need none
//            are finished (roulette) and go to the next extend queue
//   shadow   occlusion loop of sampleLight + rest of the bounce        -> next extend queue
//   glossy   search loop of sampleLightGlossy + rest of the bounce     -> next extend queue
//   tail     after a few staged bounces: every lane walks the rest of a path, taking the next record of its
//            wave's queue region when its path ends
//   resolve  adds the radiance of the batch's paths to the per-pixel sums in ascending sample order
//
// The launch sequence of a pass is fixed (no decision needs a value read back from the device: a stage whose
// queue region is empty ends at once), so rt_trace only enqueues work and can be captured in a HIP graph:
//   [candidate lists of the camera rays, when the frame has changed]
//   rounds x ( first, [shadow, glossy of depth 0: the probe records the depth-0 kernel still had parked at its end],
//              staged depths x ( extend, shadow, glossy ), tail ), resolve
// The queues hold a fraction of a pass; a pass whose survivors do not fit goes on in the next round (MAX_ROUNDS), and
// the kernels of a round that has nothing to do return on their first instruction.  The extend and probe stages take
// their records in blocks of 64, part of them from pools that any wave may draw on (BlockFeed).
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
//                 pd = (seed, path id, current triangle | kept flags << 30, -)
//                 p0 = (probe_d.xyz, max_t)
//                 p1 = (lit.xyz, roughness), p2 = (albedo.xyz, -) for shadow probes
//                 p1 = (specular.xyz, w) for glossy probes
//   Shadow records fill a region of the probe queue from its front, glossy records from its back.
// By path id:     energy = radiance of a path that has ended (written once, read by resolve)
//                 kd, ks = kept light results (see struct PathState) of paths that go on, touched only when their
//                          flag is set
//                 h0     = (u, v, t, hit triangle id) beside each record of the extend queue being read (only
//                          the lane-refill form of the extend stage uses it)
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
// Waves per workgroup.  The waves of a workgroup share its LDS copy of the hierarchies, so one workgroup of 16 waves per CU
// holds that copy once where two of 8 held it twice; the 46 KB that frees take the shading planes and materials of a
// resident scene (stage_shading) or a larger node cache (stage_hierarchies).  Measured against 8: cubes +1.2 %,
// _ObiWan +1 %, 1 M triangles +0.5 %.
#ifndef RT_WG_WAVES
#define RT_WG_WAVES 16
#endif
constexpr int PIPE_WAVES = 256 * RT_WAVES_PER_CU;    // persistent grid, all waves resident
constexpr int WG_WAVES = RT_WG_WAVES;
static_assert(PIPE_WAVES % WG_WAVES == 0, "the grid is a whole number of workgroups");

// Wave-uniform by construction; readfirstlane tells the compiler so, which moves everything derived from it
// (region bases, depth-0 tile coordinates) to the scalar unit.
RT_DEV uint32_t wave_index() {
    return blockIdx.x * (uint32_t)WG_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
RT_DEV uint32_t lane_index() { return threadIdx.x & 63u; }

// Order in which the depth-0 kernel hands the 64-path chunks of a round (PIPE_WAVES chunks) to its waves.  Workgroups are
// dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has an L2 of its own: with the plain
// order chunk = wave, each L2 sees 8-tile pieces from all over the round's band of the image and so the whole of the
// hierarchy under it.  Here the waves of one XCD take one contiguous eighth of the round's chunks instead, so that an
// L2 serves one compact patch of the image.  (A wave's queue region stays its own, so the deeper stages of those
// paths run on the same XCD as well.)
#ifndef RT_XCD_ORDER
#define RT_XCD_ORDER 1
#endif
#ifndef RT_DYNAMIC_CHUNKS
#define RT_DYNAMIC_CHUNKS 1
#endif

constexpr uint32_t N_XCD = 8;
#ifndef RT_RUNS_PER_XCD
#define RT_RUNS_PER_XCD 16
#endif
#ifndef RT_CHUNK_BLOCK
#define RT_CHUNK_BLOCK 64
#endif
#ifndef RT_DYNAMIC_SHARE
#define RT_DYNAMIC_SHARE 3
#endif
#ifndef RT_DYNAMIC_SHARE_LARGE
#define RT_DYNAMIC_SHARE_LARGE 1
#endif
#ifndef RT_FIXED_GROUP
#define RT_FIXED_GROUP 8
#endif
constexpr uint32_t FIXED_GROUP = RT_FIXED_GROUP;         // consecutive chunks a wave renders in the fixed part of a pass
// ... and that it takes at once in the part handed out as waves become free: the samples of a tile that follow each other
// in a run then go to one wave and not to whichever waves ask next, so the chunk that takes up parked probes is of the
// tile that parked them and the tile's triangles stay in this CU's caches.  Measured (1 / 2 / 4 / 6 / 8 / 16): 424 triangles
// 15 130-15 310 / - / 15 350 / 15 370 / 15 370 / 15 470, 12 030 triangles 14 720 / 14 900 / 14 910 / 14 920 / 14 890 / 14 700 (large
// groups leave the waves' regions unevenly filled for the stages that follow), 1 M triangles 7350 / - / 7390 / - / 7320 / -:
// no gain there, and the depth-0 kernel fetches 1.9 times as much from beyond the L2 (the samples of a tile no longer run
// on the XCD's compute units at the same time), so scenes from LARGE_SCENE_TRIANGLES on keep single chunks.
#ifndef RT_DYNAMIC_GROUP
#define RT_DYNAMIC_GROUP 4
#endif
constexpr uint32_t DYNAMIC_GROUP = RT_DYNAMIC_GROUP;
constexpr uint32_t RUNS_PER_XCD = RT_RUNS_PER_XCD, N_RUNS = N_XCD * RUNS_PER_XCD;
constexpr uint32_t CHUNK_BLOCK = RT_CHUNK_BLOCK;        // consecutive chunks that stay in one run
// 1/share of the rounds of the depth-0 kernel is handed out as waves become free (PipeBuffers::dynamic_share): a third
// for scenes below LARGE_SCENE_TRIANGLES triangles, all of them from there on (measured on MI355X, 128 counters: cubes
// and _ObiWan 1/3 = 1/2 > 1/1 > 1/4, 1 M triangles 1/1 > 1/2 > 1/3 by 9 %: the longer the searches, the more the
// chunks differ in cost)
constexpr uint32_t DYNAMIC_SHARE = RT_DYNAMIC_SHARE, DYNAMIC_SHARE_LARGE = RT_DYNAMIC_SHARE_LARGE;
// Rounds of a pass.  The queue regions are sized for 1/rounds of the paths a wave may have to hand on (every path of
// every chunk it renders, if all of them survive).  The launch sequence of a pass is repeated `rounds` times: a wave of
// the depth-0 kernel stops taking chunks when its region cannot hold another one whole, and carries on in the next
// round, after the deeper stages have emptied the regions.  A region holds one chunk more than its share, so `rounds`
// rounds always get through the pass; usually (a fifth of the camera paths goes on in the bundled scenes) the first
// round does, raises no flag, and the kernels of the later rounds return at once.
constexpr uint32_t MAX_ROUNDS = 8;
RT_DEV uint32_t *more_flags(uint32_t *work) { return work + N_RUNS; }
// True when these launches have nothing to do: a later round of a pass whose earlier rounds got through.
RT_DEV bool round_is_idle(const uint32_t *work, uint32_t round) {
    return round > 0u && __builtin_nontemporal_load(work + N_RUNS + (round - 1u)) == 0u;
}
// The run a wave works on as its `part`-th: its own, then the others of its XCD, then those of the other XCDs.
RT_DEV uint32_t run_of_part(uint32_t home, uint32_t part) {
    const uint32_t xcd = home / RUNS_PER_XCD, sub = home % RUNS_PER_XCD;
    return ((xcd + part / RUNS_PER_XCD) % N_XCD) * RUNS_PER_XCD + (sub + part) % RUNS_PER_XCD;
}
static_assert((PIPE_WAVES / WG_WAVES) % N_XCD == 0, "workgroups divide evenly among the XCDs");
RT_DEV uint32_t first_chunk_of_wave() {
    if (!RT_XCD_ORDER) return wave_index();
    const uint32_t wg = blockIdx.x, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return (wg % N_XCD) * (uint32_t)(PIPE_WAVES / N_XCD) + (wg / N_XCD) * (uint32_t)WG_WAVES + wv;
}

// Numbering of the 8x8 pixel tiles of the target: vertical strips of TILE_STRIP tiles' width, row-major inside a
// strip (the last strip may be narrower), so that a run of consecutive tile numbers is a compact block of the image
// rather than a thin line across it (512 tiles at 16 tiles' width: 128 x 256 pixels).
#ifndef RT_TILE_STRIP
#define RT_TILE_STRIP 16
#endif
constexpr uint32_t TILE_STRIP = RT_TILE_STRIP;
RT_DEV void tile_position(uint32_t tile, uint32_t tiles_x, uint32_t tiles_y, uint32_t &tx, uint32_t &ty) {
    if (TILE_STRIP == 0) { tx = tile % tiles_x; ty = tile / tiles_x; return; }
    const uint32_t strip = tile / (TILE_STRIP * tiles_y);
    const uint32_t rest = tile - strip * (TILE_STRIP * tiles_y);
    const uint32_t left = tiles_x - strip * TILE_STRIP;
    const uint32_t w = left < TILE_STRIP ? left : TILE_STRIP;
    ty = rest / w;
    tx = strip * TILE_STRIP + (rest - ty * w);
}
RT_DEV uint32_t tile_number(uint32_t tx, uint32_t ty, uint32_t tiles_x, uint32_t tiles_y) {
    if (TILE_STRIP == 0) return ty * tiles_x + tx;
    const uint32_t strip = tx / TILE_STRIP;
    const uint32_t left = tiles_x - strip * TILE_STRIP;
    const uint32_t w = left < TILE_STRIP ? left : TILE_STRIP;
    return strip * (TILE_STRIP * tiles_y) + ty * w + (tx - strip * TILE_STRIP);
}

// Compile-time form of a traversal kernel.
//   BVH    hierarchy traversal (false: the reference's loops over all triangles)
//   SHARE  any-hit searches shared among the lanes of a wave: 0 off, 1 on, 2 on with a node-step budget per round
//   RES    the whole hierarchy, leaf triangles included, is staged into LDS (scenes of a few hundred triangles)
#ifndef RT_LDS_SHADE_KB
#define RT_LDS_SHADE_KB (RT_WG_WAVES >= 16 ? 44 : 0)    // LDS for the shading planes and materials of a resident scene
#endif
#ifndef RT_RES_STACK
#define RT_RES_STACK 16
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4                               // waves per SIMD the traversal kernels are compiled for (register budget)
#endif
constexpr int RES_STACK = RT_RES_STACK;              // traversal stack entries of the resident form (hierarchy depth < RES_STACK)
#ifndef RT_RES_KB
#define RT_RES_KB 46
#endif
constexpr int RES_BYTES = RT_RES_KB * 1024;                 // room for the resident records: 80 bytes per node, 64 per leaf slot, both hierarchies
// LDS a non-resident kernel has beyond stacks of BVH_STACK entries (hierarchies of lesser depth leave their stacks' rest to
// the node cache as well): 16 KB with two workgroups per CU, 26 KB with one (154 KB in all)
constexpr int NODE_CACHE_MIN_INTS = WG_WAVES >= 16 ? 6656 : (BVH_TOP_NODES + BVH_LIGHT_TOP_NODES) * 16;
template <bool BVH_, int SHARE_, bool RES_>
struct KernelCfg {
    static constexpr bool BVH = BVH_;
    static constexpr int SHARE = SHARE_;
    static constexpr bool RES = RES_;
    static constexpr int STACK = RES_ ? RES_STACK : BVH_STACK;
    static constexpr int NODE_BUDGET = SHARE_ == 2 ? SHARE_NODE_BUDGET : 0;
    typedef BvhScratchT<RES_> WS;
    // LDS of a workgroup: the stacks; then either cached nodes of the two hierarchies and the waves' 128-byte exchange
    // areas (carved out of one array at run time, see stage_hierarchies), or the resident scene, the exchange areas and
    // the shading planes (stage_shading).  At most 156 KB of a CU's 160 KB: one workgroup of 16 waves per CU.
    static constexpr int STACK_INTS = !BVH_ ? 64 : (RES_ ? STACK * 64 * WG_WAVES : BVH_STACK * 64 * WG_WAVES + NODE_CACHE_MIN_INTS);
    static constexpr int SCENE_V4F = (BVH_ && RES_) ? RES_BYTES / 16 : 1;
    static constexpr int SHARE_BYTES = (BVH_ && RES_) ? 128 * WG_WAVES : 1;
    static constexpr int SHADE_V4F = (BVH_ && RES_ && RT_LDS_SHADE_KB > 0) ? RT_LDS_SHADE_KB * 64 : 1;
};

// Wave priorities.  The SIMD's arbiter serves the wave with the higher priority first.  A wave that shades (camera ray, hit
// attributes, materials, light sampling, queue records) starts chains of dependent global loads, and the sooner those
// loads are issued the more of their latency the other waves' searches cover; a wave inside a search only reads LDS and
// computes.  So the searches run at the lower priority.  Measured: cubes +2.5 %, _ObiWan +0.8 %, 1 M triangles +0.5 %;
// priorities 3 / 0 and 1 / 0 give the same, the other way round (searches first) gains less.
#ifndef RT_PRIO_SHADE
#define RT_PRIO_SHADE 1
#endif
#ifndef RT_PRIO_SEARCH
#define RT_PRIO_SEARCH 0
#endif
struct SearchPriority {
    RT_DEV SearchPriority() { if (RT_PRIO_SHADE != RT_PRIO_SEARCH) __builtin_amdgcn_s_setprio(RT_PRIO_SEARCH); }
    RT_DEV ~SearchPriority() { if (RT_PRIO_SHADE != RT_PRIO_SEARCH) __builtin_amdgcn_s_setprio(RT_PRIO_SHADE); }
};

// An LDS address as a plain number in a scalar register.  Left as "start of the array + offset" the compiler keeps the
// array's start (beyond the 16 bits of an LDS instruction's offset field) apart from the rest and adds it again for every
// access: three vector additions per triangle test.
RT_DEV lds_v4f_ptr opaque_lds(const v4f *p) {
    uint32_t a = (uint32_t)(uintptr_t)(lds_v4f_ptr)p;
    asm volatile("" : "+s"(a));
    return (lds_v4f_ptr)(uintptr_t)a;
}

// Fills the LDS of a traversal kernel and the scratch descriptors of the two hierarchies; every thread of the
// workgroup calls it.  `with_light`: the kernel searches the hierarchy over the emissive triangles as well.
template <class CFG>
RT_DEV void stage_hierarchies(const DBvh &bv, const DBvh &lbv, bool with_light, int *stack_mem, v4f *scene_mem,
                              unsigned char *share_mem, typename CFG::WS &ws, typename CFG::WS &lws, lds_u8_ptr &share) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    ws.stack = stack_mem + (CFG::BVH ? wv * (CFG::STACK * 64) + lane : 0);
    ws.top = (lds_v4f_ptr)scene_mem;
    ws.leaves = nullptr;
    ws.n_top = 0;
    ws.grid = nullptr;
    ws.root = 0;
    lws = ws;
    share = (lds_u8_ptr)share_mem;
    if (RT_PRIO_SHADE != RT_PRIO_SEARCH) __builtin_amdgcn_s_setprio(RT_PRIO_SHADE);
    if (!CFG::BVH) return;
    if (CFG::RES) {
        // node records are 80 bytes = 5 float4, leaf records 4 float4
        v4f *nodes = scene_mem, *leaves = nodes + res_nodes_v4f(bv.n_nodes);
        v4f *lnodes = leaves + 4 * bv.n_slots, *lleaves = lnodes + res_nodes_v4f(lbv.n_nodes);
        bvh_stage_nodes_resident(bv, (float *)nodes);
        ws.n_top = bv.n_nodes;
        ws.root = (int)(uintptr_t)(lds_v4f_ptr)nodes;
        ws.leaves = opaque_lds(leaves);
        bvh_stage_leaves(bv, leaves);
        lws.top = (lds_v4f_ptr)lnodes;
        lws.root = (int)(uintptr_t)(lds_v4f_ptr)lnodes;
        lws.leaves = opaque_lds(lleaves);
        if (with_light) {
            bvh_stage_nodes_resident(lbv, (float *)lnodes);
            lws.n_top = lbv.n_nodes;
            bvh_stage_leaves(lbv, lleaves);
        }
        share = (lds_u8_ptr)share_mem + 128 * wv;
    } else {
        // The stacks take what the depth of the hierarchies asks for (bv.stack_entries per lane) and the rest of the
        // array caches 32-byte nodes: the first ones of the light hierarchy (all of it if it has up to 256 nodes and a
        // third of the room suffices) and, in what is left, the first ones of the scene's.  The waves' exchange areas are
        // the last 128 bytes per wave.
        const int entries = bv.stack_entries;
        ws.stack = stack_mem + wv * (entries * 64) + lane;
        v4f *free_mem = (v4f *)(stack_mem + WG_WAVES * entries * 64);
        const int room = (CFG::STACK_INTS - WG_WAVES * entries * 64) / 8 - 4 * WG_WAVES - 2;     // 32-byte node slots (less the exchange areas and the two grids)
        int light_slots = 0;
        if (with_light && lbv.enabled) {
            light_slots = lbv.n_nodes < 256 ? lbv.n_nodes : 256;
            if (light_slots > room / 3) light_slots = room / 3;
        }
        const int scene_slots = room - light_slots;
        v4f *light_top = free_mem + scene_slots * 2;
        ws.top = (lds_v4f_ptr)free_mem;
        ws.n_top = bvh_stage_top(bv, free_mem, scene_slots);
        lws = ws;
        lws.top = (lds_v4f_ptr)light_top;
        lws.n_top = 0;
        if (with_light) lws.n_top = bvh_stage_top(lbv, light_top, light_slots);
        share = (lds_u8_ptr)(free_mem + room * 2) + 128 * wv;
        float *grids = (float *)(free_mem + (room + 4 * WG_WAVES) * 2);
        bvh_stage_grid(bv, grids);
        ws.grid = (lds_f32_ptr)grids;
        lws.grid = ws.grid;
        if (with_light) { bvh_stage_grid(lbv, grids + 8); lws.grid = (lds_f32_ptr)(grids + 8); }
    }
}
#define RT_TRAVERSAL_LDS(CFG)                                              \
    __shared__ int stack_mem[CFG::STACK_INTS];                             \
    __shared__ v4f scene_mem[CFG::SCENE_V4F];                              \
    __shared__ unsigned char share_mem[CFG::SHARE_BYTES];                  \
    __shared__ v4f shade_mem[CFG::SHADE_V4F]

// Resident scenes: the four shading planes every hit reads and the material table go to LDS as well (every thread of the
// workgroup calls this).  That they fit is part of what makes a scene resident (shading_fits_lds, checked by the host),
// so that the kernels know at compile time where these reads go: LDS instructions, not generic-address loads.
constexpr int LDS_SHADE_V4F = RT_LDS_SHADE_KB * 64;
inline bool shading_fits_lds(int count, int n_mat) { return LDS_SHADE_V4F <= 1 || 4 * (long long)count + 3 * (long long)n_mat <= LDS_SHADE_V4F; }
template <class CFG>
RT_DEV void stage_shading(DScene &sc, v4f *shade_mem) {
    if (!CFG::RES || CFG::SHADE_V4F <= 1) {
        sc.shade_lds = sc.mats_lds = sc.light_geo_lds = 0u;      // (known to the compiler from here on: no LDS form of these reads in this kernel)
        return;
    }
    const int n_planes = 4 * sc.count, n_mats = 3 * sc.n_mat;
    for (int i = threadIdx.x; i < n_planes; i += blockDim.x) shade_mem[i] = ((const v4f *)sc.shade)[i];
    for (int i = threadIdx.x; i < n_mats; i += blockDim.x) {
        v4f rec = ((const v4f *)sc.mats)[i];
        // the first of a material's three words has an unused component: it carries hasTexture[m] in the LDS copy, so that a
        // shaded lane needs no read from memory to know whether its material is textured
        if (i % 3 == 0) rec.w = __int_as_float(sc.has_tex[i / 3]);
        shade_mem[n_planes + i] = rec;
    }
    // the geometry of the emissive triangles too, if there is room left: sampleLight reads one of them per bounce, per
    // lane from the second bounce on
    const int n_light_geo = 3 * (sc.lights < sc.count ? sc.lights : sc.count);
    const bool light_geo = n_light_geo > 0 && n_planes + n_mats + n_light_geo <= CFG::SHADE_V4F;
    if (light_geo)
        for (int i = threadIdx.x; i < n_light_geo; i += blockDim.x) shade_mem[n_planes + n_mats + i] = ((const v4f *)sc.geo)[i];
    __syncthreads();
    // (as plain numbers in scalar registers: left as "start of the array + offset" every read would add the start again)
    uint32_t at = (uint32_t)(uintptr_t)(lds_v4f_ptr)shade_mem;
    asm volatile("" : "+s"(at));
    __builtin_assume(at != 0u);        // (the memory form of these reads is not compiled into this kernel)
    sc.shade_lds = at;
    sc.mats_lds = at + 16u * (uint32_t)n_planes;
    if (light_geo) sc.light_geo_lds = at + 16u * (uint32_t)(n_planes + n_mats);
}

// Closest-hit searches of large scenes (the configurations with a node-step budget) leave a round's node loop when fewer
// than this share of the running lanes still descend (see bvh_advance); 0: never.
#ifndef RT_LARGE_LOOP_SHARE
#define RT_LARGE_LOOP_SHARE 4
#endif
#ifndef RT_REFILL_LOOP_SHARE
#define RT_REFILL_LOOP_SHARE 4
#endif
constexpr int LARGE_LOOP_SHARE = RT_LARGE_LOOP_SHARE, REFILL_LOOP_SHARE = RT_REFILL_LOOP_SHARE;

// The three searches of a bounce in the form the kernel's configuration selects.
template <class CFG>
RT_DEV int closest_hit_cfg(const DScene &sc, const DBvh &bv, const Ray &ray, bool active, const typename CFG::WS &ws, f4 &isec,
                           BvhCounters &cnt) {
    SearchPriority prio;
    if (CFG::BVH) return closest_hit_bvh<(CFG::NODE_BUDGET > 0 ? LARGE_LOOP_SHARE : 0)>(sc, bv, ray, active, 0, sc.count, -1, 1.0f, ws, isec, cnt);
    return closest_hit_bf(sc, ray, isec);
}
template <class CFG>
RT_DEV bool shadow_any_hit_cfg(const DScene &sc, const DBvh &bv, const Ray &probe, float max_t, int current, bool run,
                               const typename CFG::WS &ws, lds_u8_ptr share, BvhCounters &cnt) {
    SearchPriority prio;
    if (CFG::SHARE) return shadow_any_hit_shared<CFG::NODE_BUDGET>(sc, bv, probe, max_t, current, run, ws, share, cnt);
    if (CFG::BVH) return shadow_any_hit_bvh(sc, bv, probe, max_t, current, run, ws, cnt);
    return shadow_any_hit_bf(sc, probe, max_t, current, run);
}
template <class CFG>
RT_DEV int glossy_search_cfg(const DScene &sc, const DBvh &bv, const DBvh &lbv, const Ray &probe, int current, bool run,
                             int &light_id, f4 &lsec, const typename CFG::WS &ws, const typename CFG::WS &lws, lds_u8_ptr share,
                             BvhCounters &cnt) {
    SearchPriority prio;
#ifdef RT_DEBUG_PROBE
    if (CFG::BVH) {
        int a_id = -1, b_id = -1; f4 a_sec, b_sec;
        const int a = CFG::SHARE ? glossy_search_shared<CFG::NODE_BUDGET>(sc, bv, lbv, probe, current, run, a_id, a_sec, ws, lws, share, cnt)
                                 : glossy_search_bvh(sc, bv, lbv, probe, current, run, a_id, a_sec, ws, lws, cnt);
        const int b = glossy_search_bf(sc, probe, current, run, b_id, b_sec);
        if (run && (a != b || (a == 2 && a_id != b_id)))
            printf("GLOSSY share %d: o %.9g %.9g %.9g d %.9g %.9g %.9g cur %d  tree %d id %d sec %.9g %.9g %.9g %.9g  loop %d id %d sec %.9g %.9g %.9g %.9g\n",
                   (int)CFG::SHARE, probe.o.x, probe.o.y, probe.o.z, probe.d.x, probe.d.y, probe.d.z, current, a, a_id, a_sec.x, a_sec.y, a_sec.z, a_sec.w,
                   b, b_id, b_sec.x, b_sec.y, b_sec.z, b_sec.w);
        light_id = a_id; lsec = a_sec;
        return a;
    }
#endif
    if (CFG::SHARE) return glossy_search_shared<CFG::NODE_BUDGET>(sc, bv, lbv, probe, current, run, light_id, lsec, ws, lws, share, cnt);
    if (CFG::BVH) return glossy_search_bvh(sc, bv, lbv, probe, current, run, light_id, lsec, ws, lws, cnt);
    return glossy_search_bf(sc, probe, current, run, light_id, lsec);
}

// Queues without atomics.  Every pipeline kernel runs exactly PIPE_WAVES waves (in workgroups of WG_WAVES), and
// wave v owns region v of every queue: it is the only one that appends to it (keeping the fill count
// in a scalar register and storing it once at the end) and the only one that later consumes it.  A
// wave can never emit more entries than it consumed, so a region of ceil(chunks / PIPE_WAVES) * 64
// entries always suffices.  With one atomic per wave and queue on three shared counters the shade
// stage spent 87 % of its time waiting (2.2 ms for 8.3 M paths); the regions took that away.
struct Radiance { float x, y, z; };
struct PipeBuffers {
    float4 *xa[2], *xb[2], *xc[2];              // extend queues (ping-pong): path records
    uint4 *xd[2];
    float4 *pa, *pb, *pc, *p0, *p1, *p2;        // probe queue: shadow records from the front of a region, glossy from its back
    uint4 *pd;
    float4 *h0, *kd, *ks;
    Radiance *energy;            // 12 bytes per path: this plane is written once and read once by every path of a pass
    uint32_t *n_ext[2], *n_shadow, *n_glossy;   // fill count of each region
    uint32_t *work;                             // [N_RUNS] chunks of the depth-0 kernel handed out so far, per part of the target;
                                                // then [MAX_ROUNDS + 1] `more` flags: more[r] != 0 = round r of the pass left chunks undone
    uint32_t *resume;                           // [PIPE_WAVES] the fixed share of each wave: how far the rounds so far have come
    uint32_t *claim;                            // [3][PIPE_WAVES] pool blocks of each region handed out so far, for the extend, the shadow
                                                // and the glossy stage; each zero when its stage starts
    unsigned long long *pool_map;               // [PIPE_WAVES / 64] one bit per region whose pool may have blocks left (k_extend)
    uint32_t region_cap;
    uint32_t round, rounds;                     // round of the pass these launches belong to, and how many it has (see MAX_ROUNDS)
    uint32_t defer_below;                       // a probe search of the depth-0 kernel wants this many lanes, else the chunk parks its probes; 0: never
    uint32_t dynamic_share;                     // 1/dynamic_share of the depth-0 kernel's rounds is handed out as waves become free
    uint32_t dynamic_group;                     // ... in groups of so many consecutive chunks per wave (DYNAMIC_GROUP)
    unsigned long long *wstats;                 // [PIPE_WAVES][WSTAT_SLOTS] per-wave work counters (see WSTAT_*)
};

enum { WSTAT_PATHS = 0, WSTAT_RAYS, WSTAT_SHADOW, WSTAT_GLOSSY, WSTAT_EXT_NODES, WSTAT_EXT_TRIS, WSTAT_PROBE_NODES,
       WSTAT_PROBE_TRIS, WSTAT_DEPTH1, WSTAT_DEFERRED, WSTAT_PARKED, WSTAT_SLOTS = 16 };

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
    b.energy[id] = Radiance{ps.energy.x, ps.energy.y, ps.energy.z};
}

// Even shares of a queue for the tail kernel.  A stage reads what the stage before left in the 4096 regions, and
// regions fill unevenly (a wave whose tiles show the object leaves many survivors, one that saw sky none): a wave that
// reads its own region only takes as long as that region is full, and the kernel as long as its fullest one (wave
// slots in use during the tail kernel of a RECURSION 16 pass: 32 %).  The records of all regions, taken in region order,
// form one sequence; wave w reads the w-th of 4096 equal slices of it, wherever its records lie.  The tail kernel
// writes no queue, so nothing else changes (_ObiWan RECURSION 16: +3.6 %).  The extend stage could read its input the
// same way (a wave still writes to its own regions, and never more than it read) but measured slower with it, cubes
// -2 %, _ObiWan -1 %; a probe stage may not: it adds to the region the extend stage of the same depth has already
// written to, and must stay with the records that stage counted for it.
struct SliceCursor { uint32_t region, offset, left, count; };      // wave-uniform; count = fill of `region`
static_assert(PIPE_WAVES == 64 * 64, "slice_begin sums the fill counts as 64 lanes x 64 regions");

RT_DEV uint32_t wave_scan_inclusive(uint32_t v) {
    const uint32_t lane = lane_index();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= (uint32_t)off) v += up;
    }
    return v;
}
// The slice of wave_index() in the sequence of all records of a queue whose fill counts are counts[PIPE_WAVES].
RT_DEV SliceCursor slice_begin(const uint32_t *__restrict__ counts) {
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint4 *mine = (const uint4 *)(counts + lane * 64u);
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const uint4 c = mine[j]; s += (c.x + c.y) + (c.z + c.w); }
    const uint32_t inc = wave_scan_inclusive(s);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const uint32_t a = (uint32_t)((unsigned long long)total * wave / (uint32_t)PIPE_WAVES);
    const uint32_t e = (uint32_t)((unsigned long long)total * (wave + 1u) / (uint32_t)PIPE_WAVES);
    SliceCursor c;
    c.region = 0; c.offset = 0; c.count = 0; c.left = e - a;
    if (c.left == 0) return c;
    // the lane whose 64 regions hold record a, then the region among them
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(a < inc);
    const uint32_t l1 = (uint32_t)__builtin_ctzll(m1);
    const uint32_t before1 = (uint32_t)__shfl((int)(inc - s), (int)l1, 64);
    const uint32_t s2 = counts[l1 * 64u + lane];
    const uint32_t inc2 = wave_scan_inclusive(s2);
    const uint32_t a2 = a - before1;
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(a2 < inc2);
    const uint32_t l2 = (uint32_t)__builtin_ctzll(m2);
    const uint32_t before2 = (uint32_t)__shfl((int)(inc2 - s2), (int)l2, 64);
    c.region = l1 * 64u + l2;
    c.offset = a2 - before2;
    c.count = (uint32_t)__shfl((int)s2, (int)l2, 64);
    return c;
}
// Hands the next n records of the slice (n <= 64, n <= c.left) to the lanes with rank 0 .. n-1; `got` tells a lane whether
// it received one, the value returned is the record's position: region and offset inside it.
struct SlicePos { uint32_t region, offset; };
RT_DEV SlicePos slice_take(SliceCursor &c, const uint32_t *__restrict__ counts, uint32_t n, uint32_t rank, bool &got) {
    SlicePos pos;
    pos.region = 0; pos.offset = 0;
    got = false;
    uint32_t acc = 0;
    while (acc < n) {
        if (c.offset >= c.count) {          // on to the next region (its fill is read once, not at every call)
            if (c.region + 1u >= (uint32_t)PIPE_WAVES) { c.left = n; break; }    // cannot happen while the counts hold still
            c.region++;
            c.offset = 0;
            c.count = (uint32_t)__builtin_amdgcn_readfirstlane((int)counts[c.region]);
            continue;
        }
        const uint32_t avail = c.count - c.offset, want = n - acc;
        const uint32_t use = avail < want ? avail : want;
        if (rank >= acc && rank < acc + use) { pos.region = c.region; pos.offset = c.offset + (rank - acc); got = true; }
        acc += use;
        c.offset += use;
    }
    c.left -= n;
    return pos;
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

// A probe record (see the head of this file) at position i of the probe queue.
RT_DEV void store_probe_rec(const PipeBuffers &b, size_t i, const PathState &ps, uint32_t id, int hit, uint32_t kept_flags,
                            const Bounce &bn, bool shadow) {
    b.pa[i] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    b.pb[i] = make_float4(ps.d.y, ps.d.z, ps.path.x, ps.path.y);
    b.pc[i] = make_float4(ps.path.z, ps.energy.x, ps.energy.y, ps.energy.z);
    b.pd[i] = make_uint4(ps.seed, id, (uint32_t)hit | (kept_flags << 30), 0u);
    b.p0[i] = make_float4(bn.probe_d.x, bn.probe_d.y, bn.probe_d.z, bn.max_t);
    if (shadow) {
        b.p1[i] = make_float4(bn.lit.x, bn.lit.y, bn.lit.z, bn.roughness);
        b.p2[i] = make_float4(bn.albedo.x, bn.albedo.y, bn.albedo.z, 0.0f);
    } else {
        b.p1[i] = make_float4(bn.c.x, bn.c.y, bn.c.z, bn.w);
    }
}

// Shading of one path after its closest hit is known (raytracer.glsl:452-556): adds the sky on a
// miss, otherwise runs shade_pre and hands the path to the shadow queue, the glossy queue or (no
// probe needed) through the rest of the bounce to the next extend queue.  Called by whole waves.
// `tab` holds the light seeds as they stand at this depth (the host advances its copy of the table from launch to launch).
RT_DEV void shade_and_route(const DScene &sc, const DFrame &fr, const PipeBuffers &b, int depth, int q_in, const PassTables &tab,
                            uint32_t paths_per_sample, int max_depth, bool live, uint32_t id, PathState &ps, uint32_t kept_flags,
                            int hit, float hu, float hv, float ht, RouteFill &fill) {
    bool to_shadow = false, to_glossy = false, go_on = false;
    Bounce bn;
    bn.kind = PROBE_NONE;
    if (live) {
        if (hit < 0) {
            ps.energy = ps.energy + sky_color(sc, fr, ps.d) * ps.path;             // :452-456
        } else {
            uint32_t ls = tab.light_seed[id / paths_per_sample];
            const bool coin_diffuse = draw_coin(ls);
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
    if (to_shadow | to_glossy)
        store_probe_rec(b, to_shadow ? base + s_slot : base + (b.region_cap - 1u - g_slot), ps, id, hit, kept_flags, bn, to_shadow);
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

// ---------------------------------------------------------------------------------------------------
// Candidate lists of the camera rays.  All camera rays of a frame start at one point, and the rays of one 8x8 pixel
// tile (every SAMPLE of it: the jitter stays inside the pixel) lie in a narrow pyramid.  Once per frame, for every tile,
// the triangles that pyramid can touch are listed (k_build_tile_lists); the depth-0 kernel then folds a camera ray
// over its tile's list in ascending triangle index with the shader's own test (closest_hit_list), which is the
// reference's loop (raytracer.glsl:217-235) with the triangles left out that cannot pass its test for any ray of the
// tile: same result, bit for bit, without a hierarchy traversal.  The list of a wave's tile is wave-uniform, so ids
// and triangle records come through scalar loads.
//   A triangle is left out only if (a) its plane distance `relative_depth` (raytracer.glsl:137, the same expression on the
//   same values for every camera ray) is <= 0, which the shader rejects, or (b) its three corners lie outside one of the four
//   side planes of the tile's pyramid, widened by a quarter pixel and tested with a relative margin of 1e-4.
// A list holds up to TILE_LIST_CAP triangles; a tile that would need more keeps the hierarchy (count 0xFFFF), and so do
// frames whose camera matrix is not finite.  Rays that are not finite take the loop over all triangles, where every
// comparison being false makes the last triangle win.
constexpr uint32_t TILE_LIST_CAP = 31;            // 16 words per tile: count and up to 31 triangle ids, 16 bits each
constexpr uint32_t TILE_LIST_WORDS = 16;
constexpr uint32_t TILE_LIST_OVERFLOW = 0xFFFFu;
constexpr int TILE_LIST_MAX_TRIANGLES = 16384;    // scenes beyond this keep the hierarchy for their camera rays
struct DTileLists { const uint32_t *lists; int enabled; };

__global__ __launch_bounds__(64) void k_build_tile_lists(DScene sc, DFrame fr, uint32_t tiles_x, uint32_t tiles_y,
                                                         uint32_t *__restrict__ lists) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    __shared__ uint32_t ids[TILE_LIST_CAP + 1];
    uint32_t tile_x, tile_y;
    tile_position(tile, tiles_x, tiles_y, tile_x, tile_y);
    const float W = (float)fr.width, H = (float)fr.height;
    const float inv_h = 1.0f / H, inv_w = 1.0f / W, margin = 0.25f * inv_h;
    // the camera-space directions (a, b, 1/2) of the tile's rays before they are normalised (raytracer.glsl:602)
    const float a0 = ((float)(int)(tile_x * 8u) - 0.5f * W + 0.5f) * inv_h - margin;
    const float a1 = ((float)(int)(tile_x * 8u + 7u) - 0.5f * W + 0.5f) * inv_h + inv_w + margin;
    const float b0 = ((float)(int)(tile_y * 8u) + 0.5f) * inv_h - 0.5f - margin;
    const float b1 = ((float)(int)(tile_y * 8u + 7u) + 0.5f) * inv_h - 0.5f + inv_h + margin;
    const float *M = fr.cam;
    auto world = [&](float a, float b) {
        return mk3((M[0] * a + M[4] * b) + M[8] * 0.5f, (M[1] * a + M[5] * b) + M[9] * 0.5f, (M[2] * a + M[6] * b) + M[10] * 0.5f);
    };
    const f3 c0 = world(a0, b0), c1 = world(a1, b0), c2 = world(a1, b1), c3 = world(a0, b1);
    const f3 mid = world(0.5f * (a0 + a1), 0.5f * (b0 + b1));
    const f3 o = camera_origin(fr);
    const float check = (((o.x + o.y) + o.z) + ((c0.x + c0.y) + c0.z)) + (((c2.x + c2.y) + c2.z) + ((c1.x + c3.x) + (c1.y + c3.y)) + (c1.z + c3.z));
    bool usable = (check - check) == 0.0f;          // no NaN, no infinity
    // side planes through the origin, normals pointing inwards (towards the middle ray); a plane that does not tell the
    // middle ray from the outside (a matrix that flattens the pyramid) is not used
    f3 n[4] = {cross(c0, c1), cross(c1, c2), cross(c2, c3), cross(c3, c0)};
    float nl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float s = dot(n[k], mid);
        if (s < 0.0f) n[k] = mk3(-n[k].x, -n[k].y, -n[k].z);
        if (!(s != 0.0f)) n[k] = mk3(0.0f, 0.0f, 0.0f);
        nl[k] = length(n[k]);
    }
    uint32_t count = 0;
    for (int base = 0; usable && base < sc.count && count <= TILE_LIST_CAP; base += 64) {
        const int i = base + (int)lane;
        bool keep = false;
        if (i < sc.count) {
            const float4 *g = sc.geo + 3 * (size_t)i;
            const float4 qa = g[0], qb = g[1], qc = g[2];
            const f3 tn = mk3(qa.x, qa.y, qa.z), tp = mk3(qa.w, qb.x, qb.y), tu = mk3(qb.z, qb.w, qc.x), tv = mk3(qc.y, qc.z, qc.w);
            // (a) relative_depth of intersectTriangle, the same value for every camera ray
            const f3 delta = o - tp;
            const float rd = dot(tn, delta);
            keep = !(rd <= 0.0f);
            // (b) all three corners outside one side plane
            const f3 v0 = tp - o, v1 = (tp + tu) - o, v2 = (tp + tv) - o;
            const float l0 = length(v0), l1 = length(v1), l2 = length(v2);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float t = 1e-4f * nl[k];
                const bool out = (dot(n[k], v0) < -t * l0) & (dot(n[k], v1) < -t * l1) & (dot(n[k], v2) < -t * l2);
                keep = keep & !out;
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (keep && count + rank < TILE_LIST_CAP) ids[1u + count + rank] = (uint32_t)i;      // ascending: lanes and rounds both ascend
        count += (uint32_t)__popcll(m);
    }
    __syncthreads();
    const uint32_t header = (usable && count <= TILE_LIST_CAP) ? count : TILE_LIST_OVERFLOW;
    if (lane < TILE_LIST_WORDS) {
        const uint32_t k0 = 2u * lane, k1 = 2u * lane + 1u;
        const uint32_t h0 = k0 == 0u ? header : (k0 <= count && count <= TILE_LIST_CAP ? ids[k0] : 0u);
        const uint32_t h1 = (k1 <= count && count <= TILE_LIST_CAP) ? ids[k1] : 0u;
        lists[(size_t)tile * TILE_LIST_WORDS + lane] = (h0 & 0xFFFFu) | (h1 << 16);
    }
}

// findIntersection (raytracer.glsl:217-235) over the `n` triangles of a tile's list (ids[1..n], 16 bits each, ascending).
// `ids` is wave-uniform.  Returns the triangle id (-1 = miss); isec as the fold leaves it.
// The ids and the triangle records come through scalar loads (the index is wave-uniform).  Asking for the next round's
// record before the current one is tested was tried and gained nothing: the other waves of the SIMD cover the wait.  So was
// keeping the list's words in a vector register (an id by v_readlane instead of a scalar load): cubes +0.5 %, _ObiWan -1.5 %.
RT_DEV int closest_hit_list(const DScene &sc, const uint32_t *ids, uint32_t n, const Ray &ray, bool active, f4 &isec,
                            BvhCounters &cnt) {
    typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
    const const_u32_ptr w = (const_u32_ptr)ids;
    isec.x = 0.0f; isec.y = 0.0f; isec.z = -1.0f; isec.w = 1.0f;
    int best = -1;
    const float s = ((ray.o.x + ray.o.y) + ray.o.z) + ((ray.d.x + ray.d.y) + ray.d.z);
    const bool finite = (s - s) == 0.0f;
    const bool run = active & finite;
    for (uint32_t k = 1; k <= n; k++) {
        const int i = (int)((w[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu);
        const Tri t = load_tri_uniform(sc.geo, (uint32_t)i);
        // (measurement builds: a triangle test per lane with a ray, or the 64 lane slots of the wave's test)
#if defined(RT_COUNT_WAVE_STEPS)
        cnt.tris += bvh_count_step();
#else
        cnt.tris += run ? bvh_count_step() : 0u;
#endif
        PlaneTest pt;
        const bool p = run & plane_test_closest(ray, t, isec, pt);
        if (__builtin_amdgcn_ballot_w64(p) != 0ull) {
            float rx, ry;
            if (p & inside_test(ray, t, pt, rx, ry)) {
                isec.x = rx; isec.y = ry; isec.z = pt.rd; isec.w = pt.det;
                best = i;
            }
        }
    }
    // a ray that is not finite passes every rejection of the shader's test: it needs all triangles, in order
    const bool redo = active & !finite;
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
        f4 s2;
        s2.x = 0.0f; s2.y = 0.0f; s2.z = -1.0f; s2.w = 1.0f;
        const int b2 = closest_hit_range(sc.geo, 0, sc.count, ray, redo, -1, s2, -1);
        if (redo) { isec = s2; best = b2; }
    }
    return best;
}

// Camera path of queue position `qi` at depth 0 (raytracer.glsl:586-607); `sample` is the SAMPLE of its slot.
// Returns false for the padding lanes of ragged tiles.
RT_DEV bool first_path(const DFrame &fr, uint32_t sample, uint32_t in_sample, uint32_t tiles_x, bool live, PathState &ps,
                       const float *matrix = nullptr) {
    const uint32_t lane = lane_index();
    uint32_t tile_x, tile_y;
    tile_position(in_sample >> 6, tiles_x, (fr.height + 7u) >> 3, tile_x, tile_y);
    const uint32_t tx = tile_x * 8u + (lane & 7u);
    const uint32_t ty = tile_y * 8u + (lane >> 3);
    ps.seed = pixel_seed(fr, tx, ty, sample);
    const Ray cam = camera_ray(fr, tx, ty, ps.seed, matrix);
    ps.o = cam.o; ps.d = cam.d;
    ps.path = mk3(1.0f, 1.0f, 1.0f);
    ps.energy = mk3(0.0f, 0.0f, 0.0f);
    return live && (tx < fr.width) && (ty < fr.height);
}

// Extend stage of the bounces after the first: closest hit of every path of the wave's queue region, then the
// shading up to the probe loop (shade_and_route).
//
// Regions hold about the same number of records, but the records differ in cost, and with every wave bound to its own
// region a tenth (424 triangles) to a quarter (_ObiWan) of the wave slots stood empty while the slowest waves finished.
// So the last quarter of every region's 64-record blocks is a pool: its blocks are handed out through a counter per
// region (b.claim), to the region's own wave when it gets there and to any wave that has run out of work before
// (it looks through the regions after its own, 64 at a time).  Survivors go to the regions of the wave that processed
// them, which takes foreign blocks only while its regions have room for all of their paths; a path's arithmetic does
// not depend on where it is processed, so the image stays the same bits.
#ifndef RT_POOL_SHARE
#define RT_POOL_SHARE 4
#endif
constexpr uint32_t POOL_SHARE = RT_POOL_SHARE;       // 1/POOL_SHARE of a region's blocks can be taken by other waves; 0: none
#ifndef RT_PROBE_POOL_SHARE
#define RT_PROBE_POOL_SHARE 2
#endif
constexpr uint32_t PROBE_POOL_SHARE = RT_PROBE_POOL_SHARE;      // the same for the probe stages
#ifndef RT_STEAL_TRIES
#define RT_STEAL_TRIES 1
#endif
constexpr uint32_t STEAL_TRIES = RT_STEAL_TRIES;     // a wave gives up looking for foreign blocks after so many claims that came too late
#ifndef RT_STEAL_BLOCKS
#define RT_STEAL_BLOCKS 4
#endif
constexpr uint32_t STEAL_BLOCKS = RT_STEAL_BLOCKS;   // blocks a wave takes from another region's pool with one claim
RT_DEV uint32_t claim_block(uint32_t *claim, uint32_t region, uint32_t n = 1u) {
    uint32_t c = 0;
    if (lane_index() == 0) c = atomicAdd(&claim[region], n);
    return c;       // valid in lane 0
}
// Hands a wave the 64-record blocks it is to process, one per call of feed_next: its own blocks before the pool, then
// the pool of its own region (the next claim is under way while a block is processed), then blocks from the pools of
// other regions.  `counts` = records per region, `claim` = the stage's counters (zero at the start of the kernel),
// `share`: 1/share of a region's blocks are its pool (1: all of them).  All values are wave-uniform.
struct BlockFeed {
    const uint32_t *counts;
    uint32_t *claim;
    uint32_t share, total, n_blocks, pool0;
    int phase;
    uint32_t next_own, ahead, failed, pick;
    uint32_t got_src, got_total, got_next, got_end;      // blocks of another region this wave has claimed
};
RT_DEV uint32_t feed_pool_first(uint32_t n_blocks, uint32_t share) { return share ? n_blocks - n_blocks / share : n_blocks; }
RT_DEV void feed_begin(BlockFeed &f, const PipeBuffers &b, const uint32_t *counts, uint32_t *claim, uint32_t share) {
    const uint32_t wave = wave_index();
    f.counts = counts; f.claim = claim; f.share = share;
    f.total = counts[wave];
    f.n_blocks = (f.total + 63u) >> 6;
    f.pool0 = feed_pool_first(f.n_blocks, share);
    f.phase = 0;
    f.next_own = 0; f.ahead = 0; f.failed = 0; f.pick = wave * 2654435761u;
    f.got_src = f.got_total = f.got_next = f.got_end = 0;
    if (share && f.n_blocks > f.pool0 && lane_index() == 0) atomicOr(&b.pool_map[wave >> 6], 1ull << (wave & 63u));
}
// The next block: block `block` of region `src`, which holds `src_total` records; false when there is none.
// `room_for_foreign`: this wave's regions can take all the paths of the blocks of another claim (64 * STEAL_BLOCKS).
RT_DEV bool feed_next(BlockFeed &f, const PipeBuffers &b, bool room_for_foreign, uint32_t &src, uint32_t &src_total, uint32_t &block) {
    const uint32_t lane = lane_index(), wave = wave_index();
    for (;;) {
        src = wave; src_total = f.total;
        if (f.phase == 0) {
            if (f.next_own < f.pool0) { block = f.next_own++; return true; }
            if (!f.share) return false;
            f.phase = 1;
            f.ahead = claim_block(f.claim, wave);
        } else if (f.phase == 1) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.ahead);
            if (f.pool0 + c < f.n_blocks) {
                f.ahead = claim_block(f.claim, wave);
                block = f.pool0 + c;
                return true;
            }
            // the region's pool is used up (every wave that finds it so says so; the owner always gets here, so the
            // map is all zeros again when the kernel ends)
            if (lane == 0 && f.n_blocks > f.pool0) atomicAnd(&b.pool_map[wave >> 6], ~(1ull << (wave & 63u)));
            f.phase = 2;
        } else if (f.got_next < f.got_end) {
            src = f.got_src; src_total = f.got_total; block = f.got_next++;
            return true;
        } else {
            // not for ever: near the end many waves look for the same few blocks
            if (f.failed >= STEAL_TRIES || !room_for_foreign) return false;
            // b.pool_map: one bit per region whose pool may have blocks left (lane l reads the word of regions 64 l ...).  The
            // words change under way: read past this XCD's L2, which may hold an old value for as long as it likes.
            const unsigned long long word = __hip_atomic_load(&b.pool_map[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long lanes = __builtin_amdgcn_ballot_w64(word != 0ull);
            if (lanes == 0ull) return false;
            // a set bit picked by a number that differs from wave to wave and from try to try, so that the idle waves spread
            // over the regions that have work left
            f.pick = f.pick * 1664525u + 1013904223u;
            const uint32_t r1 = (f.pick >> 8) & 63u, r2 = (f.pick >> 16) & 63u;
            const unsigned long long lanes_rot = (lanes >> r1) | (r1 ? lanes << (64u - r1) : 0ull);
            const uint32_t l = ((uint32_t)__builtin_ctzll(lanes_rot) + r1) & 63u;
            const unsigned long long w = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)word, (int)l) |
                                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(word >> 32), (int)l) << 32);
            const unsigned long long w_rot = (w >> r2) | (r2 ? w << (64u - r2) : 0ull);
            const uint32_t victim = l * 64u + (((uint32_t)__builtin_ctzll(w_rot) + r2) & 63u);
            const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.counts[victim]);
            const uint32_t nb = (t + 63u) >> 6, p0 = feed_pool_first(nb, f.share);
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim_block(f.claim, victim, STEAL_BLOCKS));
            if (p0 + c >= nb) {       // used up: another wave was faster
                if (lane == 0) atomicAnd(&b.pool_map[victim >> 6], ~(1ull << (victim & 63u)));
                f.failed++;
                continue;
            }
            src = victim; src_total = t; block = p0 + c;
            f.got_src = victim; f.got_total = t; f.got_next = block + 1u;
            f.got_end = block + STEAL_BLOCKS < nb ? block + STEAL_BLOCKS : nb;
            return true;
        }
    }
}

template <class CFG>
__global__ __launch_bounds__(64 * WG_WAVES, CFG::BVH ? RT_MIN_WAVES : 1) void k_extend(DScene sc, DBvh bv, DFrame fr, PipeBuffers b, int depth, int q_in,
                                                          PassTables tab, uint32_t paths_per_sample, int max_depth) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, bv, false, stack_mem, scene_mem, share_mem, ws, lws, share);
    stage_shading<CFG>(sc, shade_mem);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    uint32_t n_rays = 0;
    RouteFill fill;
    fill.shadow = fill.glossy = fill.next = 0;
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
    unsigned long long xclk[4] = {0, 0, 0, 0};
#endif

    BlockFeed feed;
    feed_begin(feed, b, b.n_ext[q_in], b.claim, POOL_SHARE);
    for (;;) {
        // foreign blocks only while this wave's regions can take all their paths (what goes to the probe regions may come
        // on to the next extend region)
        uint32_t src, src_total, block;
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
        unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
        if (!feed_next(feed, b, fill.next + fill.shadow + fill.glossy + 64u * STEAL_BLOCKS <= b.region_cap, src, src_total, block)) break;
        const uint32_t qi = block * 64u + lane;
        const bool live = qi < src_total;
        uint32_t id = 0, kept_flags = 0;
        PathState ps;
        load_path_rec(b, q_in, (size_t)src * b.region_cap + (live ? qi : 0u), ps, id, kept_flags);
        n_rays += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
        // measurement build: cycles spent getting a block and its records, in the closest hit, in shading and routing
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t1_ = __builtin_amdgcn_s_memtime();
        xclk[0] += t1_ - t0_;
#endif
        f4 isec;
        const int hit = closest_hit_cfg<CFG>(sc, bv, ray, live, ws, isec, bc);
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
        unsigned long long t2_ = __builtin_amdgcn_s_memtime();
        xclk[1] += t2_ - t1_;
#endif
        const float inv_det = 1.0f / isec.w;
        shade_and_route(sc, fr, b, depth, q_in, tab, paths_per_sample, max_depth, live, id, ps, kept_flags, hit,
                        isec.x * inv_det, isec.y * inv_det, isec.z * inv_det, fill);
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xclk[2] += __builtin_amdgcn_s_memtime() - t2_;
        xclk[3] += 1;
#endif
    }
    store_route_fill(b, q_in, fill);
    wstat_add(b, WSTAT_RAYS, n_rays);
#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS == 3
    wstat_add(b, WSTAT_EXT_NODES, xclk[0]);
    wstat_add(b, WSTAT_EXT_TRIS, xclk[1]);
    wstat_add(b, WSTAT_PROBE_NODES, xclk[2]);
    wstat_add(b, WSTAT_PROBE_TRIS, xclk[3]);
#else
    if (CFG::BVH) {
        wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
    }
#endif
}

// A probe search of the depth-0 kernel runs with at least this many lanes (own probes + parked records taken up by idle
// lanes); a chunk that cannot muster them parks its probes (PipeBuffers::defer_below, see k_first).  Scenes from
// DEFER_MAX_TRIANGLES triangles on do not park (their probe searches are long whatever the lanes' neighbours do).
#ifndef RT_DEFER_BELOW
#define RT_DEFER_BELOW 48
#endif
constexpr uint32_t DEFER_BELOW = RT_DEFER_BELOW;
constexpr int DEFER_MAX_TRIANGLES = 65536;
constexpr int LARGE_SCENE_TRIANGLES = 65536;     // from here on the whole depth-0 kernel hands its chunks out dynamically

// Depth 0 in one kernel: camera ray, closest hit, shading, the bounce's light probe and the rest of the bounce.
// At depth 0 the 64 lanes of a wave are the pixels of one 8x8 tile of one sample, so they all draw the same
// coin (raytracer.glsl:553: light_seed depends on SAMPLE and depth only) and run the same kind of probe, from
// neighbouring surface points: the probe can run in place with full lanes, and the probe records (112 bytes
// per path written and read back) are never made.  Deeper bounces mix samples in a wave and keep the queues.
// LATER: the instantiation launched for the rounds of a pass after the first (see MAX_ROUNDS), the same code under a
// name of its own, so that a profile tells the launches that render a pass from those that usually return at once.
template <class CFG, bool LATER>
__global__ __launch_bounds__(64 * WG_WAVES, CFG::SHARE ? RT_MIN_WAVES : 1) void k_first(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b,
                                                                              PassTables tab, uint32_t paths_per_sample,
                                                                              int max_depth, DTileLists tl) {
    if (LATER && round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = paths_per_sample * tab.n;
    const uint32_t tiles_x = (fr.width + 7u) >> 3;
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, lbv, true, stack_mem, scene_mem, share_mem, ws, lws, share);
    stage_shading<CFG>(sc, shade_mem);
    // the camera matrix is read from LDS, once per chunk: 16 scalar registers fewer for the whole kernel, whose arguments do
    // not fit the scalar registers (what overflows is parked in vector lanes and fetched with v_readlane: 440 of them in
    // this kernel before, 305 after; _ObiWan +2.3 %, 424 triangles +0.5 %)
    __shared__ float lds_cam[16];
    if (threadIdx.x < 16) lds_cam[threadIdx.x] = fr.cam[threadIdx.x];
    __syncthreads();
    BvhCounters bc, pc;
    bc.nodes = 0; bc.tris = 0; pc.nodes = 0; pc.tris = 0;
    uint32_t n_paths = 0, n_shadow = 0, n_glossy = 0, fill_next = 0;
    uint32_t q_shadow = 0, q_glossy = 0;       // probe records parked in this wave's region at present
    uint32_t n_parked = 0;                     // ... and how many were ever written
    const bool deeper = 1 < max_depth;

#if defined(RT_SECTION_CLOCKS) && RT_SECTION_CLOCKS > 0
    unsigned long long clk[4] = {0, 0, 0, 0};
#define SECTION_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); clk[k] += now_ - mark_; mark_ = now_; } while (0)
#else
#define SECTION_MARK(k) do { } while (0)
#define RT_SECTION_CLOCKS 0
#endif
#ifndef RT_PROBE_HISTO
#define RT_PROBE_HISTO 0
#endif
#if RT_PROBE_HISTO > 0
    unsigned long long histo[4] = {0, 0, 0, 0};
#endif
    // A chunk is one tile of one sample.  Chunks are numbered tile-major (all samples of a tile follow each other), so
    // that the waves at work at the same time are on few tiles: the nodes and triangles under a tile are fetched once
    // and then found in the caches by the tile's other samples (slot-major order meets a tile again only after the whole
    // image, i.e. after the whole scene, has gone through the caches).
    const uint32_t n_tiles = paths_per_sample >> 6;
    const uint32_t n_chunks = n_tiles * tab.n;
    // Who renders which chunk.  Most of the pass is dealt out in fixed shares: in round r the waves take the chunks
    // r * PIPE_WAVES ... in the XCD-aware order of first_chunk_of_wave(), so that every wave, and with it every queue
    // region, gets the same number of chunks from all over the image (the deeper stages work region by region and
    // take as long as their fullest region).  But chunks differ a lot in cost (sky, surfaces, lights), and with fixed
    // shares alone a fifth of the wave slots stood empty while the slowest waves finished (SQ_WAVE_CYCLES against waves
    // x kernel time).  So the last third of the rounds (all of them for large scenes: b.dynamic_share) is handed out
    // chunk by chunk as waves become free, through N_RUNS counters (b.work; one counter sustains about ten million
    // atomic operations a second, they are carried out beyond the XCDs' own L2s, and the waves wait for them: with 32
    // counters instead of 128 the kernel takes 8 % longer, with 16 counters 37 %): run x holds the blocks x,
    // x + N_RUNS, ... of CHUNK_BLOCK consecutive chunks of that part; a wave starts on a run of its own XCD and goes on
    // to the others when it is used up.  Which wave renders a chunk only decides in which queue region its survivors
    // wait, the image does not depend on it.  The counter value for the next chunk is fetched while the current chunk
    // is rendered.
    const uint32_t n_rounds = (n_chunks + (uint32_t)PIPE_WAVES - 1u) / (uint32_t)PIPE_WAVES;
    // (a whole number of groups of FIXED_GROUP rounds: in the fixed part a wave renders FIXED_GROUP consecutive chunks, i.e.
    // samples of one tile, before it moves on, so that the probe records a chunk parks are taken up by a chunk of the same tile)
    const uint32_t fixed_rounds = (RT_DYNAMIC_CHUNKS ? n_rounds - n_rounds / b.dynamic_share : n_rounds) / FIXED_GROUP * FIXED_GROUP;
    const uint32_t fixed_chunks = fixed_rounds * (uint32_t)PIPE_WAVES;          // may exceed n_chunks when nothing is left
    const uint32_t n_blocks = n_chunks > fixed_chunks ? (n_chunks - fixed_chunks + CHUNK_BLOCK - 1u) / CHUNK_BLOCK : 0u;
    const uint32_t home = (blockIdx.x % N_XCD) * RUNS_PER_XCD + (blockIdx.x / N_XCD) % RUNS_PER_XCD;
    const uint32_t first = first_chunk_of_wave();
    uint32_t round = LATER ? b.resume[wave] : 0u, part = 0, run_x = home, run_count = 0, ahead = 0;
    bool have_ahead = false;
    // this wave's region is full while chunks may be left (other waves may still take them): the pass gets another round.
    // In the last round the regions that have room left can hold everything that is left (region_cap_for), so a wave that
    // stalls there loses nothing; rt_get_stats checks that by counting the camera paths that were rendered.
    bool stalled = false;
    auto enter_part = [&]() {
        run_x = run_of_part(home, part);
        run_count = n_blocks > run_x ? ((n_blocks - run_x + N_RUNS - 1u) / N_RUNS) * CHUNK_BLOCK : 0u;
    };
    // (`n` consecutive chunks of the run with one operation: samples of one tile, rendered by this wave one after the other,
    // so that what the first fetched the others find in this CU's L1, see DYNAMIC_GROUP)
    auto take = [&](uint32_t n) -> uint32_t {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&b.work[run_x], n);
        return c;
    };
    // a run that is used up is seen with a plain read; only a run that has chunks left costs an atomic operation
    auto used_up = [&]() -> bool { return __builtin_nontemporal_load(&b.work[run_x]) >= run_count; };
    uint32_t grp_next = 0, grp_left = 0, ahead_n = 0;       // the rest of the group this wave has claimed; chunks of the claim under way
    enter_part();
    for (;;) {
        uint32_t chunk;
        if (round < fixed_rounds) {
            if (fill_next + q_shadow + q_glossy + 64u > b.region_cap) { stalled = true; break; }
            chunk = ((round / FIXED_GROUP) * (uint32_t)PIPE_WAVES + first) * FIXED_GROUP + round % FIXED_GROUP;
            round++;
        } else {
            uint32_t c;
            if (grp_left != 0u) {
                c = grp_next++;
                grp_left--;
                if (c >= run_count) { grp_left = 0u; continue; }       // the run ended inside the group
            } else {
                uint32_t n;
                if (have_ahead) {
                    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ahead);
                    n = ahead_n;
                } else {
                    if (used_up()) {
                        if (++part == N_RUNS) break;
                        enter_part();
                        continue;
                    }
                    const uint32_t held = fill_next + q_shadow + q_glossy;
                    if (held + 64u > b.region_cap) { stalled = true; break; }        // this wave's regions cannot take another chunk whole
                    // a whole group if the regions hold it whatever survives, else one chunk (so that a wave still gets as far
                    // in a round as the sizing of the regions promises)
                    n = held + 64u * b.dynamic_group <= b.region_cap ? b.dynamic_group : 1u;
                    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)take(n));
                }
                have_ahead = false;
                if (c >= run_count) {
                    if (++part == N_RUNS) break;
                    enter_part();
                    continue;
                }
                grp_next = c + 1u;
                grp_left = n - 1u;
                // the next group is asked for now if the regions hold this one and the next whatever survives
                have_ahead = fill_next + q_shadow + q_glossy + 64u * (n + b.dynamic_group) <= b.region_cap;
                if (have_ahead) { ahead = take(b.dynamic_group); ahead_n = b.dynamic_group; }
            }
            const uint32_t block_in_run = c / CHUNK_BLOCK;
            chunk = fixed_chunks + (block_in_run * N_RUNS + run_x) * CHUNK_BLOCK + (c - block_in_run * CHUNK_BLOCK);
        }
        if (chunk >= n_chunks) continue;
#if RT_SECTION_CLOCKS > 0
        unsigned long long mark_ = __builtin_amdgcn_s_memtime();
#endif
        // A chunk is one tile of one sample: slot, SAMPLE and the coin are the same for the whole wave.  Chunks are numbered
        // tile-major (all samples of a tile follow each other), so that the waves at work at the same time are on few
        // tiles: the nodes and triangles under a tile are fetched once and then found in the caches by the tile's other
        // samples (slot-major order meets a tile again only after the whole image, i.e. the whole scene, has gone through
        // the caches).  The slot is rotated by the round, by an amount that depends on the tile alone so that the chunks of
        // a tile still cover every slot once: a wave whose chunks all had the same slot, as happens when the pass size
        // divides the wave count, would draw the same coin for all its paths, and the probe stages of the next depths
        // would find their work in half of the regions.
        const uint32_t tile_of_chunk = chunk / tab.n, first_of_tile = tile_of_chunk * tab.n;
        const uint32_t slot = (chunk - first_of_tile + first_of_tile / (uint32_t)PIPE_WAVES) % tab.n;
        const uint32_t base = slot * paths_per_sample + tile_of_chunk * 64u;
        const uint32_t id = base + lane;
        PathState ps;
        const bool live = first_path(fr, sample_of_slot(tab, slot), base - slot * paths_per_sample, tiles_x, id < total, ps, lds_cam);
        n_paths += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        int hit;
        // the tile's candidate list if it has one, else the hierarchy (or the loop over all triangles)
        const uint32_t *list = tl.lists + (size_t)tile_of_chunk * TILE_LIST_WORDS;
        const uint32_t listed = (CFG::BVH && tl.enabled) ? (*(const __attribute__((address_space(4))) uint32_t *)list & 0xFFFFu) : TILE_LIST_OVERFLOW;
#if RT_PROBE_HISTO == 3
        // measurement build: chunks by the length of their tile's candidate list (up to 4, 5-12, 13-31, none: hierarchy)
        histo[listed == TILE_LIST_OVERFLOW ? 3u : (listed <= 4u ? 0u : (listed <= 12u ? 1u : 2u))]++;
#endif
        if (listed != TILE_LIST_OVERFLOW) hit = closest_hit_list(sc, list, listed, ray, live, isec, bc);
        else hit = closest_hit_cfg<CFG>(sc, bv, ray, live, ws, isec, bc);
        SECTION_MARK(0);
        const float inv_det = 1.0f / isec.w;
        uint32_t ls = tab.light_seed[slot];
        const bool coin_diffuse = draw_coin(ls);
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
        SECTION_MARK(RT_SECTION_CLOCKS == 4 ? 0 : 1);         // 4: closest hit and shading booked together, the slot goes to parking / taking up
        Ray probe; probe.o = ps.o; probe.d = bn.probe_d;
        float probe_max_t = bn.max_t;
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        uint32_t kept_flags = 0;
        bool go_on = false;
        // The chunk's probes.  All lanes of a chunk want the same kind of probe, but not all of them want one: sky pixels
        // do not, and sampleLight gives up before its loop for every light triangle that faces away (half of them, lane
        // by lane).  A search with a third or half of the lanes idle wastes that share of the wave, so probes are PARKED
        // and TAKEN UP AGAIN: a chunk with too few probes writes them to the wave's region of the probe queue as records
        // (as the deeper bounces do for every probe) and goes on to the next chunk; in a later chunk of the same kind
        // the lanes that have no probe of their own retire their path first and then take a parked record each, search
        // for it and finish its bounce.  A search runs when own probes + records that fit the idle lanes reach
        // b.defer_below (or the regions are too full to park more).  Records still parked when the kernel ends are left
        // to the probe stages of depth 0.  Which lane finishes a path does not change its arithmetic.
        const bool wants = shaded && bn.kind == (coin_diffuse ? PROBE_SHADOW : PROBE_GLOSSY);
        bool parked = false, host = false;
        bool own_retired = false;       // wave-uniform: the lanes without a probe have retired their own paths before the search
        uint32_t path_id = id;          // the path a lane finishes after the search: its own, or that of the record it took up
        int path_hit = hit;
        size_t rec = 0;
        if (b.defer_below) {
            uint32_t &pending = coin_diffuse ? q_shadow : q_glossy;
            const unsigned long long want_mask = __builtin_amdgcn_ballot_w64(wants);
            const uint32_t n_wants = (uint32_t)__popcll(want_mask);
            const uint32_t n_take = pending < 64u - n_wants ? pending : 64u - n_wants;
            const size_t region = (size_t)wave * b.region_cap;
            // (parking only while the wave's regions are at most 7/8 full: these records must not be what makes a pass need a further round)
            const bool room = 8u * (fill_next + q_shadow + q_glossy + 64u) <= 7u * b.region_cap;
            if (n_wants + n_take < b.defer_below && room) {
                if (n_wants != 0u) {
                    parked = wants;
                    const uint32_t slot = region_slot(pending, wants);
                    if (wants) store_probe_rec(b, coin_diffuse ? region + slot : region + (b.region_cap - 1u - slot), ps, id, hit, 0u, bn, coin_diffuse);
                    n_parked += n_wants;
                }
            } else if (n_take != 0u) {
                // the own paths of the lanes without a probe end their bounce now (the probe function returned false, or the
                // ray left the scene): these lanes are free then
                bool own_on = false;
                if (shaded && !wants) {
                    f4 none;
                    none.x = none.y = none.z = 0.0f; none.w = 1.0f;
                    own_on = (coin_diffuse ? finish_diffuse(sc, ps, hit, false, bn.lit, bn.albedo, bn.roughness)
                                           : finish_glossy(sc, fr, ps, hit, 0, bn.probe_d, -1, none, bn.c, bn.w)) && deeper;
                }
                route_next(b, 1, fill_next, live && !wants, own_on, ps, id, 0u);
                own_retired = true;
                const unsigned long long idle = ~want_mask;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                host = !wants && rank < n_take;
                // the records were written by other lanes of this wave: their stores have to be done before they are read
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (host) {
                    const uint32_t k = pending - 1u - rank;          // the records parked last are taken first
                    rec = coin_diffuse ? region + k : region + (b.region_cap - 1u - k);
                    const float4 a = b.pa[rec], pr = b.p0[rec];
                    probe.o = mk3(a.x, a.y, a.z);
                    probe.d = mk3(pr.x, pr.y, pr.z);
                    probe_max_t = pr.w;
                    path_hit = (int)(b.pd[rec].z & CURRENT_MASK);
                }
                pending -= n_take;
            }
        }
        const bool run = (wants && !parked) || host;
#if RT_SECTION_CLOCKS == 4
        SECTION_MARK(1);
#endif
#if RT_PROBE_HISTO > 0
        // measurement build: searches of the depth-0 kernel by the number of lanes they run with (1: shadow, 2: glossy)
        if (RT_PROBE_HISTO < 3 && (RT_PROBE_HISTO == 1) == coin_diffuse) {
            const uint32_t n_run = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run));
            if (n_run) histo[n_run < 48u ? n_run / 16u : 3u]++;
        }
#endif
        // the lanes that finish a path after the search
        const bool fin = own_retired ? run : (shaded && !parked);
        f3 lit = bn.lit, colour = coin_diffuse ? bn.albedo : bn.c;
        float factor = coin_diffuse ? bn.roughness : bn.w;
        if (coin_diffuse) {
            bool blocked = true;
            if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                blocked = shadow_any_hit_cfg<CFG>(sc, bv, probe, probe_max_t, path_hit, run, ws, share, pc);
                n_shadow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run));
            }
            SECTION_MARK(2);
            if (host) {     // the rest of the record
                load_probe_path(b, rec, ps);
                const uint4 d = b.pd[rec];
                const float4 c = b.p1[rec], al = b.p2[rec];
                ps.seed = d.x; path_id = d.y;
                lit = mk3(c.x, c.y, c.z); factor = c.w;
                colour = mk3(al.x, al.y, al.z);
            }
            if (fin) {
                const bool written = run && !blocked;
                go_on = finish_diffuse(sc, ps, path_hit, written, lit, colour, factor) && deeper;
                // the kept light result matters to later bounces only
                if (written && go_on) { b.kd[path_id] = make_float4(lit.x, lit.y, lit.z, 0.0f); kept_flags |= KEPT_DIFF; }
            }
        } else {
            int outcome = 0, light_id = -1;
            f4 lsec;
            lsec.x = lsec.y = lsec.z = 0.0f; lsec.w = 1.0f;
            if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                outcome = glossy_search_cfg<CFG>(sc, bv, lbv, probe, path_hit, run, light_id, lsec, ws, lws, share, pc);
                n_glossy += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run));
            }
            SECTION_MARK(RT_SECTION_CLOCKS == 2 ? 3 : 2);     // 2: the glossy probe is booked on the last slot
            if (host) {
                load_probe_path(b, rec, ps);
                const uint4 d = b.pd[rec];
                const float4 c = b.p1[rec];
                ps.seed = d.x; path_id = d.y;
                colour = mk3(c.x, c.y, c.z); factor = c.w;
            }
            if (fin) {
                go_on = finish_glossy(sc, fr, ps, path_hit, run ? outcome : 0, probe.d, light_id, lsec, colour, factor) && deeper;
                if (run && outcome != 0 && go_on) {
                    b.ks[path_id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
                    kept_flags |= KEPT_SPEC;
                }
            }
        }
        route_next(b, 1, fill_next, own_retired ? run : (live && !parked), go_on, ps, path_id, kept_flags);
        SECTION_MARK(3);
    }
    if (lane == 0) {
        b.n_shadow[wave] = q_shadow;
        b.n_glossy[wave] = q_glossy;
        b.n_ext[1][wave] = fill_next;
        b.resume[wave] = round;
        if (stalled) more_flags(b.work)[b.round] = 1u;
    }
    wstat_add(b, WSTAT_PATHS, n_paths);
    wstat_add(b, WSTAT_RAYS, n_paths);
    wstat_add(b, WSTAT_DEPTH1, fill_next);
    wstat_add(b, WSTAT_DEFERRED, q_shadow + q_glossy);
    wstat_add(b, WSTAT_PARKED, n_parked);
    wstat_add(b, WSTAT_SHADOW, n_shadow);
    wstat_add(b, WSTAT_GLOSSY, n_glossy);
#if RT_PROBE_HISTO > 0
    // measurement build: the four hierarchy counters carry the numbers of searches with 1-15, 16-31, 32-47, 48-64 lanes instead
    wstat_add(b, WSTAT_EXT_NODES, histo[0]);
    wstat_add(b, WSTAT_EXT_TRIS, histo[1]);
    wstat_add(b, WSTAT_PROBE_NODES, histo[2]);
    wstat_add(b, WSTAT_PROBE_TRIS, histo[3]);
#elif RT_SECTION_CLOCKS == 3
    // (measurement build of the extend stage: the counters are its own)
#elif RT_SECTION_CLOCKS > 0
    // measurement build: the four hierarchy counters carry the cycles of the four sections instead
    wstat_add(b, WSTAT_EXT_NODES, clk[0]);
    wstat_add(b, WSTAT_EXT_TRIS, clk[1]);
    wstat_add(b, WSTAT_PROBE_NODES, clk[2]);
    wstat_add(b, WSTAT_PROBE_TRIS, clk[3]);
#else
    if (CFG::BVH) {
        wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(pc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(pc.tris));
    }
#endif
#undef SECTION_MARK
}

// The tail of the batch in one launch.  After a few bounces a small fraction of the paths is still under way, and
// a stage of the pipeline over so few records takes as long as its chain of dependent memory accesses, whatever
// the record count: 45-60 us per kernel plus launch gaps, three kernels per depth (12 depths of that are a fifth of
// a RECURSION 16 batch).  Here every lane walks the rest of its path: closest hit, shading, probe in place, rest of
// the bounce, next bounce, as in the single-kernel form but starting from the records of the extend queue.  Lanes
// of a wave end at different depths; a lane whose path has ended takes the next record of the wave's region, so
// every bounce runs with a full wave until the region is used up.  The kept light results stay in registers.
template <class CFG>
__global__ __launch_bounds__(64 * WG_WAVES, RT_MIN_WAVES) void k_tail(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b, int depth0,
                                                           int q_in, PassTables tab, uint32_t paths_per_sample, int max_depth) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, lbv, true, stack_mem, scene_mem, share_mem, ws, lws, share);
    stage_shading<CFG>(sc, shade_mem);
    BvhCounters bc, pc;
    bc.nodes = 0; bc.tris = 0; pc.nodes = 0; pc.tris = 0;
    uint32_t n_rays = 0, n_shadow = 0, n_glossy = 0;
    const uint32_t *counts = b.n_ext[q_in];
    SliceCursor cur = slice_begin(counts);
    // per-lane state of the path a lane is walking
    PathState ps;
    uint32_t id = 0, kept_flags = 0, light_seed = 0;
    bool alive = false;
    ps.o = ps.d = ps.path = ps.energy = ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
    ps.seed = 0;
    auto take_record = [&](SlicePos at) {
        load_path_rec(b, q_in, (size_t)at.region * b.region_cap + at.offset, ps, id, kept_flags);
        ps.kept_diff = ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
        if (kept_flags & KEPT_DIFF) { const float4 k = b.kd[id]; ps.kept_diff = mk3(k.x, k.y, k.z); }
        if (kept_flags & KEPT_SPEC) { const float4 k = b.ks[id]; ps.kept_spec = mk3(k.x, k.y, k.z); }
        light_seed = tab.light_seed[id / paths_per_sample];       // as it stands at depth0: the host has advanced the table
    };
    // one bounce of every lane with `alive`, each at its own `depth`
    auto bounce = [&](int depth) {
        n_rays += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(alive));
        Ray ray; ray.o = ps.o; ray.d = ps.d;
        f4 isec;
        const int hit = closest_hit_cfg<CFG>(sc, bv, ray, alive, ws, isec, bc);
        const float inv_det = 1.0f / isec.w;
        const bool coin_diffuse = draw_coin(light_seed);
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
            blocked = shadow_any_hit_cfg<CFG>(sc, bv, probe, bn.max_t, hit, run_s, ws, share, pc);
            n_shadow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_s));
        }
        int outcome = 0, light_id = -1;
        f4 lsec;
        lsec.x = lsec.y = lsec.z = 0.0f; lsec.w = 1.0f;
        if (__builtin_amdgcn_ballot_w64(run_g) != 0ull) {
            outcome = glossy_search_cfg<CFG>(sc, bv, lbv, probe, hit, run_g, light_id, lsec, ws, lws, share, pc);
            n_glossy += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(run_g));
        }
        bool go_on = false;
        if (shaded) {
            if (coin_diffuse) go_on = finish_diffuse(sc, ps, hit, run_s && !blocked, bn.lit, bn.albedo, bn.roughness);
            else go_on = finish_glossy(sc, fr, ps, hit, run_g ? outcome : 0, bn.probe_d, light_id, lsec, bn.c, bn.w);
        }
        go_on = go_on && (depth + 1 < max_depth);
        if (alive && !go_on) finish_path(b, id, ps);
        alive = alive && go_on;
    };
    // Lanes whose path has ended take the next record of the region, so the wave stays full while lanes are at
    // different depths of different paths.
    int depth = depth0;
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
        if (cur.left > 0 && idle != 0ull) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t take = n_idle < cur.left ? n_idle : cur.left;
            bool got;
            const SlicePos at = slice_take(cur, counts, take, alive ? 64u : rank, got);
            if (got) { take_record(at); depth = depth0; alive = depth0 < max_depth; }
        }
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        bounce(depth);
        depth++;
    }
    if (lane == 0) {
        b.n_shadow[wave] = 0;
        b.n_glossy[wave] = 0;
        b.n_ext[q_in ^ 1][wave] = 0;
    }
    wstat_add(b, WSTAT_RAYS, n_rays);
    wstat_add(b, WSTAT_SHADOW, n_shadow);
    wstat_add(b, WSTAT_GLOSSY, n_glossy);
    if (CFG::BVH) {
        wstat_add(b, WSTAT_EXT_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_EXT_TRIS, wave_sum_u32(bc.tris));
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(pc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(pc.tris));
    }
}

// Shading of the paths whose hit records the lane-refill form of the extend stage has left in h0.
__global__ __launch_bounds__(64 * WG_WAVES) void k_shade(DScene sc, DFrame fr, PipeBuffers b, int depth, int q_in, PassTables tab,
                                                         uint32_t paths_per_sample, int max_depth) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    const uint32_t total = b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    RouteFill fill;
    fill.shadow = fill.glossy = fill.next = 0;
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t qi = base + lane;
        const bool live = qi < total;
        uint32_t id = 0, kept_flags = 0;
        PathState ps;
        const size_t hi = region + (live ? qi : 0u);
        load_path_rec(b, q_in, hi, ps, id, kept_flags);
        const float4 h = b.h0[hi];
        shade_and_route(sc, fr, b, depth, q_in, tab, paths_per_sample, max_depth, live, id, ps, kept_flags,
                        __float_as_int(h.w), h.x, h.y, h.z, fill);
    }
    store_route_fill(b, q_in, fill);
}

// What the probe stages read of a probe record before the search.
struct ProbeHead {
    Ray probe;
    float max_t;
    uint32_t seed, id, kept_flags;
    int current;
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
}
// Rest of a bounce after its shadow probe (raytracer.glsl:557-579).  Returns whether the path goes on; only then is
// the kept light result stored (nothing reads it otherwise).
RT_DEV bool retire_shadow(const DScene &sc, const PipeBuffers &b, size_t i, const ProbeHead &h, bool blocked, bool deeper,
                          PathState &ps, uint32_t &kept_flags) {
    load_probe_path(b, i, ps);
    ps.seed = h.seed;
    const float4 c = b.p1[i], al = b.p2[i];
    kept_flags = h.kept_flags;
    ps.kept_diff = mk3(0.0f, 0.0f, 0.0f);
    if (blocked && (kept_flags & KEPT_DIFF)) { const float4 k = b.kd[h.id]; ps.kept_diff = mk3(k.x, k.y, k.z); }
    const bool go_on = finish_diffuse(sc, ps, h.current, !blocked, mk3(c.x, c.y, c.z), mk3(al.x, al.y, al.z), c.w) && deeper;
    if (!blocked && go_on) {
        b.kd[h.id] = make_float4(c.x, c.y, c.z, 0.0f);
        kept_flags |= KEPT_DIFF;
    }
    return go_on;
}
// Rest of a bounce after its glossy probe (raytracer.glsl:420-437, :557-579).
RT_DEV bool retire_glossy(const DScene &sc, const DFrame &fr, const PipeBuffers &b, size_t i, const ProbeHead &h, int outcome,
                          int light_id, f4 lsec, bool deeper, PathState &ps, uint32_t &kept_flags) {
    load_probe_path(b, i, ps);
    ps.seed = h.seed;
    const float4 c = b.p1[i];
    kept_flags = h.kept_flags;
    ps.kept_spec = mk3(0.0f, 0.0f, 0.0f);
    if (outcome == 0 && (kept_flags & KEPT_SPEC)) { const float4 k = b.ks[h.id]; ps.kept_spec = mk3(k.x, k.y, k.z); }
    const bool go_on = finish_glossy(sc, fr, ps, h.current, outcome, h.probe.d, light_id, lsec, mk3(c.x, c.y, c.z), c.w) && deeper;
    if (outcome != 0 && go_on) {
        b.ks[h.id] = make_float4(ps.kept_spec.x, ps.kept_spec.y, ps.kept_spec.z, 0.0f);
        kept_flags |= KEPT_SPEC;
    }
    return go_on;
}

// Shadow-probe stage: occlusion query of sampleLight (raytracer.glsl:372-378), then :557-579.
template <class CFG>
__global__ __launch_bounds__(64 * WG_WAVES, CFG::BVH ? RT_MIN_WAVES : 1) void k_probe_shadow(DScene sc, DBvh bv, PipeBuffers b, int depth, int q_in,
                                                                int max_depth) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, bv, false, stack_mem, scene_mem, share_mem, ws, lws, share);
    stage_shading<CFG>(sc, shade_mem);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    uint32_t fill_next = b.n_ext[q_in ^ 1][wave], n_done = 0;
    const uint32_t fill0 = fill_next, own_glossy = b.n_glossy[wave];
    const bool deeper = depth + 1 < max_depth;
    BlockFeed feed;
    feed_begin(feed, b, b.n_shadow, b.claim + PIPE_WAVES, PROBE_POOL_SHARE);
    for (;;) {
        // foreign blocks only while the next extend region can take their paths on top of this wave's own glossy probes
        uint32_t src, src_total, block;
        if (!feed_next(feed, b, fill_next + own_glossy + 64u * STEAL_BLOCKS <= b.region_cap, src, src_total, block)) break;
        const bool live = block * 64u + lane < src_total;
        const size_t i = (size_t)src * b.region_cap + (live ? block * 64u + lane : 0u);
        n_done += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        ProbeHead h;
        load_probe_head(b, i, h);
        const bool blocked = shadow_any_hit_cfg<CFG>(sc, bv, h.probe, h.max_t, h.current, live, ws, share, bc);
        PathState ps;
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (live) go_on = retire_shadow(sc, b, i, h, blocked, deeper, ps, kept_flags);
        route_next(b, q_in ^ 1, fill_next, live, go_on, ps, h.id, kept_flags);
    }
    if (depth == 0) wstat_add(b, WSTAT_DEPTH1, fill_next - fill0);
    if (lane == 0) {
        b.n_ext[q_in ^ 1][wave] = fill_next;
        b.claim[2 * PIPE_WAVES + wave] = 0;      // the glossy stage's counters: used by the launch after this one
    }
    wstat_add(b, WSTAT_SHADOW, n_done);
    if (CFG::BVH) {
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(bc.tris));
    }
}

// Glossy-probe stage: search of sampleLightGlossy (raytracer.glsl:403-437), then :557-579.
template <class CFG>
__global__ __launch_bounds__(64 * WG_WAVES, CFG::BVH ? RT_MIN_WAVES : 1) void k_probe_glossy(DScene sc, DBvh bv, DBvh lbv, DFrame fr, PipeBuffers b,
                                                                int depth, int q_in, int max_depth) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t lane = lane_index(), wave = wave_index();
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, lbv, true, stack_mem, scene_mem, share_mem, ws, lws, share);
    stage_shading<CFG>(sc, shade_mem);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    uint32_t fill_next = b.n_ext[q_in ^ 1][wave], n_done = 0;
    const uint32_t fill0 = fill_next;
    const bool deeper = depth + 1 < max_depth;
    BlockFeed feed;
    feed_begin(feed, b, b.n_glossy, b.claim + 2 * PIPE_WAVES, PROBE_POOL_SHARE);
    for (;;) {
        uint32_t src, src_total, block;
        if (!feed_next(feed, b, fill_next + 64u * STEAL_BLOCKS <= b.region_cap, src, src_total, block)) break;
        const bool live = block * 64u + lane < src_total;
        // glossy records fill a region from its back
        const size_t i = (size_t)src * b.region_cap + (b.region_cap - 1u) - (live ? block * 64u + lane : 0u);
        n_done += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        ProbeHead h;
        load_probe_head(b, i, h);
        int light_id;
        f4 lsec;
        const int outcome = glossy_search_cfg<CFG>(sc, bv, lbv, h.probe, h.current, live, light_id, lsec, ws, lws, share, bc);
        PathState ps;
        uint32_t kept_flags = 0;
        bool go_on = false;
        if (live) go_on = retire_glossy(sc, fr, b, i, h, outcome, light_id, lsec, deeper, ps, kept_flags);
        route_next(b, q_in ^ 1, fill_next, live, go_on, ps, h.id, kept_flags);
    }
    if (depth == 0) wstat_add(b, WSTAT_DEPTH1, fill_next - fill0);
    if (lane == 0) {
        b.n_ext[q_in ^ 1][wave] = fill_next;
        // the counters of the extend and of the shadow stage: used before this launch, and by the next ones
        b.claim[wave] = 0;
        b.claim[PIPE_WAVES + wave] = 0;
    }
    wstat_add(b, WSTAT_GLOSSY, n_done);
    if (CFG::BVH) {
        wstat_add(b, WSTAT_PROBE_NODES, wave_sum_u32(bc.nodes));
        wstat_add(b, WSTAT_PROBE_TRIS, wave_sum_u32(bc.tris));
    }
}

// ---------------------------------------------------------------------------------------------------
// Extend stage with lane refill.  Rays that start on surfaces are incoherent: in a wave that takes 64
// queue entries and waits for the slowest, under a third of the lanes do useful work in the node loop of a
// large scene.  Here a wave keeps its lanes busy instead: as soon as REFILL_MIN lanes have finished their ray,
// their results are retired and the free lanes take the next records of the wave's queue region while the other
// lanes carry on where they were (bvh_advance keeps per-lane traversal state).  A path's arithmetic does
// not depend on which lane or in which order it is processed, and the order of a queue region only
// decides the order of the next region, so results stay bit-identical.
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 32
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

// The closest-hit search alone; it leaves the hit record h0 = (u, v, t, triangle id; -1 = miss) beside each path
// record and k_shade does the shading with full waves.
template <class CFG>
__global__ __launch_bounds__(64 * WG_WAVES, CFG::BVH ? RT_MIN_WAVES : 1) void k_extend_refill(DScene sc, DBvh bv, PipeBuffers b, int q_in) {
    if (round_is_idle(b.work, b.round)) return;
    const uint32_t wave = wave_index();
    RT_TRAVERSAL_LDS(CFG);
    typename CFG::WS ws, lws;
    lds_u8_ptr share;
    stage_hierarchies<CFG>(bv, bv, false, stack_mem, scene_mem, share_mem, ws, lws, share);
    BvhCounters bc;
    bc.nodes = 0; bc.tris = 0;
    const uint32_t total = b.n_ext[q_in][wave];
    const size_t region = (size_t)wave * b.region_cap;
    uint32_t next = 0;
    size_t i = region;
    bool busy = false;
    TravState ts;
    trav_begin(ws, ts, false);
    Ray ray;
    ray.o = ray.d = mk3(0.0f, 0.0f, 0.0f);
    BvhRay br = bvh_ray_setup<CFG::RES>(ray, bv, ws);
    ClosestFold cf;
    closest_fold_reset(cf, 1.0f);
    for (;;) {
        uint32_t k = 0;
        if (refill_take(total, next, busy, k)) {
            i = region + k;
            const float4 a = b.xa[q_in][i], c = b.xb[q_in][i];
            ray.o = mk3(a.x, a.y, a.z);
            ray.d = mk3(a.w, c.x, c.y);
            br = bvh_ray_setup<CFG::RES>(ray, bv, ws);
            closest_fold_reset(cf, 1.0f);
            busy = true;
            ts.node = ws.root; ts.sp = 1; ts.run = br.finite;
        }
        if (__builtin_amdgcn_ballot_w64(busy) == 0ull) break;
        auto leaf = [&](int slot, const Tri &t, int id_ahead) -> bool { closest_fold_leaf(bv, ws, ray, slot, t, id_ahead, -1, cf); return true; };
        auto limit = [&]() -> float { return cf.t_min * BVH_NEAR; };
        bvh_advance<false, REFILL_LOOP_SHARE>(bv, br, ws, ts, next < total ? REFILL_MIN : 65, leaf, limit, bc);
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

// Resolve stage: sum[pixel] += radiance of the batch's samples, ascending slot order.  SKIP: samples whose radiance is
// not finite are counted and left out (option "skip_nonfinite"; the reference adds them, which poisons the pixel).
// A wave takes one 8x8 tile, thread = position in the radiance plane: per slot it reads 768 consecutive bytes (threads
// in raster order read eight pieces of 96 bytes, each straddling cache lines that other waves fetch again).
template <bool SKIP>
__global__ __launch_bounds__(256) void k_resolve(PipeBuffers b, float4 *__restrict__ sum, uint32_t width,
                                                 uint32_t height, uint32_t paths_per_sample, uint32_t n_slots,
                                                 DStats *__restrict__ stats) {
    const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tiles_x = (width + 7u) >> 3, tiles_y = (height + 7u) >> 3;
    uint32_t tile_x, tile_y;
    tile_position(local >> 6, tiles_x, tiles_y, tile_x, tile_y);
    const uint32_t tx = tile_x * 8u + (threadIdx.x & 7u), ty = tile_y * 8u + ((threadIdx.x >> 3) & 7u);
    const bool in_image = local < paths_per_sample && tx < width && ty < height;
    uint32_t skipped = 0;
    if (in_image) {
        const uint32_t pix = ty * width + tx;
        float4 s = sum[pix];
        for (uint32_t k = 0; k < n_slots; k++) {
            const Radiance e = b.energy[(size_t)k * paths_per_sample + local];
            if (SKIP && nonfinite3(e.x, e.y, e.z)) { skipped++; continue; }
            s.x += e.x; s.y += e.y; s.z += e.z;
        }
        sum[pix] = s;
    }
    if (SKIP) {
        const unsigned long long n = wave_sum_u32(skipped);
        if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&stats->skipped, n);
    }
}

// What a pass may use, decided by the caller from the scene and the options.
struct PassForm {
    bool bvh;            // hierarchies are there and wanted
    bool resident;       // ... and fit the LDS of a workgroup (KernelCfg::RES)
    bool share;          // shared any-hit searches
    bool refill;         // lane-refill form of the extend stage
    int staged_depths;   // bounces after the first that run as staged kernels before the tail kernel takes the rest; < 0: all
    DTileLists lists;    // candidate lists of the camera rays (k_build_tile_lists), or not enabled
};

// pcgHash (raytracer.glsl:89-94) on the host, for the tables of a pass.
inline uint32_t pcg_hash_on_host(uint32_t k) {
    const uint32_t state = k * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// Host side of the pipeline: buffers sized for one batch, and the launch sequence.
struct Pipeline {
    PipeBuffers buf = {};
    size_t cap_paths = 0;
    bool have_h0 = false;
    uint32_t rounds = 1;           // the queues are sized for 1/rounds of cap_paths (see MAX_ROUNDS)
    // Profiling: events are recorded around every stage launch and turned into times lazily
    // (collect_times), so that the timed region carries no extra synchronisation.
    std::vector<hipEvent_t> events;
    std::vector<int> kinds;        // interval i = [event i, event i+1]: 0 extend, 3 extend at depth 0, 1 probes, 2 shade, -1 not counted (4-6: see collect_times)
    size_t ev_used = 0;

    std::vector<void **> path_buffers() {
        return {(void **)&buf.xa[0], (void **)&buf.xa[1], (void **)&buf.xb[0], (void **)&buf.xb[1], (void **)&buf.xc[0],
                (void **)&buf.xc[1], (void **)&buf.xd[0], (void **)&buf.xd[1], (void **)&buf.pa, (void **)&buf.pb,
                (void **)&buf.pc, (void **)&buf.pd, (void **)&buf.p0, (void **)&buf.p1, (void **)&buf.p2, (void **)&buf.h0,
                (void **)&buf.energy, (void **)&buf.kd, (void **)&buf.ks};
    }
    std::vector<void **> counter_buffers() {
        return {(void **)&buf.wstats, (void **)&buf.n_ext[0], (void **)&buf.n_ext[1], (void **)&buf.n_shadow, (void **)&buf.n_glossy,
                (void **)&buf.work, (void **)&buf.resume, (void **)&buf.claim, (void **)&buf.pool_map};
    }

    void free_paths() {
        for (void **p : path_buffers()) { (void)hipFree(*p); *p = nullptr; }
        cap_paths = 0;
        have_h0 = false;
    }

    void release() {
        free_paths();
        for (void **p : counter_buffers()) { (void)hipFree(*p); *p = nullptr; }
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        buf = PipeBuffers{};
    }

    // From depth 1 on a wave can never emit more entries than it consumed.  The depth-0 kernel hands out its chunks
    // as waves become free and a wave stops taking chunks when its region cannot hold another one whole, so with one
    // chunk of slack per region the regions that still have room can always hold what is left: a region holds
    // (ceil(ceil(chunks / PIPE_WAVES) / rounds) + 1) * 64 entries, and `rounds` rounds of the launch sequence get
    // through the pass even if every path survives (each round a wave renders at least ceil(share / rounds) + 1 chunks).
    static uint32_t region_cap_for(size_t paths, uint32_t rounds) {
        const size_t chunks = (paths + 63) / 64;
        const size_t share = (chunks + PIPE_WAVES - 1) / PIPE_WAVES;
        return (uint32_t)(((share + rounds - 1) / rounds + 1) * 64);
    }
    // Device memory of the queues of a pass of `paths` pixel-samples.
    static size_t queue_bytes_for(size_t paths, bool with_h0, uint32_t rounds) {
        const size_t qn = (size_t)region_cap_for(paths, rounds) * PIPE_WAVES;
        return qn * (2 * 64 + 112 + (with_h0 ? 16 : 0)) + paths * (sizeof(Radiance) + 32);
    }
    size_t queue_bytes() const { return cap_paths ? queue_bytes_for(cap_paths, have_h0, rounds) : 0; }

    hipError_t ensure(size_t paths, bool need_h0, uint32_t want_rounds, const char **what) {
        hipError_t e = hipSuccess;
#define PIPE_ALLOC(ptr, bytes) do { *what = #ptr; e = hipMalloc((void **)&(ptr), (bytes)); if (e != hipSuccess) goto failed; } while (0)
        if (!buf.wstats) {
            PIPE_ALLOC(buf.wstats, sizeof(unsigned long long) * WSTAT_SLOTS * PIPE_WAVES);
            *what = "memset wstats";
            e = hipMemset(buf.wstats, 0, sizeof(unsigned long long) * WSTAT_SLOTS * PIPE_WAVES);
            if (e != hipSuccess) goto failed;
            PIPE_ALLOC(buf.n_ext[0], sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_ext[1], sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_shadow, sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.n_glossy, sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.work, sizeof(uint32_t) * (N_RUNS + MAX_ROUNDS + 1));
            PIPE_ALLOC(buf.resume, sizeof(uint32_t) * PIPE_WAVES);
            PIPE_ALLOC(buf.claim, sizeof(uint32_t) * 3 * PIPE_WAVES);
            *what = "memset claim";
            e = hipMemset(buf.claim, 0, sizeof(uint32_t) * 3 * PIPE_WAVES);
            if (e != hipSuccess) goto failed;
            PIPE_ALLOC(buf.pool_map, sizeof(unsigned long long) * (PIPE_WAVES / 64));
            *what = "memset pool_map";
            e = hipMemset(buf.pool_map, 0, sizeof(unsigned long long) * (PIPE_WAVES / 64));
            if (e != hipSuccess) goto failed;
            *what = "memset resume";
            e = hipMemset(buf.resume, 0, sizeof(uint32_t) * PIPE_WAVES);
            if (e != hipSuccess) goto failed;
        }
        if (paths > cap_paths || (need_h0 && !have_h0) || want_rounds != rounds) {
            const size_t want = (paths > cap_paths || want_rounds != rounds) ? paths : cap_paths;
            free_paths();
            rounds = want_rounds;
            const size_t qn = (size_t)region_cap_for(want, rounds) * PIPE_WAVES;      // queue planes hold PIPE_WAVES regions
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
            if (need_h0) PIPE_ALLOC(buf.h0, sizeof(float4) * qn);
            PIPE_ALLOC(buf.energy, sizeof(Radiance) * want);
            PIPE_ALLOC(buf.kd, sizeof(float4) * want);
            PIPE_ALLOC(buf.ks, sizeof(float4) * want);
            cap_paths = want;
            have_h0 = need_h0;
        }
#undef PIPE_ALLOC
        return hipSuccess;
    failed:
        // leave nothing half-allocated behind: the next call starts from scratch
        free_paths();
        for (void **p : counter_buffers()) { (void)hipFree(*p); *p = nullptr; }
        return e;
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
                host_stats->trace_ms += ms;
                // 4, 5, 6: the launches of kinds 3, 0, 1 in the rounds of a pass after the first
                if (kinds[i] == 0 || kinds[i] == 3 || kinds[i] == 4 || kinds[i] == 5) {
                    host_stats->extend_ms += ms;
                    if (kinds[i] == 0 || kinds[i] == 3) host_stats->extend_launches++;
                    if (kinds[i] == 3 || kinds[i] == 4) host_stats->extend0_ms += ms;
                    if (kinds[i] == 3) host_stats->extend0_launches++;
                } else if (kinds[i] == 2) {
                    host_stats->shade_ms += ms;
                } else {
                    host_stats->probe_ms += ms;
                    if (kinds[i] == 1) host_stats->probe_launches += 2;
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

    // Renders the samples listed in `tab` and adds them to `sum`.  Enqueues kernels only (with `profiling`, events as
    // well); the buffers must have been sized by ensure().
    template <class CFG>
    hipError_t launch_pass(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum,
                           const PassTables &tab, hipStream_t stream, bool profiling, const PassForm &form, rt_stats *host_stats,
                           DStats *dstats, const char **what);
    hipError_t run(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum, const PassTables &tab,
                   hipStream_t stream, bool profiling, const PassForm &form, rt_stats *host_stats, DStats *dstats, const char **what);
};

template <class CFG>
inline hipError_t Pipeline::launch_pass(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum,
                                        const PassTables &tab, hipStream_t stream, bool profiling, const PassForm &form,
                                        rt_stats *host_stats, DStats *dstats, const char **what) {
    const int max_depth = fr.recursion > 1 ? fr.recursion : 1;
    const uint32_t tiles = ((fr.width + 7u) >> 3) * ((fr.height + 7u) >> 3);
    const uint32_t pps = tiles * 64u;
    hipError_t e = hipSuccess;
#define PIPE_TRY(expr) do { *what = #expr; e = (expr); if (e != hipSuccess) return e; } while (0)
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
    // depth 0 runs its probes in place (k_first): queue 0 is never read and the first queue is 1
    PIPE_TRY(hipMemsetAsync(buf.work, 0, sizeof(uint32_t) * (N_RUNS + MAX_ROUNDS + 1), stream));
    const int staged_end = form.staged_depths < 0 ? max_depth : (1 + form.staged_depths < max_depth ? 1 + form.staged_depths : max_depth);
    // light seeds of the pass's samples as they stand at each staged depth and where the tail kernel starts: every bounce
    // draws once from the stream all pixels of a sample share (raytracer.glsl:553), so the table a launch gets is the
    // one before hashed once more, and no kernel has to redo the draws of the earlier bounces
    std::vector<PassTables> tab_at((size_t)staged_end + 1, tab);
    for (int depth = 1; depth <= staged_end; depth++)
        for (uint32_t k = 0; k < tab.n; k++) tab_at[depth].light_seed[k] = pcg_hash_on_host(tab_at[depth - 1].light_seed[k]);
    for (uint32_t round = 0; round < rounds; round++) {
        // launches of the rounds after the first are timed with the others but not counted as launches (most return at once)
        buf.round = round;
        buf.rounds = rounds;
        buf.defer_below = (CFG::BVH && sc.count < DEFER_MAX_TRIANGLES) ? DEFER_BELOW : 0u;
        buf.dynamic_share = (CFG::BVH && sc.count >= LARGE_SCENE_TRIANGLES) ? DYNAMIC_SHARE_LARGE : DYNAMIC_SHARE;
        buf.dynamic_group = (CFG::BVH && sc.count >= LARGE_SCENE_TRIANGLES) ? 1u : DYNAMIC_GROUP;
        PIPE_TRY(stamp(round == 0 ? 3 : 4));
        if (round == 0) hipLaunchKernelGGL((k_first<CFG, false>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, tab, pps, max_depth, form.lists);
        else hipLaunchKernelGGL((k_first<CFG, true>), grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, tab, pps, max_depth, form.lists);
        PIPE_TRY(hipGetLastError());
        if (buf.defer_below) {
            // the probes that k_first left to the probe stages (chunks with few of them); survivors join queue 1
            PIPE_TRY(stamp(round == 0 ? 1 : 6));
            hipLaunchKernelGGL(k_probe_shadow<CFG>, grid, block, 0, stream, sc, bvh, buf, 0, 0, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_probe_glossy<CFG>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, 0, 0, max_depth);
            PIPE_TRY(hipGetLastError());
        }
        PIPE_TRY(stamp(-1));
        int q_in = 1;
        for (int depth = 1; depth < staged_end; depth++) {
            PIPE_TRY(stamp(round == 0 ? 0 : 5));
            if (CFG::BVH && form.refill) {
                hipLaunchKernelGGL(k_extend_refill<CFG>, grid, block, 0, stream, sc, bvh, buf, q_in);
                PIPE_TRY(hipGetLastError());
                PIPE_TRY(stamp(2));
                hipLaunchKernelGGL(k_shade, grid, block, 0, stream, sc, fr, buf, depth, q_in, tab_at[depth], pps, max_depth);
            } else {
                hipLaunchKernelGGL(k_extend<CFG>, grid, block, 0, stream, sc, bvh, fr, buf, depth, q_in, tab_at[depth], pps, max_depth);
            }
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(round == 0 ? 1 : 6));
            hipLaunchKernelGGL(k_probe_shadow<CFG>, grid, block, 0, stream, sc, bvh, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_probe_glossy<CFG>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, depth, q_in, max_depth);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(-1));
            q_in ^= 1;
        }
        if (staged_end < max_depth) {
            PIPE_TRY(stamp(round == 0 ? 0 : 5));
            hipLaunchKernelGGL(k_tail<CFG>, grid, block, 0, stream, sc, bvh, light_bvh, fr, buf, staged_end, q_in, tab_at[staged_end], pps, max_depth);
            PIPE_TRY(hipGetLastError());
            PIPE_TRY(stamp(-1));
        }
    }
    if (fr.skip_nonfinite)
        hipLaunchKernelGGL(k_resolve<true>, dim3((pps + 255u) / 256u), dim3(256), 0, stream, buf, sum, fr.width,
                           fr.height, pps, tab.n, dstats);
    else
        hipLaunchKernelGGL(k_resolve<false>, dim3((pps + 255u) / 256u), dim3(256), 0, stream, buf, sum, fr.width,
                           fr.height, pps, tab.n, dstats);
    PIPE_TRY(hipGetLastError());
#undef PIPE_TRY
    return hipSuccess;
}

inline hipError_t Pipeline::run(const DScene &sc, const DBvh &bvh, const DBvh &light_bvh, const DFrame &fr, float4 *sum,
                                const PassTables &tab, hipStream_t stream, bool profiling, const PassForm &form,
                                rt_stats *host_stats, DStats *dstats, const char **what) {
    const uint32_t tiles = ((fr.width + 7u) >> 3) * ((fr.height + 7u) >> 3);
    const size_t paths = (size_t)tiles * 64u * (size_t)tab.n;
    if (paths > 0xFFFFFFF0ull) { *what = "batch has more than 2^32 paths"; return hipErrorInvalidValue; }
    if (paths > cap_paths || (form.bvh && form.refill && !have_h0)) { *what = "queues not sized for this pass"; return hipErrorInvalidValue; }
    buf.region_cap = region_cap_for(cap_paths, rounds);
#define RUN_AS(B, S, R) return launch_pass<KernelCfg<B, S, R>>(sc, bvh, light_bvh, fr, sum, tab, stream, profiling, form, host_stats, dstats, what)
    if (!form.bvh) RUN_AS(false, 0, false);
    if (form.resident) { if (form.share) RUN_AS(true, 1, true); RUN_AS(true, 0, true); }
    if (!form.share) RUN_AS(true, 0, false);
    // shared any-hit searches: with a node budget per round where traversals are long
    if (sc.count >= SHARE_BUDGET_MIN_TRIANGLES) RUN_AS(true, 2, false);
    RUN_AS(true, 1, false);
#undef RUN_AS
}

}  // namespace rt
