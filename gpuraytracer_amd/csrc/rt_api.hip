// rt_api.hip - C ABI of libgpurt.so (declared in include/gpurt.h) and the launch logic.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero -shared -fPIC
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/gpurt.h"
#include "rt_kernels.h"
#include "rt_pipeline.h"
#include "rt_bvh_build.h"
#include "rt_irradiance.h"

using namespace rt;

struct rt_context {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    // scene
    float4 *d_geo = nullptr, *d_shade = nullptr, *d_mats = nullptr;
    int *d_has_tex = nullptr;
    uint32_t **d_atlas_ptrs = nullptr;
    std::vector<uint32_t *> atlas_layers;
    float4 *d_env = nullptr;
    int env_w = 0, env_h = 0;
    int count = 0, lights = 0, n_mat = 0, atlas_size = 4096;
    bool have_scene = false;

    // hierarchies (all triangles; light triangles only), see rt_bvh.h
    struct DevBvh {
        float4 *nodes = nullptr, *leaf_geo = nullptr;
        uint4 *qnodes = nullptr;
        float qbase[3] = {0.0f, 0.0f, 0.0f}, qscale[3] = {1.0f, 1.0f, 1.0f};
        int *leaf_ids = nullptr;
        float max_abs = 0.0f;
        int n_nodes = 0, n_slots = 0, depth = 0;
        bool ok = false;
        bool tiny_normals = false;     // HostBvh::min_normal_len2 < TINY_NORMAL_LEN2
    };
    DevBvh bvh, light_bvh;
    int accel = -1;            // -1 auto (hierarchy from ACCEL_AUTO_MIN triangles on), 0 brute force, 1 hierarchy

    // the selected target (rt_select_target); the other one waits in `parked`
    uint32_t width = 0, height = 0;
    float4 *d_sum_own = nullptr, *d_sum = nullptr;
    uint64_t n_samples = 0;
    struct Target {
        uint32_t width = 0, height = 0;
        float4 *d_sum_own = nullptr, *d_sum = nullptr;
        uint64_t n_samples = 0;
    };
    Target parked;
    int target = 0;            // 0 = the still image, 1 = the low-resolution preview

    // frame
    DFrame frame;
    bool have_frame = false;

    uint32_t chain_pos = 0, chain_val = 0;   // light-seed chain: chain_val = hash^chain_pos(clock)
    bool chain_valid = false;

    // options
    int pipeline = 1;
    int batch_samples = 32;
    int queue_rounds = 3;      // the path queues hold 1/queue_rounds of a pass's paths (Pipeline::rounds)
    int profiling = 0;
    int share = 1;             // lanes of a wave share the any-hit searches of its probes (anyhit_shared)
    int staged_depths = 2;     // bounces after the first that run as staged kernels before the tail kernel takes the rest; -1: all
    int resident = -1;         // -1 auto (hierarchies that fit are staged into LDS whole), 0 never
    int tonemap = 0;           // rt_read_srgb8: 0 sRGB transfer (displayQuad.fs), 1 ACES film curve first (Scene::exportBMP)
    int refill = -1;           // -1 auto (lane refill from REFILL_AUTO_MIN triangles on), 0 off, 1 on
    int skip_nonfinite = 0;    // 1: samples whose radiance is NaN or infinite are left out of the sums and counted

    // candidate lists of the camera rays (k_build_tile_lists): one per 8x8 tile of the selected target, for the frame set last
    uint32_t *d_tile_lists = nullptr;
    size_t tile_lists_cap = 0;     // tiles the buffer holds
    bool tile_lists_valid = false;
    int tile_lists = -1;           // option: -1 auto (scenes of up to TILE_LIST_MAX_TRIANGLES triangles), 0 never

    // stats
    DStats *d_stats = nullptr;
    rt_stats host_stats;
    uint64_t staged_paths_expected = 0;   // camera paths the staged passes since the last reset were asked to render
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    Pipeline pipe;
};

namespace {

int fail(rt_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

int bind_device(rt_context *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

DScene device_scene(const rt_context *ctx) {
    DScene sc;
    sc.geo = ctx->d_geo;
    sc.shade = ctx->d_shade;
    sc.shade03 = ctx->d_shade;
    sc.shade_lds = 0;
    sc.mats_lds = 0;
    sc.light_geo_lds = 0;
    sc.mats = ctx->d_mats;
    sc.has_tex = ctx->d_has_tex;
    sc.atlas = ctx->d_atlas_ptrs;
    sc.env = ctx->d_env;
    sc.count = ctx->count;
    sc.lights = ctx->lights;
    sc.n_mat = ctx->n_mat;
    sc.atlas_size = ctx->atlas_size;
    sc.atlas_layers = (int)ctx->atlas_layers.size();
    sc.env_w = ctx->env_w;
    sc.env_h = ctx->env_h;
    return sc;
}

constexpr int ACCEL_AUTO_MIN = 64;
// Lane refill pays once traversals are long: +18 % on a 1 M-triangle scene, -14 % on a 424-triangle one.
constexpr int REFILL_AUTO_MIN = 65536;

DBvh device_bvh(const rt_context *ctx, const rt_context::DevBvh &b, bool want) {
    DBvh d;
    d.nodes = b.nodes;
    d.qnodes = b.qnodes;
    for (int a = 0; a < 3; a++) { d.qbase[a] = b.qbase[a]; d.qscale[a] = b.qscale[a]; }
    d.leaf_geo = b.leaf_geo;
    d.leaf_ids = b.leaf_ids;
    d.max_abs_coord = b.max_abs;
    d.enabled = (want && b.ok) ? 1 : 0;
    d.n_nodes = b.n_nodes;
    d.n_slots = b.n_slots;
    // a traversal pushes at most one child per level
    const int deepest = std::max(ctx->bvh.depth, ctx->light_bvh.ok ? ctx->light_bvh.depth : 0);
    d.stack_entries = std::min(BVH_STACK, std::max(8, deepest + 2));
    return d;
}

bool use_accel(const rt_context *ctx) {
    if (!ctx->bvh.ok) return false;
    // automatic choice: scenes too small to gain, and scenes whose depth comparisons can underflow, keep the loops
    if (ctx->accel < 0) return ctx->count >= ACCEL_AUTO_MIN && !ctx->bvh.tiny_normals;
    return ctx->accel != 0;
}

// Do both hierarchies fit the LDS of a workgroup (KernelCfg::RES)?
bool fits_lds(const rt_context *ctx) {
    const rt_context::DevBvh &a = ctx->bvh, &l = ctx->light_bvh;
    const int bytes = RES_NODE_BYTES * (a.n_nodes + (l.ok ? l.n_nodes : 0)) + 32 + 64 * (a.n_slots + (l.ok ? l.n_slots : 0));
    return bytes <= RES_BYTES && a.depth < RES_STACK && (!l.ok || l.depth < RES_STACK) && shading_fits_lds((int)ctx->count, (int)ctx->n_mat);
}

void free_bvh(rt_context::DevBvh &b) {
    if (b.nodes) (void)hipFree(b.nodes);
    if (b.qnodes) (void)hipFree(b.qnodes);
    if (b.leaf_geo) (void)hipFree(b.leaf_geo);
    if (b.leaf_ids) (void)hipFree(b.leaf_ids);
    b = rt_context::DevBvh();
}

hipError_t upload_bvh(const rt_triangle_model *tris, uint32_t count, rt_context::DevBvh &out) {
    HostBvh hb;
    build_bvh(tris, count, hb);
    hipError_t e = hipMalloc(&out.nodes, hb.nodes.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&out.qnodes, hb.qnodes.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(out.qnodes, hb.qnodes.data(), hb.qnodes.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    for (int a = 0; a < 3; a++) { out.qbase[a] = hb.qbase[a]; out.qscale[a] = hb.qscale[a]; }
    if (e == hipSuccess) e = hipMalloc(&out.leaf_geo, std::max<size_t>(1, hb.leaf_geo.size()) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&out.leaf_ids, std::max<size_t>(1, hb.leaf_ids.size()) * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(out.nodes, hb.nodes.data(), hb.nodes.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && !hb.leaf_geo.empty())
        e = hipMemcpy(out.leaf_geo, hb.leaf_geo.data(), hb.leaf_geo.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && !hb.leaf_ids.empty())
        e = hipMemcpy(out.leaf_ids, hb.leaf_ids.data(), hb.leaf_ids.size() * sizeof(int), hipMemcpyHostToDevice);
    out.max_abs = hb.max_abs_coord;
    out.n_nodes = (int)(hb.nodes.size() / 16);
    out.n_slots = (int)hb.leaf_ids.size();
    out.depth = hb.depth;
    out.tiny_normals = hb.min_normal_len2 < TINY_NORMAL_LEN2;
    out.ok = e == hipSuccess;
    return e;
}

// light_seed of a sample = CLOCK hashed SAMPLE times (raytracer.glsl:595-598); the chain is
// continued from the last position when samples ascend.
uint32_t light_seed_for(rt_context *ctx, uint32_t sample) {
    if (!ctx->chain_valid || sample < ctx->chain_pos) {
        ctx->chain_pos = 0;
        ctx->chain_val = ctx->frame.clock;
        ctx->chain_valid = true;
    }
    while (ctx->chain_pos < sample) {
        ctx->chain_val = pcg_hash_on_host(ctx->chain_val);
        ctx->chain_pos++;
    }
    return ctx->chain_val;
}

void free_atlas(rt_context *ctx) {
    for (uint32_t *p : ctx->atlas_layers)
        if (p) (void)hipFree(p);
    ctx->atlas_layers.clear();
    if (ctx->d_atlas_ptrs) (void)hipFree(ctx->d_atlas_ptrs);
    ctx->d_atlas_ptrs = nullptr;
}

void free_scene(rt_context *ctx) {
    if (ctx->d_geo) (void)hipFree(ctx->d_geo);
    if (ctx->d_shade) (void)hipFree(ctx->d_shade);
    if (ctx->d_mats) (void)hipFree(ctx->d_mats);
    if (ctx->d_has_tex) (void)hipFree(ctx->d_has_tex);
    ctx->d_geo = ctx->d_shade = ctx->d_mats = nullptr;
    ctx->d_has_tex = nullptr;
    free_atlas(ctx);
    free_bvh(ctx->bvh);
    free_bvh(ctx->light_bvh);
    ctx->have_scene = false;
}

// RGBA32F -> packed R11F_G11F_B10F, the conversion of the texture upload at Scene.cpp:93-98.  OpenGL lets
// the implementation pick either neighbouring value; Mesa: negative -> 0, NaN -> NaN, +Inf -> Inf, above
// the largest finite value -> the largest finite value, below 2^-14 -> 0, otherwise the mantissa is truncated.
__device__ uint32_t encode_ufloat(float f, int mbits) {
    const uint32_t bits = __float_as_uint(f);
    const uint32_t neg = bits >> 31, ex = (bits >> 23) & 0xFFu, man = bits & 0x7FFFFFu;
    const uint32_t top = (30u << mbits) | ((1u << mbits) - 1u);
    if (ex == 255u) return man ? ((31u << mbits) | 1u) : (neg ? 0u : (31u << mbits));
    if (neg) return 0u;
    if (f > (mbits == 6 ? 65024.0f : 64512.0f)) return top;
    const int e = (int)ex - 127;
    if (e > -15) return ((uint32_t)(e + 15) << mbits) | (man >> (23 - mbits));
    return 0u;
}

__global__ void k_pack_atlas(const float4 *__restrict__ src, uint32_t w, uint32_t h, uint32_t *__restrict__ dst,
                             uint32_t size) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float4 p = src[(size_t)y * w + x];
    dst[(size_t)y * size + x] = encode_ufloat(p.x, 6) | (encode_ufloat(p.y, 6) << 11) | (encode_ufloat(p.z, 5) << 22);
}

__global__ void k_add_sums(float4 *__restrict__ sum, const float4 *__restrict__ other, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = sum[i];
    const float4 o = other[i];
    s.x += o.x; s.y += o.y; s.z += o.z;
    sum[i] = s;
}

__global__ void k_sum_to_rgb(const float4 *__restrict__ sum, unsigned long long n, float *__restrict__ rgb) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s = sum[i];
    rgb[i * 3 + 0] = s.x;
    rgb[i * 3 + 1] = s.y;
    rgb[i * 3 + 2] = s.z;
}

// ---- staging buffer of rt_read_sum_static: device memory that belongs to the loaded code object, so the entry point
// never allocates.  32 MiB = 2 796 202 pixels of three floats.  Each GPU the library runs on holds its own copy of the
// array at its own address until the process ends; in host code the array's name is only a handle for
// hipGetSymbolAddress, the address that call returns (per device) is what kernels and copies are given.
constexpr size_t STATIC_FLOATS = (size_t)8 << 20;
constexpr uint64_t STATIC_MAX_PIXELS = STATIC_FLOATS / 3;
static __device__ __align__(256) float scratch[STATIC_FLOATS];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
float *g_scratch_ptr[MAX_DEVICES] = {};       // device address of `scratch` on each GPU, looked up once per device

// Looks up this device's address of `scratch` (the device must be current).  Called from rt_create.
void lookup_scratch(int device) {
    if (device < 0 || device >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return;
    float *ptr = nullptr;
    if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) == hipSuccess) g_scratch_ptr[device] = ptr;
    else (void)hipGetLastError();             // rt_read_sum_static reports the missing buffer; everything else works
}

}  // namespace

extern "C" {

int rt_create(int device, rt_context **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RT_ERR_DEVICE;
    rt_context *ctx = new (std::nothrow) rt_context();
    if (!ctx) return RT_ERR_NOMEM;
    ctx->device = device;
    memset(&ctx->host_stats, 0, sizeof(ctx->host_stats));
    memset(&ctx->frame, 0, sizeof(ctx->frame));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess ||
        hipMalloc(&ctx->d_stats, sizeof(DStats)) != hipSuccess ||
        hipMemset(ctx->d_stats, 0, sizeof(DStats)) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return RT_ERR_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    lookup_scratch(device);
    // The kernels are gfx950 code and their work distribution (PIPE_WAVES persistent waves, N_XCD, one workgroup per
    // compute unit) is laid out for a whole MI355X of 256 compute units.  Another architecture cannot run them; a
    // partition of the chip can, with the same results (no wave waits for another), but not at the documented rates:
    // that is said once, where rt_last_error() finds it.
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        (void)rt_destroy(ctx);
        return RT_ERR_DEVICE;
    }
    if (prop.multiProcessorCount != PIPE_WAVES / RT_WAVES_PER_CU) {
        char note[160];
        snprintf(note, sizeof(note), "note: device %d has %d compute units; the work distribution is laid out for %d (results are the same)",
                 device, prop.multiProcessorCount, PIPE_WAVES / RT_WAVES_PER_CU);
        ctx->err = note;
    }
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_context *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    ctx->pipe.release();
    if (ctx->d_env) (void)hipFree(ctx->d_env);
    if (ctx->d_sum_own) (void)hipFree(ctx->d_sum_own);
    if (ctx->parked.d_sum_own) (void)hipFree(ctx->parked.d_sum_own);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_tile_lists) (void)hipFree(ctx->d_tile_lists);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

const char *rt_last_error(const rt_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rt_set_stream(rt_context *ctx, void *hip_stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RT_OK;
}

int rt_synchronize(rt_context *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_upload_scene(rt_context *ctx, const rt_triangle_model *models, const rt_triangle_shading *shadings,
                    uint32_t count, uint32_t lights, const rt_material *materials, uint32_t n_materials) {
    if (!ctx) return RT_ERR_INVALID;
    if (count == 0 || !models || !shadings || !materials || n_materials == 0)
        return fail(ctx, RT_ERR_INVALID, "rt_upload_scene: empty scene");
    if (lights > count) return fail(ctx, RT_ERR_INVALID, "rt_upload_scene: lights (%u) > count (%u)", lights, count);
    // a leaf of the hierarchy is coded as ~((first slot << 3) | count) in 32 bits, and the kernels index records with int
    if (count >= (1u << 28)) return fail(ctx, RT_ERR_LIMIT, "rt_upload_scene: %u triangles, the limit is 2^28 - 1", count);
    for (uint32_t i = 0; i < count; i++)
        if (shadings[i].material_id >= n_materials)
            return fail(ctx, RT_ERR_INVALID, "rt_upload_scene: triangle %u names material %u of %u", i,
                        shadings[i].material_id, n_materials);
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    ctx->tile_lists_valid = false;

    // geometry: the 48-byte record as is (3 x float4); shading: 7 float4 planes
    std::vector<float4> shade((size_t)7 * count);
    for (uint32_t i = 0; i < count; i++) {
        const rt_triangle_shading &s = shadings[i];
        float idbits;
        memcpy(&idbits, &s.material_id, 4);
        shade[(size_t)0 * count + i] = make_float4(s.normals[0][0], s.normals[0][1], s.normals[0][2], idbits);
        shade[(size_t)1 * count + i] = make_float4(s.normals[1][0], s.normals[1][1], s.normals[1][2], s.tex_p[0]);
        shade[(size_t)2 * count + i] = make_float4(s.normals[2][0], s.normals[2][1], s.normals[2][2], s.tex_p[1]);
        shade[(size_t)3 * count + i] = make_float4(s.tex_u[0], s.tex_u[1], s.tex_v[0], s.tex_v[1]);
        for (int k = 0; k < 3; k++)
            shade[(size_t)(4 + k) * count + i] = make_float4(s.tangents[k][0], s.tangents[k][1], s.tangents[k][2], 0.0f);
    }
    HIP_TRY(ctx, hipMalloc(&ctx->d_geo, sizeof(rt_triangle_model) * (size_t)count));
    HIP_TRY(ctx, hipMalloc(&ctx->d_shade, sizeof(float4) * shade.size()));
    HIP_TRY(ctx, hipMalloc(&ctx->d_mats, sizeof(rt_material) * (size_t)n_materials));
    HIP_TRY(ctx, hipMalloc(&ctx->d_has_tex, sizeof(int) * (size_t)n_materials));
    HIP_TRY(ctx, hipMemcpy(ctx->d_geo, models, sizeof(rt_triangle_model) * (size_t)count, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_shade, shade.data(), sizeof(float4) * shade.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_mats, materials, sizeof(rt_material) * (size_t)n_materials, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->d_has_tex, 0, sizeof(int) * (size_t)n_materials));
    ctx->atlas_layers.assign((size_t)n_materials * 3, nullptr);
    HIP_TRY(ctx, hipMalloc(&ctx->d_atlas_ptrs, sizeof(uint32_t *) * ctx->atlas_layers.size()));
    HIP_TRY(ctx, hipMemset(ctx->d_atlas_ptrs, 0, sizeof(uint32_t *) * ctx->atlas_layers.size()));
    ctx->count = (int)count;
    ctx->lights = (int)lights;
    ctx->n_mat = (int)n_materials;
    // hierarchies need finite geometry; a scene with NaN/Inf coordinates keeps the index-order loops
    bool finite = true;
    for (uint32_t i = 0; i < count && finite; i++) {
        const float *f = (const float *)&models[i];
        for (int k = 0; k < 12; k++) finite = finite && std::isfinite(f[k]);
    }
    if (finite) {
        HIP_TRY(ctx, upload_bvh(models, count, ctx->bvh));
        if (lights > 0) HIP_TRY(ctx, upload_bvh(models, lights, ctx->light_bvh));
    }
    ctx->have_scene = true;
    return RT_OK;
}

int rt_set_atlas_size(rt_context *ctx, uint32_t size) {
    if (!ctx) return RT_ERR_INVALID;
    if (size == 0 || size > 16384) return fail(ctx, RT_ERR_LIMIT, "rt_set_atlas_size: size %u not in [1, 16384]", size);
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t *&p : ctx->atlas_layers) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    if (ctx->d_atlas_ptrs)
        HIP_TRY(ctx, hipMemset(ctx->d_atlas_ptrs, 0, sizeof(uint32_t *) * ctx->atlas_layers.size()));
    ctx->atlas_size = (int)size;
    return RT_OK;
}

int rt_upload_atlas_layer(rt_context *ctx, uint32_t material, uint32_t slot, uint32_t w, uint32_t h,
                          const float *rgba) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_upload_atlas_layer: no scene");
    if (material >= (uint32_t)ctx->n_mat || slot > 2 || !rgba || w == 0 || h == 0)
        return fail(ctx, RT_ERR_INVALID, "rt_upload_atlas_layer: bad material/slot/pixels");
    if (w > (uint32_t)ctx->atlas_size || h > (uint32_t)ctx->atlas_size)
        return fail(ctx, RT_ERR_LIMIT, "rt_upload_atlas_layer: %ux%u exceeds the %d-texel atlas", w, h, ctx->atlas_size);
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const size_t layer = (size_t)material * 3 + slot;
    const size_t texels = (size_t)ctx->atlas_size * ctx->atlas_size;
    if (!ctx->atlas_layers[layer]) HIP_TRY(ctx, hipMalloc(&ctx->atlas_layers[layer], texels * sizeof(uint32_t)));
    uint32_t *dst = ctx->atlas_layers[layer];
    float4 *staging = nullptr;
    HIP_TRY(ctx, hipMalloc(&staging, sizeof(float4) * (size_t)w * h));
    hipError_t e = hipMemcpyAsync(staging, rgba, sizeof(float4) * (size_t)w * h, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dst, 0, texels * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
        dim3 grid((w + 255) / 256, h);
        hipLaunchKernelGGL(k_pack_atlas, grid, dim3(256), 0, ctx->stream, staging, w, h, dst, (uint32_t)ctx->atlas_size);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->d_atlas_ptrs + layer, &ctx->atlas_layers[layer], sizeof(uint32_t *),
                           hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_upload_atlas_layer: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_set_has_texture(rt_context *ctx, const int32_t *flags, uint32_t n) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_set_has_texture: no scene");
    if (!flags || n > (uint32_t)ctx->n_mat) return fail(ctx, RT_ERR_INVALID, "rt_set_has_texture: bad flags");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_has_tex, flags, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_upload_env(rt_context *ctx, uint32_t w, uint32_t h, const float *rgba) {
    if (!ctx) return RT_ERR_INVALID;
    if (!rgba || w == 0 || h == 0 || w > 65536 || h > 65536) return fail(ctx, RT_ERR_INVALID, "rt_upload_env: bad image");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_env) (void)hipFree(ctx->d_env);
    ctx->d_env = nullptr;
    ctx->env_w = ctx->env_h = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_env, sizeof(float4) * (size_t)w * h));
    HIP_TRY(ctx, hipMemcpy(ctx->d_env, rgba, sizeof(float4) * (size_t)w * h, hipMemcpyHostToDevice));
    ctx->env_w = (int)w;
    ctx->env_h = (int)h;
    return RT_OK;
}

int rt_prefilter_irradiance(rt_context *ctx, float *rgb, uint32_t *out_w, uint32_t *out_h) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->d_env) return fail(ctx, RT_ERR_INVALID, "rt_prefilter_irradiance: no environment uploaded");
    const uint32_t w = (uint32_t)ctx->env_w, h = (uint32_t)ctx->env_h, iw = w / 4u, ih = h / 4u;
    if (out_w) *out_w = iw;
    if (out_h) *out_h = ih;
    if (!rgb) return RT_OK;                   // size query
    if (iw == 0 || ih == 0) return RT_OK;     // as in the reference: an image of no texels
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const size_t n = (size_t)iw * ih;
    float4 *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_out, sizeof(float4) * n));
    hipLaunchKernelGGL(k_irradiance, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->d_env, w, h, d_out);
    hipError_t e = hipGetLastError();
    std::vector<float4> host(n);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out, sizeof(float4) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_prefilter_irradiance: %s", hipGetErrorString(e));
    for (size_t i = 0; i < n; i++) { rgb[3 * i] = host[i].x; rgb[3 * i + 1] = host[i].y; rgb[3 * i + 2] = host[i].z; }
    return RT_OK;
}

int rt_resize(rt_context *ctx, uint32_t width, uint32_t height) {
    if (!ctx) return RT_ERR_INVALID;
    if (width == 0 || height == 0 || width > 32768 || height > 32768)
        return fail(ctx, RT_ERR_LIMIT, "rt_resize: %ux%u not in [1, 32768]^2", width, height);
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const bool external = ctx->d_sum && ctx->d_sum != ctx->d_sum_own;
    if (external && (width != ctx->width || height != ctx->height)) {
        ctx->d_sum = nullptr;   // a caller-owned buffer of the old size cannot be kept
    }
    if (!external || !ctx->d_sum) {
        if (ctx->d_sum_own) (void)hipFree(ctx->d_sum_own);
        ctx->d_sum_own = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->d_sum_own, sizeof(float4) * (size_t)width * height));
        ctx->d_sum = ctx->d_sum_own;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sum, 0, sizeof(float4) * (size_t)width * height, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(DStats), ctx->stream));
    HIP_TRY(ctx, ctx->pipe.reset_wstats(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->width = width;
    ctx->height = height;
    ctx->n_samples = 0;
    ctx->tile_lists_valid = false;
    memset(&ctx->host_stats, 0, sizeof(ctx->host_stats));
    ctx->staged_paths_expected = 0;
    return RT_OK;
}

int rt_select_target(rt_context *ctx, int target) {
    if (!ctx) return RT_ERR_INVALID;
    if (target != 0 && target != 1) return fail(ctx, RT_ERR_INVALID, "rt_select_target: target %d is neither 0 (still) nor 1 (preview)", target);
    if (target == ctx->target) return RT_OK;
    // work already enqueued keeps the buffer it was launched with; nothing has to be waited for
    rt_context::Target now;
    now.width = ctx->width; now.height = ctx->height;
    now.d_sum_own = ctx->d_sum_own; now.d_sum = ctx->d_sum;
    now.n_samples = ctx->n_samples;
    ctx->width = ctx->parked.width; ctx->height = ctx->parked.height;
    ctx->d_sum_own = ctx->parked.d_sum_own; ctx->d_sum = ctx->parked.d_sum;
    ctx->n_samples = ctx->parked.n_samples;
    ctx->parked = now;
    ctx->target = target;
    ctx->tile_lists_valid = false;
    return RT_OK;
}

int rt_clear(rt_context *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (ctx->width == 0 || !ctx->d_sum) return fail(ctx, RT_ERR_STATE, "rt_clear: no target (rt_resize)");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sum, 0, sizeof(float4) * (size_t)ctx->width * ctx->height, ctx->stream));
    ctx->n_samples = 0;
    return RT_OK;
}

int rt_set_sum_buffer(rt_context *ctx, float *device_rgba) {
    if (!ctx) return RT_ERR_INVALID;
    if (ctx->width == 0) return fail(ctx, RT_ERR_STATE, "rt_set_sum_buffer: call rt_resize first");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (device_rgba) {
        ctx->d_sum = (float4 *)device_rgba;
    } else {
        if (!ctx->d_sum_own)
            HIP_TRY(ctx, hipMalloc(&ctx->d_sum_own, sizeof(float4) * (size_t)ctx->width * ctx->height));
        ctx->d_sum = ctx->d_sum_own;
    }
    return RT_OK;
}

int rt_accumulate_from(rt_context *ctx, rt_context *other) {
    if (!ctx || !other || ctx == other) return RT_ERR_INVALID;
    if (ctx->width == 0 || !ctx->d_sum || !other->d_sum) return fail(ctx, RT_ERR_STATE, "rt_accumulate_from: no target (rt_resize)");
    if (ctx->width != other->width || ctx->height != other->height)
        return fail(ctx, RT_ERR_INVALID, "rt_accumulate_from: targets differ (%ux%u, %ux%u)", ctx->width, ctx->height,
                    other->width, other->height);
    const unsigned long long n = (unsigned long long)ctx->width * ctx->height;
    if (bind_device(other)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(other->stream));
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const float4 *src = other->d_sum;
    float4 *staging = nullptr;
    if (other->device != ctx->device) {
        HIP_TRY(ctx, hipMalloc(&staging, sizeof(float4) * n));
        const hipError_t e = hipMemcpyPeerAsync(staging, ctx->device, other->d_sum, other->device, sizeof(float4) * n, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(staging); return fail(ctx, RT_ERR_DEVICE, "rt_accumulate_from: %s", hipGetErrorString(e)); }
        src = staging;
    }
    hipLaunchKernelGGL(k_add_sums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sum, src, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (staging) (void)hipFree(staging);
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_accumulate_from: %s", hipGetErrorString(e));
    ctx->n_samples += other->n_samples;
    return RT_OK;
}

int rt_set_frame(rt_context *ctx, const float camera[16], uint32_t clock, int32_t recursion, float exposure) {
    if (!ctx || !camera) return RT_ERR_INVALID;
    if (recursion > 4096) return fail(ctx, RT_ERR_LIMIT, "rt_set_frame: recursion %d > 4096", recursion);
    memcpy(ctx->frame.cam, camera, sizeof(float) * 16);
    ctx->frame.clock = clock;
    ctx->frame.recursion = recursion;
    ctx->frame.exposure = exposure;
    ctx->have_frame = true;
    ctx->chain_valid = false;
    ctx->tile_lists_valid = false;
    return RT_OK;
}

int rt_trace(rt_context *ctx, uint32_t sample_begin, uint32_t sample_count, uint32_t sample_stride) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_trace: no scene uploaded");
    if (!ctx->have_frame) return fail(ctx, RT_ERR_STATE, "rt_trace: no frame set");
    if (ctx->width == 0 || !ctx->d_sum) return fail(ctx, RT_ERR_STATE, "rt_trace: no target (rt_resize)");
    if (sample_count == 0) return RT_OK;
    if (sample_stride == 0) return fail(ctx, RT_ERR_INVALID, "rt_trace: sample_stride 0");
    if ((uint64_t)sample_begin + (uint64_t)(sample_count - 1) * sample_stride > 0xFFFFFFFEull)
        return fail(ctx, RT_ERR_LIMIT, "rt_trace: sample index beyond 2^32 - 2");
    if (bind_device(ctx)) return RT_ERR_DEVICE;

    ctx->frame.width = ctx->width;
    ctx->frame.height = ctx->height;
    ctx->frame.skip_nonfinite = ctx->skip_nonfinite ? 1u : 0u;
    const DScene sc = device_scene(ctx);

    // samples are processed in passes of at most `batch_samples`; the tables of a pass travel in the kernel arguments
    uint32_t chunk_max = (uint32_t)(ctx->batch_samples > 0 ? ctx->batch_samples : 1);
    if (chunk_max > (uint32_t)MAX_BATCH) chunk_max = (uint32_t)MAX_BATCH;
    if (sample_count < chunk_max) chunk_max = sample_count;
    const bool acc = use_accel(ctx);
    PassForm form;
    form.bvh = acc;
    form.resident = acc && ctx->resident != 0 && fits_lds(ctx);
    form.share = acc && ctx->share != 0;
    form.refill = acc && (ctx->refill < 0 ? ctx->count >= REFILL_AUTO_MIN : ctx->refill != 0);
    form.staged_depths = ctx->staged_depths;
    form.lists.lists = nullptr;
    form.lists.enabled = 0;
    if (ctx->pipeline != 0 && acc && ctx->tile_lists != 0 && ctx->count <= TILE_LIST_MAX_TRIANGLES) {
        // candidate lists of the camera rays, rebuilt when the camera, the target or the scene has changed
        const uint32_t tiles_x = (ctx->width + 7u) >> 3, tiles_y = (ctx->height + 7u) >> 3;
        const size_t tiles = (size_t)tiles_x * tiles_y;
        if (tiles > ctx->tile_lists_cap) {
            if (ctx->d_tile_lists) (void)hipFree(ctx->d_tile_lists);
            ctx->d_tile_lists = nullptr;
            ctx->tile_lists_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_tile_lists, tiles * TILE_LIST_WORDS * sizeof(uint32_t)));
            ctx->tile_lists_cap = tiles;
            ctx->tile_lists_valid = false;
        }
        if (!ctx->tile_lists_valid) {
            hipLaunchKernelGGL(k_build_tile_lists, dim3((unsigned)tiles), dim3(64), 0, ctx->stream, sc, ctx->frame, tiles_x, tiles_y, ctx->d_tile_lists);
            HIP_TRY(ctx, hipGetLastError());
            ctx->tile_lists_valid = true;
        }
        form.lists.lists = ctx->d_tile_lists;
        form.lists.enabled = 1;
    }
    if (ctx->pipeline != 0) {
        // The queues of a pass hold 1/queue_rounds of the case that every path survives every bounce (a pass that needs
        // more takes more rounds); fewer samples per pass if that does not fit the free device memory (the image does
        // not depend on the pass size).
        const size_t pps = (size_t)(((ctx->width + 7u) >> 3) * ((ctx->height + 7u) >> 3)) * 64u;
        const bool h0 = form.refill;
        const uint32_t rounds = (uint32_t)ctx->queue_rounds;
        if (pps * chunk_max > ctx->pipe.cap_paths || (h0 && !ctx->pipe.have_h0) || rounds != ctx->pipe.rounds) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t have = ctx->pipe.queue_bytes();      // released before the larger allocation is made
                while (chunk_max > 1u && (double)Pipeline::queue_bytes_for(pps * chunk_max, h0, rounds) > 0.9 * (double)(free_b + have))
                    chunk_max = (chunk_max + 1u) / 2u;
            }
            const char *msg = nullptr;
            const hipError_t e = ctx->pipe.ensure(pps * chunk_max, h0, rounds, &msg);
            if (e != hipSuccess)
                return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "rt_trace: %s: %s", msg ? msg : "queues",
                            hipGetErrorString(e));
        }
    }

    for (uint32_t done = 0; done < sample_count;) {
        const uint32_t n = (sample_count - done < chunk_max) ? sample_count - done : chunk_max;
        PassTables tab;
        tab.n = n;
        tab.first_sample = sample_begin + done * sample_stride;
        tab.stride = sample_stride;
        for (uint32_t k = 0; k < (uint32_t)MAX_BATCH; k++)
            tab.light_seed[k] = k < n ? light_seed_for(ctx, tab.first_sample + k * sample_stride) : 0u;
        if (ctx->pipeline == 0) {
            if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            const uint32_t tiles = ((ctx->width + 7u) >> 3) * ((ctx->height + 7u) >> 3);
            hipLaunchKernelGGL(k_trace_mega, dim3(tiles), dim3(64), 0, ctx->stream, sc, ctx->frame, ctx->d_sum, tab, ctx->d_stats);
            HIP_TRY(ctx, hipGetLastError());
            if (ctx->profiling) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
                HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
                float ms = 0.0f;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
                ctx->host_stats.trace_ms += (double)ms;
                ctx->host_stats.extend_ms += (double)ms;     // the single kernel holds every stage
                ctx->host_stats.extend_launches++;
            }
        } else {
            const char *msg = nullptr;
            const hipError_t e = ctx->pipe.run(sc, device_bvh(ctx, ctx->bvh, acc), device_bvh(ctx, ctx->light_bvh, acc), ctx->frame,
                                               ctx->d_sum, tab, ctx->stream, ctx->profiling != 0, form, &ctx->host_stats, ctx->d_stats, &msg);
            if (e != hipSuccess)
                return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "rt_trace: %s: %s",
                            msg ? msg : "pipeline", hipGetErrorString(e));
        }
        done += n;
    }
    if (ctx->pipeline != 0) ctx->staged_paths_expected += (uint64_t)ctx->width * ctx->height * sample_count;
    ctx->n_samples += sample_count;
    return RT_OK;
}

int rt_read_sum(rt_context *ctx, float *rgb, uint64_t *n_samples) {
    if (!ctx) return RT_ERR_INVALID;
    if (ctx->width == 0 || !ctx->d_sum) return fail(ctx, RT_ERR_STATE, "rt_read_sum: no target");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const unsigned long long n = (unsigned long long)ctx->width * ctx->height;
    if (rgb) {
        float *d_rgb = nullptr;
        HIP_TRY(ctx, hipMalloc(&d_rgb, sizeof(float) * 3 * n));
        hipLaunchKernelGGL(k_sum_to_rgb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sum, n, d_rgb);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(rgb, d_rgb, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_rgb);
        if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_read_sum: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_samples) *n_samples = ctx->n_samples;
    return RT_OK;
}

int rt_read_sum_static(rt_context *ctx, float *rgb, uint64_t *n_samples) {
    if (!ctx) return RT_ERR_INVALID;
    if (ctx->width == 0 || !ctx->d_sum) return fail(ctx, RT_ERR_STATE, "rt_read_sum_static: no target");
    const unsigned long long n = (unsigned long long)ctx->width * ctx->height;
    if (n > STATIC_MAX_PIXELS)
        return fail(ctx, RT_ERR_LIMIT, "rt_read_sum_static: %ux%u = %llu pixels, the built-in buffer holds %llu (use rt_read_sum)",
                    ctx->width, ctx->height, n, (unsigned long long)STATIC_MAX_PIXELS);
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    float *staging = nullptr;
    if (ctx->device < MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        staging = g_scratch_ptr[ctx->device];
    }
    if (!staging) return fail(ctx, RT_ERR_DEVICE, "rt_read_sum_static: the built-in buffer of device %d could not be located", ctx->device);
    if (rgb) {
        // every word that is copied out below is written by this launch first, so nothing of the previous call's
        // contents has to be cleared
        hipLaunchKernelGGL(k_sum_to_rgb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sum, n, staging);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(rgb, staging, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_read_sum_static: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_samples) *n_samples = ctx->n_samples;
    return RT_OK;
}

int rt_read_srgb8(rt_context *ctx, uint8_t *rgb) {
    if (!ctx || !rgb) return RT_ERR_INVALID;
    if (ctx->width == 0 || !ctx->d_sum) return fail(ctx, RT_ERR_STATE, "rt_read_srgb8: no target");
    if (ctx->n_samples == 0) return fail(ctx, RT_ERR_STATE, "rt_read_srgb8: nothing rendered yet");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const unsigned long long n = (unsigned long long)ctx->width * ctx->height;
    unsigned char *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_out, 3 * n));
    hipLaunchKernelGGL(k_srgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sum,
                       1.0f / (float)ctx->n_samples, n, ctx->tonemap, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb, d_out, 3 * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_read_srgb8: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_closest_hit_batch_device(rt_context *ctx, const float *o_xyz, const float *d_xyz, uint64_t n, int32_t *id,
                                float *uvt) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_closest_hit_batch: no scene uploaded");
    if (n == 0) return RT_OK;
    if (!o_xyz || !d_xyz || !id || !uvt) return fail(ctx, RT_ERR_INVALID, "rt_closest_hit_batch: null array");
    if (n > (1ull << 31)) return fail(ctx, RT_ERR_LIMIT, "rt_closest_hit_batch: more than 2^31 rays");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    const DScene sc = device_scene(ctx);
    if (use_accel(ctx))
        hipLaunchKernelGGL(k_closest_hit_batch_bvh, dim3((unsigned)std::min<uint64_t>((n + 63) / 64, 4096)), dim3(64), 0, ctx->stream, sc,
                           device_bvh(ctx, ctx->bvh, true), o_xyz, d_xyz, (unsigned long long)n, id, uvt, ctx->d_stats);
    else
        hipLaunchKernelGGL(k_closest_hit_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc, o_xyz,
                           d_xyz, (unsigned long long)n, id, uvt);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_closest_hit_batch(rt_context *ctx, const float *o_xyz, const float *d_xyz, uint64_t n, int32_t *id,
                         float *uvt) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_closest_hit_batch: no scene uploaded");
    if (n == 0) return RT_OK;
    if (!o_xyz || !d_xyz || !id || !uvt) return fail(ctx, RT_ERR_INVALID, "rt_closest_hit_batch: null array");
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    float *d_o = nullptr, *d_d = nullptr, *d_uvt = nullptr;
    int32_t *d_id = nullptr;
    int rc = RT_OK;
    hipError_t e = hipMalloc(&d_o, sizeof(float) * 3 * n);
    if (e == hipSuccess) e = hipMalloc(&d_d, sizeof(float) * 3 * n);
    if (e == hipSuccess) e = hipMalloc(&d_uvt, sizeof(float) * 3 * n);
    if (e == hipSuccess) e = hipMalloc(&d_id, sizeof(int32_t) * n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_o, o_xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, d_xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = rt_closest_hit_batch_device(ctx, d_o, d_d, n, d_id, d_uvt);
        if (rc == RT_OK) {
            e = hipMemcpyAsync(id, d_id, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(uvt, d_uvt, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_uvt); (void)hipFree(d_id);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE,
                                     "rt_closest_hit_batch: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_get_stats(rt_context *ctx, rt_stats *out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, ctx->pipe.collect_times(&ctx->host_stats));
    DStats ds;
    HIP_TRY(ctx, hipMemcpy(&ds, ctx->d_stats, sizeof(ds), hipMemcpyDeviceToHost));
    unsigned long long ws[WSTAT_SLOTS];
    HIP_TRY(ctx, ctx->pipe.read_wstats(ws));
    // the rounds of a pass get through all its chunks by the sizing of the queue regions; were that ever wrong, camera
    // paths would be missing from the count
    if (ws[WSTAT_PATHS] != ctx->staged_paths_expected)
        return fail(ctx, RT_ERR_DEVICE, "rt_get_stats: %llu of %llu camera paths were rendered (queue_rounds)",
                    (unsigned long long)ws[WSTAT_PATHS], (unsigned long long)ctx->staged_paths_expected);
    ds.pixel_samples += ws[WSTAT_PATHS];
    ds.rays += ws[WSTAT_RAYS];
    ds.shadow_probes += ws[WSTAT_SHADOW];
    ds.glossy_probes += ws[WSTAT_GLOSSY];
    ds.ext_nodes += ws[WSTAT_EXT_NODES];
    ds.ext_tris += ws[WSTAT_EXT_TRIS];
    ds.probe_nodes += ws[WSTAT_PROBE_NODES];
    ds.probe_tris += ws[WSTAT_PROBE_TRIS];
    rt_stats s = ctx->host_stats;
    s.pixel_samples = ds.pixel_samples;
    s.rays = ds.rays;
    s.shadow_probes = ds.shadow_probes;
    s.glossy_probes = ds.glossy_probes;
    s.closest_tests = ds.rays * (uint64_t)ctx->count;
    s.probe_tests = (ds.shadow_probes + ds.glossy_probes) * (uint64_t)(ctx->count > 0 ? ctx->count - 1 : 0);
    s.bvh_extend_nodes = ds.ext_nodes;
    s.bvh_extend_tris = ds.ext_tris;
    s.bvh_probe_nodes = ds.probe_nodes;
    s.bvh_probe_tris = ds.probe_tris;
    s.depth1_paths = ws[WSTAT_DEPTH1];
    s.deferred_probes = ws[WSTAT_DEFERRED];
    s.parked_probes = ws[WSTAT_PARKED];
    s.skipped_samples = ds.skipped;
    s.queue_bytes = ctx->pipe.queue_bytes();
    *out = s;
    return RT_OK;
}

int rt_reset_stats(rt_context *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (bind_device(ctx)) return RT_ERR_DEVICE;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemset(ctx->d_stats, 0, sizeof(DStats)));
    HIP_TRY(ctx, ctx->pipe.reset_wstats(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memset(&ctx->host_stats, 0, sizeof(ctx->host_stats));
    ctx->staged_paths_expected = 0;
    rt_stats discarded;
    memset(&discarded, 0, sizeof(discarded));
    HIP_TRY(ctx, ctx->pipe.collect_times(&discarded));
    return RT_OK;
}

int rt_set_option(rt_context *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return RT_ERR_INVALID;
    if (!strcmp(name, "pipeline")) {
        if (value != 0 && value != 1) return fail(ctx, RT_ERR_INVALID, "pipeline must be 0 or 1");
        ctx->pipeline = (int)value;
    } else if (!strcmp(name, "batch_samples")) {
        if (value < 1 || value > MAX_BATCH) return fail(ctx, RT_ERR_LIMIT, "batch_samples must be in [1, %d]", MAX_BATCH);
        ctx->batch_samples = (int)value;
    } else if (!strcmp(name, "queue_rounds")) {
        if (value < 1 || value > (int64_t)MAX_ROUNDS) return fail(ctx, RT_ERR_LIMIT, "queue_rounds must be in [1, %d]", (int)MAX_ROUNDS);
        ctx->queue_rounds = (int)value;
    } else if (!strcmp(name, "accel")) {
        if (value < -1 || value > 1) return fail(ctx, RT_ERR_INVALID, "accel must be -1 (auto), 0 or 1");
        ctx->accel = (int)value;
    } else if (!strcmp(name, "refill")) {
        ctx->refill = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(name, "resident")) {
        ctx->resident = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(name, "tonemap")) {
        if (value != 0 && value != 1) return fail(ctx, RT_ERR_INVALID, "tonemap must be 0 or 1");
        ctx->tonemap = (int)value;
    } else if (!strcmp(name, "tile_lists")) {
        ctx->tile_lists = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(name, "skip_nonfinite")) {
        ctx->skip_nonfinite = value ? 1 : 0;
    } else if (!strcmp(name, "share")) {
        ctx->share = value ? 1 : 0;
    } else if (!strcmp(name, "staged_depths")) {
        if (value < -1 || value > 4096) return fail(ctx, RT_ERR_INVALID, "staged_depths must be in [-1, 4096]");
        ctx->staged_depths = (int)value;
    } else if (!strcmp(name, "profiling")) {
        ctx->profiling = value ? 1 : 0;
    } else {
        return fail(ctx, RT_ERR_INVALID, "unknown option %s", name);
    }
    return RT_OK;
}

}  // extern "C"
