/*
 * gpurt.h - C ABI of the MI355X path-tracing core (libgpurt.so).
 *
 * The reference renderer has no plugin or FFI layer: its compute path tracer
 * (res/shader/raytracer.glsl) sits behind six methods of class Scene that talk
 * straight to OpenGL.  Each entry point below replaces the OpenGL traffic of one of
 * those methods; the comment on each names the reference interface it stands for.
 *
 * Conventions
 *   - Every function returns RT_OK (0) or a negative rt_status; rt_last_error()
 *     gives the text of the last failure on that context.
 *   - A context is bound to one GPU and may be used from one host thread at a time.
 *   - Host pointers passed in are copied before the call returns; the caller keeps
 *     ownership.  rt_trace() only enqueues kernels on the context's stream (it reads
 *     nothing back and copies nothing; the first call for a larger target or pass
 *     also allocates the path queues, any later one can be captured in a HIP graph);
 *     rt_read_sum(), rt_synchronize() and rt_get_stats() wait for the stream.
 *   - The three record types are byte-identical to the reference's SSBO layouts.
 */
#ifndef GPURT_H
#define GPURT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument */
    RT_ERR_STATE = -2,     /* call order: no scene, no frame, no target ... */
    RT_ERR_DEVICE = -3,    /* a HIP call failed */
    RT_ERR_NOMEM = -4,
    RT_ERR_LIMIT = -5      /* size beyond a documented maximum */
} rt_status;

/* TriangleModel: 3Dobjects.hpp:102-113, raytracer.glsl:13-18.  48 bytes. */
typedef struct rt_triangle_model {
    float true_normal[3];  /* cross(u, v), not normalised */
    float position[3];
    float u[3];
    float v[3];
} rt_triangle_model;

/* TriangleShading: 3Dobjects.hpp:115-135, raytracer.glsl:20-29.  100 bytes. */
typedef struct rt_triangle_shading {
    uint32_t material_id;
    float normals[3][3];
    float tangents[3][3];
    float tex_p[2];
    float tex_u[2];
    float tex_v[2];
} rt_triangle_shading;

/* Material: Material.hpp:9-15, raytracer.glsl:36-40.  48 bytes. */
typedef struct rt_material {
    float albedo[4];              /* rgb, a unused */
    float specular_roughness[4];  /* rgb = Ks, a = Ns / 1000 */
    float emission_ior[4];        /* rgb = Ke, a = Ni */
} rt_material;

/* Work counters of the launches since the last rt_resize()/rt_reset_stats(). */
typedef struct rt_stats {
    uint64_t pixel_samples;     /* camera paths started */
    uint64_t rays;              /* findIntersection executions (paths x bounces) */
    uint64_t shadow_probes;     /* sampleLight calls that reached the occlusion loop */
    uint64_t glossy_probes;     /* sampleLightGlossy calls */
    uint64_t closest_tests;     /* rays * COUNT: the tests the reference's loop would run */
    uint64_t probe_tests;       /* probes * (COUNT - 1), likewise */
    double   trace_ms;          /* device time of all rt_trace kernels (HIP events on the context stream) */
    double   extend_ms;         /* part of trace_ms spent in closest-hit kernels */
    double   probe_ms;          /* part of trace_ms spent in light-probe kernels */
    double   shade_ms;          /* part of trace_ms spent in shading kernels */
    uint64_t extend_launches;
    uint64_t probe_launches;
    double   extend0_ms;        /* part of extend_ms spent in the depth-0 (camera ray) closest-hit launches */
    uint64_t extend0_launches;
    /* hierarchy traversal (option "accel"): node visits and triangle tests actually executed.  Counted by the
     * measurement builds of the library only (libgpurt_steps.so; libgpurt_wavesteps.so counts lane slots instead);
     * the product leaves them zero, the counting costs its kernels 1.4 %. */
    uint64_t bvh_extend_nodes, bvh_extend_tris, bvh_probe_nodes, bvh_probe_tris;
    uint64_t depth1_paths;      /* camera paths that went on to a second bounce (staged form; else 0) */
    uint64_t queue_bytes;       /* device memory the path queues of the staged form hold at present */
    uint64_t deferred_probes;   /* light probes of depth 0 that the depth-0 kernel left to the probe kernels (as 112-byte records) */
    uint64_t parked_probes;     /* light probes of depth 0 that the depth-0 kernel wrote as records: those it took up again itself with
                                 * idle lanes of later chunks, and the deferred ones */
    uint64_t skipped_samples;   /* pixel-samples whose radiance was NaN or infinite and was left out of the sums (option "skip_nonfinite") */
} rt_stats;

typedef struct rt_context rt_context;

/* Lifetime.  `device` is the HIP device ordinal.  Scene::Scene / Scene::~Scene
 * (Scene.cpp:231-280) own every GL object in the reference; the context owns every
 * device allocation here. */
/* rt_create refuses (RT_ERR_DEVICE) a device that is not gfx950.  On a device with another number of compute units than
 * the 256 the work distribution is laid out for (a partition of the chip), it succeeds and leaves a note that starts
 * with "note:" for rt_last_error(); results do not depend on it. */
int rt_create(int device, rt_context **out);
int rt_destroy(rt_context *ctx);
const char *rt_last_error(const rt_context *ctx);

/* Launch subsequent work on `hip_stream` (a hipStream_t, or NULL for the context's own). */
int rt_set_stream(rt_context *ctx, void *hip_stream);
int rt_synchronize(rt_context *ctx);

/* Scene::finalizeObjects -> createTrianglesBuffers (Scene.cpp:382-395, :343-346):
 * the three immutable SSBOs plus the COUNT (:362) and LIGHTS (:293) uniforms.
 * Triangles [0, lights) are the emissive ones.  Records are re-laid-out to SoA on
 * the device.  At most 2^28 - 1 triangles (RT_ERR_LIMIT beyond). */
int rt_upload_scene(rt_context *ctx,
                    const rt_triangle_model *models, const rt_triangle_shading *shadings,
                    uint32_t count, uint32_t lights,
                    const rt_material *materials, uint32_t n_materials);

/* Scene::createRTCSData atlas allocation (Scene.cpp:364-370).  The reference atlas is
 * 4096 x 4096 x (3 * materials) texels of R11F_G11F_B10F; `size` selects the edge
 * length (4096 by default).  Layers are allocated when first uploaded. */
int rt_set_atlas_size(rt_context *ctx, uint32_t size);

/* Scene::loadTexture (Scene.cpp:75-102): RGBA32F host pixels -> texels [0,w) x [0,h)
 * of layer material*3 + slot (slot 0 albedo, 1 normal, 2 occlusion/roughness/metallic),
 * converted to packed 11/11/10-bit unsigned floats. */
int rt_upload_atlas_layer(rt_context *ctx, uint32_t material, uint32_t slot,
                          uint32_t w, uint32_t h, const float *rgba);

/* Scene::loadMaterial's glNamedBufferSubData(hasTextureBuffer, ...) (Scene.cpp:227). */
int rt_set_has_texture(rt_context *ctx, const int32_t *flags, uint32_t n);

/* Scene::loadEnvironmentTexture (Scene.cpp:104-160): RGBA32F, bilinear, repeat. */
int rt_upload_env(rt_context *ctx, uint32_t w, uint32_t h, const float *rgba);

/* The irradiance pre-filter Scene::loadEnvironmentTexture runs when no "<name>-irradiance.exr" exists
 * (Scene.cpp:150-186 with res/shader/irradiance.glsl:20-39): from the uploaded w x h environment to an
 * (w/4) x (h/4) RGB float image, row 0 first, as the reference reads it back for SaveEXR (Scene.cpp:188-199).
 * `rgb` may be NULL to query the size only.  The path tracer itself never samples this image
 * (skyColorDiffuse, raytracer.glsl:277-280, has no caller); it exists for hosts that write the file. */
int rt_prefilter_irradiance(rt_context *ctx, float *rgb, uint32_t *out_w, uint32_t *out_h);

/* Scene::adaptResolution (Scene.cpp:397-421): (re)allocate the accumulation target
 * and clear it.  The target holds per-pixel SUMS of sample radiance, not the
 * reference's running mean, so that partial results from several launches or
 * several GPUs add. */
int rt_resize(rt_context *ctx, uint32_t width, uint32_t height);

/* Scene::prepare's glBindImageTexture(0, moving ? renderTargetLow : renderTarget, ...) (Scene.cpp:450, :459; the two
 * textures are made at Scene.cpp:405-415).  A context has two accumulation targets, 0 = the still image (selected at
 * first) and 1 = the low-resolution preview the reference traces while the camera moves; each has its own size
 * (rt_resize), sums and sample count.  rt_resize, rt_clear, rt_set_sum_buffer, rt_trace, rt_read_sum, rt_read_sum_static,
 * rt_read_srgb8 and rt_accumulate_from work on the selected one; the other keeps what it has, so a host can trace a
 * few preview frames and then carry on accumulating the still image. */
int rt_select_target(rt_context *ctx, int target);

/* What SAMPLE = 0 does to the reference's running mean (raytracer.glsl:611-613: the old value is weighted 0 / 1):
 * the selected target starts over, sums and sample count zero.  Enqueued on the context's stream; nothing is
 * reallocated. */
int rt_clear(rt_context *ctx);

/* Accumulate into caller-owned device memory (width*height float4, rgb = sums)
 * instead of the context's own buffer; NULL restores the internal one.  Lets a
 * host keep the buffer in a tensor it can hand to a collective. */
int rt_set_sum_buffer(rt_context *ctx, float *device_rgba);

/* No counterpart in the reference (it renders on one GPU): adds the sums `other` has accumulated, and its sample count,
 * to those of `ctx`.  Both must have the same target size; `other` may live on another GPU (its sums then travel over
 * the peer link, xGMI between the GPUs of one node, through a staging buffer this call allocates and frees).  Waits for
 * `other`'s stream, enqueues the addition on `ctx`'s and returns when it is done.  For hosts that drive several GPUs
 * from one process: context r of N renders SAMPLE = r + k * N (rt_trace(r, count, N)), then context 0 adds the
 * others in ascending order, which fixes the association of the float sums. */
int rt_accumulate_from(rt_context *ctx, rt_context *other);

/* Scene::prepare (Scene.cpp:435-464): CAMERA (16 floats, column-major),
 * CLOCK, RECURSION, and the EXPOSURE uniform set at Scene.cpp:331. */
int rt_set_frame(rt_context *ctx, const float camera[16], uint32_t clock,
                 int32_t recursion, float exposure);

/* Scene::traceScene (Scene.cpp:423-433), one reference dispatch per sample index:
 * renders SAMPLE = sample_begin + k * sample_stride for k in [0, sample_count)
 * and adds each sample's radiance to the sum buffer. */
int rt_trace(rt_context *ctx, uint32_t sample_begin, uint32_t sample_count,
             uint32_t sample_stride);

/* Scene::exportEXR's glGetTextureImage(GL_RGB, GL_FLOAT) (Scene.cpp:496-509):
 * copies width*height*3 floats (row 0 = TEXEL.y 0, as the reference writes them)
 * and the number of samples summed per pixel. */
int rt_read_sum(rt_context *ctx, float *rgb, uint64_t *n_samples);

/* No counterpart in the reference.  rt_read_sum for callers that cannot allocate device memory (a preview loop that reads
 * the image back after every pass, code under HIP graph capture, harnesses that forbid hipMalloc): the float3 staging
 * image that rt_read_sum allocates and frees on every call lives in a 32 MiB array built into the library instead,
 * so this call allocates nothing.  Same output as rt_read_sum, bit for bit.  Targets of more than 2 796 202 pixels
 * (1920 x 1080 fits, 3840 x 2160 does not) return RT_ERR_LIMIT.  Every context on one GPU shares that GPU's array:
 * calls on the same GPU must not overlap in time (two host threads, two streams), or they overwrite each other's image. */
int rt_read_sum_static(rt_context *ctx, float *rgb, uint64_t *n_samples);

/* No counterpart in the reference (findIntersection, raytracer.glsl:217-235, cannot be
 * called on its own there): closest hit for a batch of rays.  o_xyz/d_xyz are n*3
 * floats, id gets the triangle index (-1 = miss), uvt gets n*3 floats (u, v, t). */
int rt_closest_hit_batch(rt_context *ctx, const float *o_xyz, const float *d_xyz,
                         uint64_t n, int32_t *id, float *uvt);

/* Same with all four arrays already in device memory (asynchronous). */
int rt_closest_hit_batch_device(rt_context *ctx, const float *o_xyz, const float *d_xyz,
                                uint64_t n, int32_t *id, float *uvt);

/* displayQuad.fs:19-34: sRGB transfer of sum / n_samples to 8-bit RGB (row order as rt_read_sum).  With option
 * "tonemap" = 1: the mapping of Scene::exportBMP instead (Scene.cpp:564-567: ACESFilm, sRGB transfer, v * 256
 * truncated), still in RGB order. */
int rt_read_srgb8(rt_context *ctx, uint8_t *rgb);

int rt_get_stats(rt_context *ctx, rt_stats *out);
int rt_reset_stats(rt_context *ctx);

/* Options; unknown names return RT_ERR_INVALID.  None of them changes a result, except "skip_nonfinite" and "tonemap".
 *   "pipeline"      0 = one kernel per sample, one lane per pixel for the whole path;
 *                   1 = staged kernels over compacted ray queues (default)
 *   "batch_samples" samples rendered per pass of the staged pipeline (default 32, at most 512; the queues of a pass
 *                   take about 44 + 240 / queue_rounds bytes of device memory per pixel and sample,
 *                   rt_stats.queue_bytes tells)
 *   "queue_rounds"  n in [1, 8] (default 3): the path queues hold 1/n of the paths a pass would hand on if every path
 *                   survived every bounce; a pass whose survivors do not fit takes up to n rounds of its launch
 *                   sequence (about a fifth of the camera paths goes on in the reference's bundled scenes: one round;
 *                   a closed bright room needs them all, and 1 then spares the repeated launches)
 *   "accel"         -1 = bounding-volume hierarchy for scenes of 64 triangles or more (default; scenes with a triangle
 *                   whose |cross(u, v)| is below 2^-40 keep the loops: the reference's depth comparison underflows
 *                   there and the later index wins whatever the depths, which only the loops reproduce),
 *                   0 = the reference's loops over all triangles, 1 = hierarchy always
 *   "resident"      -1 = a hierarchy that fits (about 480 triangles, depth under 16) is staged into LDS whole, leaf
 *                   triangles included (default), 0 = only its first nodes are
 *   "refill"        1 = the closest-hit kernels of the bounces after the first hand finished lanes new rays (needs
 *                   the hierarchy), 0 = every wave processes its queue in chunks of 64, -1 = by scene size
 *                   (default: from 65536 triangles on)
 *   "staged_depths" n >= 0: the n bounces after the first run as staged kernels, then one kernel walks the rest of
 *                   every path that is still under way (default 2); -1 = every bounce runs staged
 *   "share"         1 = the lanes of a wave share the any-hit searches of its light probes: an idle lane takes a
 *                   subtree off the stack of a busy one (default; needs the hierarchy), 0 = every lane searches alone
 *   "tile_lists"    -1 = for scenes of up to 16384 triangles the camera rays of an 8x8 pixel tile are tested against a list
 *                   of the triangles that tile's rays can touch, made once per frame, instead of searching the hierarchy
 *                   (default; needs the hierarchy for the tiles whose lists would be too long), 0 = never
 *   "skip_nonfinite" 1 = a sample whose radiance is NaN or infinite is not added to its pixel's sum (rt_stats.skipped_samples
 *                   counts them; the sample still counts in n_samples).  0 (default) = it is added, as the reference does
 *                   (raytracer.glsl:204, sqrt(1 - cos^2) of a rounded cosine above 1, and :274, asin of a direction component
 *                   beyond 1, yield NaN about once in 10^8 to 10^10 pixel-samples, and the pixel then stays NaN for the rest
 *                   of the image)
 *   "tonemap"       what rt_read_srgb8 applies: 0 = sRGB transfer (default), 1 = ACES film curve + sRGB transfer
 *   "profiling"     1 = bracket kernels with HIP events and fill the *_ms fields of rt_stats */
int rt_set_option(rt_context *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* GPURT_H */
