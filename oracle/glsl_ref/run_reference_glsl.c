/*
 * run_reference_glsl.c - runs the reference's UNMODIFIED compute shader (res/shader/raytracer.glsl)
 * under Mesa llvmpipe without a window system, to produce golden data for the CPU checker.
 * TEST INFRASTRUCTURE ONLY (build container only: it reads the shader from /root/reference at run
 * time; nothing of the reference is copied into this repository).
 *
 * There is no X server, EGL or OSMesa in the build container, so the GL context is created by loading
 * Mesa's software DRI driver (swrast_dri.so) directly through the DRI "swrast" loader interface
 * (GL/internal/dri_interface.h): createNewScreen2 -> createContextAttribs(4.5 core) ->
 * createNewDrawable -> bindContext, GL entry points through libglapi's _glapi_get_proc_address.
 *
 * Modes
 *   image  <scene.rtscene> <env.f32> <env w> <env h> <W> <H> <clock> <recursion> <n_samples> <out.f32> <camera.f32> [first [atlas.bin]]
 *          dispatches the shader for SAMPLE = first..first+n-1 (first defaults to 0) as Scene::traceScene does
 *          (Scene.cpp:423-433) on a zero-initialised target and writes the RGBA32F accumulation image (the
 *          shader's own running mean).  atlas.bin: uint32 layer edge S, uint32 n uploads, then per upload uint32 material,
 *          slot, w, h and w*h RGBA32F texels (uploaded as Scene::loadTexture does, Scene.cpp:93-98), then one int32
 *          hasTexture flag per material.  Without it the atlas is a 1 x 1 x 3M array and no material is textured.
 *   hits   <scene.rtscene> <rays.f32 (n x 6: origin, direction)> <n> <out.bin (n x int32 id, then n x 3 float uvt)>
 *          runs a harness shader assembled at run time from the reference's own text (everything up to and
 *          including findIntersection, raytracer.glsl:1-235) plus a main() that calls findIntersection per ray.
 */
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_SHADER "/root/reference/res/shader/raytracer.glsl"

static void get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p;
    *x = 0; *y = 0; *w = 16; *h = 16;
}
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)p;
    memset(data, 0, (size_t)w * h * 4);
}
static void put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)p;
    memset(data, 0, (size_t)stride * h);
}

static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = get_drawable_info,
    .putImage = put_image,
    .getImage = get_image,
    .putImage2 = put_image2,
    .getImage2 = get_image2,
};
static const __DRIextension *loader_extensions[] = {&swrast_loader.base, NULL};

typedef void (*glproc)(void);
/* GL 1.x entry points have no PFN typedefs in glext.h */
typedef void (*PFN_GENTEXTURES)(GLsizei, GLuint *);
typedef void (*PFN_BINDTEXTURE)(GLenum, GLuint);
typedef void (*PFN_TEXSUBIMAGE2D)(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *);
typedef void (*PFN_TEXPARAMETERI)(GLenum, GLenum, GLint);
typedef void (*PFN_GETTEXIMAGE)(GLenum, GLint, GLenum, GLenum, void *);
typedef void (*PFN_FINISH)(void);
typedef GLenum (*PFN_GETERROR)(void);
static glproc (*get_proc)(const char *);
#define GLF(type, name) type name = (type)get_proc(#name); if (!name) { fprintf(stderr, "missing %s\n", #name); exit(3); }

static char *read_file(const char *path, size_t *size) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    buf[n] = 0;
    fclose(f);
    if (size) *size = (size_t)n;
    return buf;
}

struct scene { uint32_t count, lights, n_mat; const void *models, *shadings, *materials; };

static struct scene load_scene(const char *path) {
    size_t n;
    char *d = read_file(path, &n);
    struct scene s;
    if (memcmp(d, "RTSC0001", 8)) { fprintf(stderr, "bad scene file\n"); exit(2); }
    memcpy(&s.count, d + 8, 4); memcpy(&s.lights, d + 12, 4); memcpy(&s.n_mat, d + 16, 4);
    s.models = d + 24;
    s.shadings = d + 24 + 48ul * s.count;
    s.materials = d + 24 + 148ul * s.count;
    return s;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s image|hits|func|pack|irradiance|probe ...\n", argv[0]); return 1; }
    setenv("GALLIUM_DRIVER", "llvmpipe", 0);
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "dlopen driver: %s\n", dlerror()); return 10; }
    const __DRIextension **(*get_ext)(void) = (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_ext) { fprintf(stderr, "no __driDriverGetExtensions_swrast\n"); return 10; }
    const __DRIextension **drv_ext = get_ext();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; drv_ext[i]; i++) {
        if (!strcmp(drv_ext[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)drv_ext[i];
        if (!strcmp(drv_ext[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)drv_ext[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { fprintf(stderr, "driver lacks core/swrast v4\n"); return 10; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, loader_extensions, drv_ext, &configs, NULL);
    if (!screen || !configs || !configs[0]) { fprintf(stderr, "createNewScreen2 failed\n"); return 11; }
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed (error %u)\n", err); return 12; }
    __DRIdrawable *draw = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!draw || !core->bindContext(ctx, draw, draw)) { fprintf(stderr, "bindContext failed\n"); return 13; }
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    get_proc = glapi ? (glproc (*)(const char *))dlsym(glapi, "_glapi_get_proc_address") : NULL;
    if (!get_proc) { fprintf(stderr, "no _glapi_get_proc_address\n"); return 14; }

    GLF(PFNGLGETSTRINGIPROC, glGetStringi);
    const GLubyte *(*p_glGetString)(GLenum) = (const GLubyte *(*)(GLenum))get_proc("glGetString");
    fprintf(stderr, "GL_VERSION  %s\nGL_RENDERER %s\n", p_glGetString(GL_VERSION), p_glGetString(GL_RENDERER));
    (void)glGetStringi;
    if (!strcmp(argv[1], "probe")) return 0;

    GLF(PFNGLCREATESHADERPROC, glCreateShader); GLF(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLF(PFNGLCOMPILESHADERPROC, glCompileShader); GLF(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog); GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLF(PFNGLATTACHSHADERPROC, glAttachShader); GLF(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv); GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    GLF(PFNGLUSEPROGRAMPROC, glUseProgram); GLF(PFNGLGENBUFFERSPROC, glGenBuffers);
    GLF(PFNGLBINDBUFFERPROC, glBindBuffer); GLF(PFNGLBUFFERDATAPROC, glBufferData);
    GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase); GLF(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData);
    GLF(PFN_GENTEXTURES, glGenTextures); GLF(PFN_BINDTEXTURE, glBindTexture);
    GLF(PFNGLTEXSTORAGE2DPROC, glTexStorage2D); GLF(PFNGLTEXSTORAGE3DPROC, glTexStorage3D);
    GLF(PFN_TEXSUBIMAGE2D, glTexSubImage2D); GLF(PFN_TEXPARAMETERI, glTexParameteri);
    GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture); GLF(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
    GLF(PFNGLUNIFORM1IPROC, glUniform1i); GLF(PFNGLUNIFORM1UIPROC, glUniform1ui); GLF(PFNGLUNIFORM1FPROC, glUniform1f);
    GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv); GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier); GLF(PFN_GETTEXIMAGE, glGetTexImage);
    GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation); GLF(PFN_FINISH, glFinish);
    GLF(PFN_GETERROR, glGetError);

    if (!strcmp(argv[1], "pack")) {
        /* pack <in.f32> <n texels, a multiple of 100> <out.f32>: what the GL implementation stores in a GL_R11F_G11F_B10F texture for
         * RGBA float input (the conversion of Scene::loadTexture, Scene.cpp:93-98), read back as RGB float. */
        const int n = atoi(argv[3]);
        size_t got;
        char *in = read_file(argv[2], &got);
        if (got < 16ul * n) { fprintf(stderr, "input too short\n"); return 2; }
        GLuint tex;
        glGenTextures(1, &tex);
        glBindTexture(GL_TEXTURE_2D, tex);
        glTexStorage2D(GL_TEXTURE_2D, 1, GL_R11F_G11F_B10F, 100, n / 100);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, 100, n / 100, GL_RGBA, GL_FLOAT, in);
        float *out = (float *)malloc(12ul * n);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGB, GL_FLOAT, out);
        FILE *f = fopen(argv[4], "wb");
        fwrite(out, 12, (size_t)n, f);
        fclose(f);
        fprintf(stderr, "pack: %d texels, gl error 0x%x\n", n, glGetError());
        return 0;
    }

    if (!strcmp(argv[1], "func")) {
        /* func <shader.glsl> <in.f32> <n_in_floats> <n_items> <out.f32> <out_floats_per_item> [ref|own [<env.f32> <w> <h>]]:
         * runs a probe shader (SSBO 7 = input floats, SSBO 8 = output floats, uniform 20 = item count), one
         * invocation per item.  With "ref" the shader text is appended to the reference's text before its
         * main(), so that the probe can call the reference's own functions.  An environment map, if given, is
         * bound as Scene::loadEnvironmentTexture does (RGBA32F, LINEAR, REPEAT; sampler uniform location 2). */
        char *probe = read_file(argv[2], NULL);
        char *text = probe;
        if (argc > 8 && !strcmp(argv[8], "ref")) {
            char *ref = read_file(REF_SHADER, NULL);
            char *cut = strstr(ref, "void main(void)");
            if (!cut) { fprintf(stderr, "cannot find main() in the reference shader\n"); return 20; }
            size_t head = (size_t)(cut - ref);
            text = (char *)malloc(head + strlen(probe) + 1);
            memcpy(text, ref, head);
            strcpy(text + head, probe);
        }
        GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
        const char *srcs[1] = {text};
        glShaderSource(sh, 1, srcs, NULL);
        glCompileShader(sh);
        GLint ok = 0;
        glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) { char log[8192]; glGetShaderInfoLog(sh, sizeof(log), NULL, log); fprintf(stderr, "compile failed:\n%s\n", log); return 21; }
        GLuint prog = glCreateProgram();
        glAttachShader(prog, sh);
        glLinkProgram(prog);
        glGetProgramiv(prog, GL_LINK_STATUS, &ok);
        if (!ok) { char log[8192]; glGetProgramInfoLog(prog, sizeof(log), NULL, log); fprintf(stderr, "link failed:\n%s\n", log); return 22; }
        glUseProgram(prog);
        const size_t n_in = (size_t)atol(argv[4]), n_items = (size_t)atol(argv[5]), per = (size_t)atol(argv[7]);
        size_t got;
        char *in = read_file(argv[3], &got);
        if (got < 4 * n_in) { fprintf(stderr, "input too short\n"); return 2; }
        GLuint b[2];
        glGenBuffers(2, b);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, b[0]); glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(4 * n_in), in, GL_STATIC_DRAW);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 7, b[0]);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]); glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(4 * n_items * per), NULL, GL_DYNAMIC_READ);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 8, b[1]);
        glUniform1ui(20, (GLuint)n_items);
        if (argc > 11) {
            const int ew = atoi(argv[10]), eh = atoi(argv[11]);
            size_t en;
            char *env = read_file(argv[9], &en);
            if (en < 16ul * ew * eh) { fprintf(stderr, "environment file too short\n"); return 2; }
            GLuint tex;
            glGenTextures(1, &tex);
            glActiveTexture(GL_TEXTURE2);
            glBindTexture(GL_TEXTURE_2D, tex);
            glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, ew, eh);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
            glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, ew, eh, GL_RGBA, GL_FLOAT, env);
            glUniform1i(2, 2);
            glUniform1f(13, 1.0f);
        }
        glDispatchCompute((GLuint)((n_items + 63) / 64), 1, 1);
        glMemoryBarrier(GL_ALL_BARRIER_BITS);
        glFinish();
        float *out = (float *)malloc(4 * n_items * per);
        glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * n_items * per), out);
        FILE *f = fopen(argv[6], "wb");
        fwrite(out, 4, n_items * per, f);
        fclose(f);
        fprintf(stderr, "func: %zu items, gl error 0x%x\n", n_items, glGetError());
        return 0;
    }

    if (!strcmp(argv[1], "irradiance")) {
        /* irradiance <env.f32> <w> <h> <out.f32>: the pre-filter of Scene::loadEnvironmentTexture (Scene.cpp:150-186)
         * with the reference's res/shader/irradiance.glsl, unmodified: radiance image w x h on image unit 0, irradiance
         * image (w/4) x (h/4) on unit 1, one dispatch per radiance row (uniform 1 = row), ((w/4) >> 6) + 1 by h/4
         * workgroups (what the reference's ceilPower2 evaluates to), read back as RGB float.  The reference starts
         * from freshly allocated texture storage; here the irradiance image is explicitly zero-filled. */
        const int w = atoi(argv[3]), h = atoi(argv[4]);
        size_t got;
        char *env = read_file(argv[2], &got);
        if (got < 16ul * w * h) { fprintf(stderr, "environment file too short\n"); return 2; }
        char *text = read_file("/root/reference/res/shader/irradiance.glsl", NULL);
        GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
        const char *srcs[1] = {text};
        glShaderSource(sh, 1, srcs, NULL);
        glCompileShader(sh);
        GLint ok = 0;
        glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) { char log[8192]; glGetShaderInfoLog(sh, sizeof(log), NULL, log); fprintf(stderr, "compile failed:\n%s\n", log); return 21; }
        GLuint prog = glCreateProgram();
        glAttachShader(prog, sh);
        glLinkProgram(prog);
        glGetProgramiv(prog, GL_LINK_STATUS, &ok);
        if (!ok) { char log[8192]; glGetProgramInfoLog(prog, sizeof(log), NULL, log); fprintf(stderr, "link failed:\n%s\n", log); return 22; }
        glUseProgram(prog);
        const int iw = w / 4, ih = h / 4;
        GLuint tex[2];
        glGenTextures(2, tex);
        glBindTexture(GL_TEXTURE_2D, tex[0]);
        glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, w, h);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w, h, GL_RGBA, GL_FLOAT, env);
        glBindTexture(GL_TEXTURE_2D, tex[1]);
        glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, iw, ih);
        float *zero = (float *)calloc((size_t)iw * ih * 4, 4);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, iw, ih, GL_RGBA, GL_FLOAT, zero);
        glBindImageTexture(0, tex[0], 0, GL_FALSE, 0, GL_READ_ONLY, GL_RGBA32F);
        glBindImageTexture(1, tex[1], 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
        for (int row = 0; row < h; row++) {
            glUniform1i(1, row);
            glDispatchCompute((GLuint)((iw >> 6) + 1), (GLuint)ih, 1);
            glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
        }
        glFinish();
        float *out = (float *)malloc((size_t)iw * ih * 12);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGB, GL_FLOAT, out);
        FILE *f = fopen(argv[5], "wb");
        fwrite(out, 12, (size_t)iw * ih, f);
        fclose(f);
        fprintf(stderr, "irradiance: %dx%d -> %dx%d, gl error 0x%x\n", w, h, iw, ih, glGetError());
        return 0;
    }

    const int hits_mode = !strcmp(argv[1], "hits");
    char *src = read_file(REF_SHADER, NULL);
    char *use = src;
    if (hits_mode) {
        /* harness = the reference's text up to the end of findIntersection + a main() of our own */
        char *cut = strstr(src, "vec3 sRGBtoLinear");
        if (!cut) { fprintf(stderr, "cannot find the end of findIntersection in the reference shader\n"); return 20; }
        const char *tail =
            "\nlayout(std430, binding=7) restrict readonly buffer rayBuffer { float rays[]; };\n"
            "layout(std430, binding=8) restrict writeonly buffer idBuffer { int out_id[]; };\n"
            "layout(std430, binding=9) restrict writeonly buffer uvtBuffer { float out_uvt[]; };\n"
            "uniform layout(location=20) uint N_RAYS;\n"
            "void main(void) {\n"
            "    uint i = gl_GlobalInvocationID.y * (gl_NumWorkGroups.x * 8u) + gl_GlobalInvocationID.x;\n"
            "    if (i >= N_RAYS) return;\n"
            "    Ray ray;\n"
            "    ray.position = vec3(rays[6u*i], rays[6u*i+1u], rays[6u*i+2u]);\n"
            "    ray.direction = vec3(rays[6u*i+3u], rays[6u*i+4u], rays[6u*i+5u]);\n"
            "    vec3 sec;\n"
            "    out_id[i] = findIntersection(ray, sec);\n"
            "    out_uvt[3u*i] = sec.x; out_uvt[3u*i+1u] = sec.y; out_uvt[3u*i+2u] = sec.z;\n"
            "}\n";
        size_t head = (size_t)(cut - src);
        use = (char *)malloc(head + strlen(tail) + 1);
        memcpy(use, src, head);
        strcpy(use + head, tail);
    }
    GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
    const char *srcs[1] = {use};
    glShaderSource(sh, 1, srcs, NULL);
    glCompileShader(sh);
    GLint ok = 0;
    glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[8192]; glGetShaderInfoLog(sh, sizeof(log), NULL, log); fprintf(stderr, "compile failed:\n%s\n", log); return 21; }
    GLuint prog = glCreateProgram();
    glAttachShader(prog, sh);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { char log[8192]; glGetProgramInfoLog(prog, sizeof(log), NULL, log); fprintf(stderr, "link failed:\n%s\n", log); return 22; }
    glUseProgram(prog);

    struct scene sc = load_scene(argv[2]);
    GLuint bufs[8];
    glGenBuffers(8, bufs);
    /* SSBO bindings 1, 2, 3, 6 as in raytracer.glsl:50-64 */
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[0]); glBufferData(GL_SHADER_STORAGE_BUFFER, 48l * sc.count, sc.models, GL_STATIC_DRAW);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, bufs[0]);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[1]); glBufferData(GL_SHADER_STORAGE_BUFFER, 100l * sc.count, sc.shadings, GL_STATIC_DRAW);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, bufs[1]);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[2]); glBufferData(GL_SHADER_STORAGE_BUFFER, 48l * sc.n_mat, sc.materials, GL_STATIC_DRAW);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 3, bufs[2]);
    int *has_tex = (int *)calloc(sc.n_mat, sizeof(int));
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[3]); glBufferData(GL_SHADER_STORAGE_BUFFER, 4l * sc.n_mat, has_tex, GL_STATIC_DRAW);
    glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 6, bufs[3]);
    glUniform1i(5, (GLint)sc.count);        /* COUNT, location 5 */

    if (hits_mode) {
        const uint32_t n = (uint32_t)atoi(argv[4]);
        size_t rn;
        char *rays = read_file(argv[3], &rn);
        if (rn < 24ul * n) { fprintf(stderr, "ray file too short\n"); return 2; }
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[4]); glBufferData(GL_SHADER_STORAGE_BUFFER, 24l * n, rays, GL_STATIC_DRAW);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 7, bufs[4]);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[5]); glBufferData(GL_SHADER_STORAGE_BUFFER, 4l * n, NULL, GL_DYNAMIC_READ);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 8, bufs[5]);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[6]); glBufferData(GL_SHADER_STORAGE_BUFFER, 12l * n, NULL, GL_DYNAMIC_READ);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 9, bufs[6]);
        glUniform1ui(20, n);
        const uint32_t gx = 16, per_row = gx * 8;
        const uint32_t gy = ((n + per_row - 1) / per_row + 7) / 8;
        glDispatchCompute(gx, gy, 1);
        glMemoryBarrier(GL_ALL_BARRIER_BITS);
        glFinish();
        int *ids = (int *)malloc(4ul * n);
        float *uvt = (float *)malloc(12ul * n);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[5]); glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, 4l * n, ids);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[6]); glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, 12l * n, uvt);
        FILE *f = fopen(argv[5], "wb");
        fwrite(ids, 4, n, f);
        fwrite(uvt, 12, n, f);
        fclose(f);
        fprintf(stderr, "hits: %u rays, gl error 0x%x\n", n, glGetError());
        return 0;
    }

    /* image mode */
    const int ew = atoi(argv[4]), eh = atoi(argv[5]);
    const int W = atoi(argv[6]), H = atoi(argv[7]);
    const uint32_t clock_v = (uint32_t)strtoul(argv[8], NULL, 10);
    const int recursion = atoi(argv[9]), n_samples = atoi(argv[10]);
    size_t en;
    char *env = read_file(argv[3], &en);
    float cam[16];
    { size_t cn; char *c = read_file(argv[12], &cn); memcpy(cam, c, 64); }
    const int first = argc > 13 ? atoi(argv[13]) : 0;
    GLuint tex[3];
    glGenTextures(3, tex);
    /* render target: image binding 0, rgba32f (Scene.cpp:405-410, :459) */
    glBindTexture(GL_TEXTURE_2D, tex[0]);
    glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, W, H);
    glBindImageTexture(0, tex[0], 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
    /* environment: sampler uniform location 2 on unit 2 (Scene.cpp:126-133, :145-151) */
    glActiveTexture(GL_TEXTURE2);
    glBindTexture(GL_TEXTURE_2D, tex[1]);
    glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, ew, eh);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, ew, eh, GL_RGBA, GL_FLOAT, env);
    glUniform1i(2, 2);
    /* atlas: sampler uniform location 1 on unit 1; a 1-texel array is enough for untextured scenes */
    glActiveTexture(GL_TEXTURE1);
    glBindTexture(GL_TEXTURE_2D_ARRAY, tex[2]);
    {
        GLF(PFNGLTEXSUBIMAGE3DPROC, glTexSubImage3D);
        const char *atlas = argc > 14 ? read_file(argv[14], NULL) : NULL;
        uint32_t S = 1, n_up = 0;
        if (atlas) { memcpy(&S, atlas, 4); memcpy(&n_up, atlas + 4, 4); atlas += 8; }
        glTexStorage3D(GL_TEXTURE_2D_ARRAY, 1, GL_R11F_G11F_B10F, (GLsizei)S, (GLsizei)S, (GLsizei)(3 * sc.n_mat));
        /* storage contents are undefined: layers that are never uploaded read as zero */
        float *zero = (float *)calloc((size_t)S * S, 16);
        for (uint32_t l = 0; l < 3 * sc.n_mat; l++)
            glTexSubImage3D(GL_TEXTURE_2D_ARRAY, 0, 0, 0, (GLint)l, (GLsizei)S, (GLsizei)S, 1, GL_RGBA, GL_FLOAT, zero);
        free(zero);
        for (uint32_t k = 0; k < n_up; k++) {
            uint32_t hdr[4];
            memcpy(hdr, atlas, 16);
            atlas += 16;
            glTexSubImage3D(GL_TEXTURE_2D_ARRAY, 0, 0, 0, (GLint)(hdr[0] * 3u + hdr[1]), (GLsizei)hdr[2], (GLsizei)hdr[3], 1,
                            GL_RGBA, GL_FLOAT, atlas);
            atlas += 16ul * hdr[2] * hdr[3];
        }
        if (atlas) {
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, bufs[3]);
            glBufferData(GL_SHADER_STORAGE_BUFFER, 4l * sc.n_mat, atlas, GL_STATIC_DRAW);
        }
    }
    glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_WRAP_S, GL_REPEAT);
    glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_WRAP_T, GL_REPEAT);
    glUniform1i(1, 1);
    glUniform1i(3, 2);                       /* IRRADIANCE: declared, never sampled by live code */
    glUniformMatrix4fv(4, 1, GL_FALSE, cam); /* CAMERA */
    glUniform1i(6, (GLint)sc.lights);        /* LIGHTS */
    glUniform1i(7, recursion);               /* RECURSION */
    glUniform1ui(9, clock_v);                /* CLOCK */
    {   /* EXPOSURE: 1.0 as Scene.cpp:331 sets it, unless REF_EXPOSURE asks for another value */
        const char *ex = getenv("REF_EXPOSURE");
        glUniform1f(13, ex ? (float)atof(ex) : 1.0f);
    }
    {   /* glTexStorage2D leaves the contents undefined: clear the target */
        float *zero = (float *)calloc((size_t)W * H, 16);
        glBindTexture(GL_TEXTURE_2D, tex[0]);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, GL_FLOAT, zero);
        free(zero);
    }
    for (int s = first; s < first + n_samples; s++) {
        glUniform1ui(8, (GLuint)s);          /* SAMPLE */
        glDispatchCompute((GLuint)((W >> 3) + 1), (GLuint)((H >> 3) + 1), 1);   /* Scene.cpp:424-431 */
        glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
    }
    glFinish();
    float *img = (float *)malloc(16ul * W * H);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, tex[0]);
    glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, img);
    FILE *f = fopen(argv[11], "wb");
    fwrite(img, 16, (size_t)W * H, f);
    fclose(f);
    fprintf(stderr, "image: %dx%d, %d samples, gl error 0x%x\n", W, H, n_samples, glGetError());
    return 0;
}
