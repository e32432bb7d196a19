/* glref — runs a GLSL compute shader on Mesa's software rasteriser (llvmpipe) without a display.
 *
 * TEST INFRASTRUCTURE (oracle side).  This is the runner of `oracle/_ref`: the reference's own
 * compute shader, /root/reference/assets/shaders/brick_raytracer.comp, compiled by Mesa's GLSL compiler
 * and executed by llvmpipe on the host cores.  The shader source is never copied into this repo: the
 * recipe (oracle/ref_gl/recipe.py) reads it where it lies under /root/reference, applies the dialect
 * edits listed there in memory, hands the text to glref_compile() and keeps only Mesa's program binary
 * (serialised NIR, no source text) under oracle/_ref/.
 *
 * The image has Mesa 23.2.1 (libglapi.so.0, dri/swrast_dri.so) but no X server, no EGL and no OSMesa,
 * so the GL context is made through the driver's own DRI "swrast" loader interface
 * (/usr/include/GL/internal/dri_interface.h): createNewScreen2 -> createContextAttribs(GL 4.5 core) ->
 * bindContext on a dummy drawable.  GL entry points come from libglapi's _glapi_get_proc_address.
 *
 * What it replaces of the reference: vkCmdDispatch of the compute pipeline
 * (src/modules/voxel_rt/ComputePipeline.zig:547-550) with its descriptor set
 * (bindings 0..7, ComputePipeline.zig:120-215) and push constants (:488-505).
 *
 * One context per process, used from the thread that called glref_init().
 */
#include <dirent.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

#define GLREF_API __attribute__((visibility("default")))

/* ---- DRI swrast loader: a drawable nobody looks at ---- */
static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_get_drawable_info, .putImage = ld_put_image, .getImage = ld_get_image,
    .putImage2 = ld_put_image2, .getImage2 = ld_get_image2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader.base, NULL};

static char g_err[4096];
static int g_ready;
static const __DRIcoreExtension *g_core;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;

#define GLFN(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
GLFN(const GLubyte *, glGetString, GLenum);
GLFN(void, glGetIntegerv, GLenum, GLint *);
GLFN(GLenum, glGetError, void);
GLFN(GLuint, glCreateShader, GLenum);
GLFN(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLFN(void, glCompileShader, GLuint);
GLFN(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLFN(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(void, glDeleteShader, GLuint);
GLFN(GLuint, glCreateProgram, void);
GLFN(void, glAttachShader, GLuint, GLuint);
GLFN(void, glLinkProgram, GLuint);
GLFN(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLFN(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(void, glProgramParameteri, GLuint, GLenum, GLint);
GLFN(void, glGetProgramBinary, GLuint, GLsizei, GLsizei *, GLenum *, void *);
GLFN(void, glProgramBinary, GLuint, GLenum, const void *, GLsizei);
GLFN(void, glDeleteProgram, GLuint);
GLFN(void, glUseProgram, GLuint);
GLFN(void, glGenBuffers, GLsizei, GLuint *);
GLFN(void, glDeleteBuffers, GLsizei, const GLuint *);
GLFN(void, glBindBuffer, GLenum, GLuint);
GLFN(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLFN(void, glBufferSubData, GLenum, GLintptr, GLsizeiptr, const void *);
GLFN(void, glGetBufferSubData, GLenum, GLintptr, GLsizeiptr, void *);
GLFN(void, glBindBufferBase, GLenum, GLuint, GLuint);
GLFN(void, glGenTextures, GLsizei, GLuint *);
GLFN(void, glDeleteTextures, GLsizei, const GLuint *);
GLFN(void, glBindTexture, GLenum, GLuint);
GLFN(void, glTexStorage2D, GLenum, GLsizei, GLenum, GLsizei, GLsizei);
GLFN(void, glBindImageTexture, GLuint, GLuint, GLint, GLboolean, GLint, GLenum, GLenum);
GLFN(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void *);
GLFN(void, glPixelStorei, GLenum, GLint);
GLFN(void, glDispatchCompute, GLuint, GLuint, GLuint);
GLFN(void, glMemoryBarrier, GLbitfield);
GLFN(void, glFinish, void);
GLFN(void, glGenVertexArrays, GLsizei, GLuint *);
GLFN(void, glDeleteVertexArrays, GLsizei, const GLuint *);
GLFN(void, glBindVertexArray, GLuint);
GLFN(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
GLFN(void, glEnableVertexAttribArray, GLuint);
GLFN(void, glGenFramebuffers, GLsizei, GLuint *);
GLFN(void, glDeleteFramebuffers, GLsizei, const GLuint *);
GLFN(void, glBindFramebuffer, GLenum, GLuint);
GLFN(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLFN(GLenum, glCheckFramebufferStatus, GLenum);
GLFN(void, glViewport, GLint, GLint, GLsizei, GLsizei);
GLFN(void, glDrawArrays, GLenum, GLint, GLsizei);
GLFN(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
GLFN(void, glTexSubImage2D, GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *);
GLFN(void, glTexParameteri, GLenum, GLenum, GLint);
GLFN(void, glActiveTexture, GLenum);
GLFN(void, glDisable, GLenum);

GLREF_API const char *glref_last_error(void) { return g_err; }

static int fail(const char *fmt, const char *arg) {
    snprintf(g_err, sizeof g_err, fmt, arg ? arg : "");
    return -1;
}

/* Creates the llvmpipe GL 4.5 core context (idempotent).  0 on success. */
GLREF_API int glref_init(void) {
    if (g_ready) return 0;
    /* llvmpipe filters 8-bit textures with 8-bit fixed-point weights unless told otherwise; the present pass is compared at
       1e-4, so ask for its float path (a documented gallivm switch; harmless for the compute shader) */
    setenv("GALLIVM_PERF", "no_aos_sampling", 0);
    const char *paths[] = {"/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", "swrast_dri.so", NULL};
    void *drv = NULL;
    for (int i = 0; paths[i] && !drv; i++) drv = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!drv) return fail("dlopen swrast_dri.so: %s", dlerror());
    const __DRIextension **(*get_exts)(void) = (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) return fail("%s", "no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = get_exts();
    const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !sw || sw->base.version < 4) return fail("%s", "DRI_Core / DRI_SWRast v4 not offered by the driver");
    const __DRIconfig **configs = NULL;
    g_screen = sw->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) return fail("%s", "createNewScreen2 failed");
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned err = 0;
    g_ctx = sw->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) return fail("%s", "createContextAttribs(GL 4.5 core) failed");
    g_draw = sw->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw || !g_core->bindContext(g_ctx, g_draw, g_draw)) return fail("%s", "bindContext failed");
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return fail("dlopen libglapi.so.0: %s", dlerror());
    void *(*gpa)(const char *) = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) return fail("%s", "no _glapi_get_proc_address");
#define LOAD(name) do { *(void **)&p_##name = gpa(#name); if (!p_##name) return fail("GL entry point missing: %s", #name); } while (0)
    LOAD(glGetString); LOAD(glGetIntegerv); LOAD(glGetError); LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader);
    LOAD(glGetShaderiv); LOAD(glGetShaderInfoLog); LOAD(glDeleteShader); LOAD(glCreateProgram); LOAD(glAttachShader);
    LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog); LOAD(glProgramParameteri); LOAD(glGetProgramBinary);
    LOAD(glProgramBinary); LOAD(glDeleteProgram); LOAD(glUseProgram); LOAD(glGenBuffers); LOAD(glDeleteBuffers); LOAD(glBindBuffer);
    LOAD(glBufferData); LOAD(glBufferSubData); LOAD(glGetBufferSubData); LOAD(glBindBufferBase); LOAD(glGenTextures);
    LOAD(glDeleteTextures); LOAD(glBindTexture); LOAD(glTexStorage2D); LOAD(glBindImageTexture); LOAD(glGetTexImage);
    LOAD(glPixelStorei); LOAD(glDispatchCompute); LOAD(glMemoryBarrier); LOAD(glFinish);
    LOAD(glGenVertexArrays); LOAD(glDeleteVertexArrays); LOAD(glBindVertexArray); LOAD(glVertexAttribPointer); LOAD(glEnableVertexAttribArray);
    LOAD(glGenFramebuffers); LOAD(glDeleteFramebuffers); LOAD(glBindFramebuffer); LOAD(glFramebufferTexture2D); LOAD(glCheckFramebufferStatus);
    LOAD(glViewport); LOAD(glDrawArrays); LOAD(glReadPixels); LOAD(glTexSubImage2D); LOAD(glTexParameteri); LOAD(glActiveTexture); LOAD(glDisable);
#undef LOAD
    g_ready = 1;
    return 0;
}

/* "GL_VERSION | GL_RENDERER | GLSL version" of the context, for the fixtures' provenance record. */
GLREF_API const char *glref_info(void) {
    static char s[512];
    if (!g_ready) return "";
    snprintf(s, sizeof s, "%s | %s | GLSL %s", (const char *)p_glGetString(GL_VERSION), (const char *)p_glGetString(GL_RENDERER),
             (const char *)p_glGetString(GL_SHADING_LANGUAGE_VERSION));
    return s;
}

/* Compiles + links one compute shader.  Returns the program name (>0) or -1 (log in glref_last_error). */
GLREF_API int glref_compile(const char *src, int len) {
    if (!g_ready) return fail("%s", "glref_init not called");
    GLuint sh = p_glCreateShader(GL_COMPUTE_SHADER);
    GLint l = len;
    p_glShaderSource(sh, 1, &src, &l);
    p_glCompileShader(sh);
    GLint ok = 0;
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3800] = {0};
        p_glGetShaderInfoLog(sh, sizeof log - 1, NULL, log);
        p_glDeleteShader(sh);
        return fail("compile: %s", log);
    }
    GLuint prog = p_glCreateProgram();
    p_glProgramParameteri(prog, GL_PROGRAM_BINARY_RETRIEVABLE_HINT, GL_TRUE);
    p_glAttachShader(prog, sh);
    p_glLinkProgram(prog);
    p_glDeleteShader(sh);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3800] = {0};
        p_glGetProgramInfoLog(prog, sizeof log - 1, NULL, log);
        p_glDeleteProgram(prog);
        return fail("link: %s", log);
    }
    return (int)prog;
}

/* Compiles + links a vertex and a fragment shader (the reference's image.vert / image.frag).  Program name or -1. */
GLREF_API int glref_compile_raster(const char *vs, int vs_len, const char *fs, int fs_len) {
    if (!g_ready) return fail("%s", "glref_init not called");
    GLuint prog = p_glCreateProgram();
    p_glProgramParameteri(prog, GL_PROGRAM_BINARY_RETRIEVABLE_HINT, GL_TRUE);
    const char *srcs[2] = {vs, fs};
    const GLint lens[2] = {vs_len, fs_len};
    const GLenum kinds[2] = {GL_VERTEX_SHADER, GL_FRAGMENT_SHADER};
    for (int i = 0; i < 2; i++) {
        GLuint sh = p_glCreateShader(kinds[i]);
        p_glShaderSource(sh, 1, &srcs[i], &lens[i]);
        p_glCompileShader(sh);
        GLint ok = 0;
        p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) {
            char log[3800] = {0};
            p_glGetShaderInfoLog(sh, sizeof log - 1, NULL, log);
            p_glDeleteShader(sh);
            p_glDeleteProgram(prog);
            return fail(i ? "fragment shader: %s" : "vertex shader: %s", log);
        }
        p_glAttachShader(prog, sh);
        p_glDeleteShader(sh);
    }
    p_glLinkProgram(prog);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3800] = {0};
        p_glGetProgramInfoLog(prog, sizeof log - 1, NULL, log);
        p_glDeleteProgram(prog);
        return fail("link: %s", log);
    }
    return (int)prog;
}

/* The reference's present pass: one fullscreen quad (two triangles; attribute 0 = vec3 position, 1 = vec2 uv, as
 * image.vert declares them) drawn into an out_w x out_h RGBA32F target, sampling an RGBA8 UNORM image of img_w x img_h
 * through texture unit 0 with the reference's sampler state (linear min / mag, repeat; Pipeline.zig:194-211).  `ubo` goes to
 * uniform binding `ubo_binding` (the push constants).  uv (0,0) is placed at the first row / column of BOTH the sampled image
 * and `out`, so rows of `out` run top to bottom like the traced image's.  0 on success. */
GLREF_API int glref_present(int prog, const void *img_rgba8, int img_w, int img_h, int ubo_binding, const void *ubo, int64_t ubo_bytes, int out_w,
                            int out_h, float *out_rgba32f) {
    if (!g_ready) return fail("%s", "glref_init not called");
    while (p_glGetError() != GL_NO_ERROR) {}
    GLuint tex[2], fbo, vao, buf[2];
    p_glGenTextures(2, tex);
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_2D, tex[0]);
    p_glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA8, img_w, img_h);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, img_w, img_h, GL_RGBA, GL_UNSIGNED_BYTE, img_rgba8);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    p_glBindTexture(GL_TEXTURE_2D, tex[1]);
    p_glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, out_w, out_h);
    p_glGenFramebuffers(1, &fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex[1], 0);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("%s", "framebuffer incomplete");
    p_glBindTexture(GL_TEXTURE_2D, tex[0]);
    /* x, y, z, u, v: framebuffer row 0 (NDC y = -1) gets v = 0 */
    static const float quad[6][5] = {{-1, -1, 0, 0, 0}, {1, -1, 0, 1, 0}, {1, 1, 0, 1, 1}, {-1, -1, 0, 0, 0}, {1, 1, 0, 1, 1}, {-1, 1, 0, 0, 1}};
    p_glGenVertexArrays(1, &vao);
    p_glBindVertexArray(vao);
    p_glGenBuffers(2, buf);
    p_glBindBuffer(GL_ARRAY_BUFFER, buf[0]);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    p_glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (const void *)0);
    p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (const void *)(3 * sizeof(float)));
    p_glEnableVertexAttribArray(0);
    p_glEnableVertexAttribArray(1);
    p_glBindBuffer(GL_UNIFORM_BUFFER, buf[1]);
    p_glBufferData(GL_UNIFORM_BUFFER, (GLsizeiptr)ubo_bytes, ubo, GL_STATIC_DRAW);
    p_glBindBufferBase(GL_UNIFORM_BUFFER, (GLuint)ubo_binding, buf[1]);
    p_glViewport(0, 0, out_w, out_h);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    p_glDisable(GL_CULL_FACE);
    p_glUseProgram((GLuint)prog);
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glFinish();
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glReadPixels(0, 0, out_w, out_h, GL_RGBA, GL_FLOAT, out_rgba32f);
    GLenum e = p_glGetError();
    p_glUseProgram(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glBindVertexArray(0);
    p_glDeleteVertexArrays(1, &vao);
    p_glDeleteBuffers(2, buf);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(2, tex);
    if (e != GL_NO_ERROR) {
        snprintf(g_err, sizeof g_err, "GL error 0x%x during the present pass", e);
        return -1;
    }
    return 0;
}

/* Mesa's program binary (serialised NIR + metadata; loadable by the same Mesa build only).
 * Returns its size; copies it to `dst` when cap is large enough.  `format` receives the binary format enum. */
GLREF_API int64_t glref_get_binary(int prog, void *dst, int64_t cap, uint32_t *format) {
    GLint n = 0;
    p_glGetProgramiv((GLuint)prog, GL_PROGRAM_BINARY_LENGTH, &n);
    if (n <= 0) return fail("%s", "program has no binary");
    if (dst && cap >= n) {
        GLenum fmt = 0;
        GLsizei got = 0;
        p_glGetProgramBinary((GLuint)prog, n, &got, &fmt, dst);
        if (format) *format = fmt;
        if (got != n) return fail("%s", "glGetProgramBinary returned a short binary");
    }
    return n;
}

GLREF_API int glref_load_binary(const void *bin, int64_t n, uint32_t format) {
    if (!g_ready) return fail("%s", "glref_init not called");
    GLuint prog = p_glCreateProgram();
    p_glProgramBinary(prog, format, bin, (GLsizei)n);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        p_glDeleteProgram(prog);
        return fail("%s", "glProgramBinary rejected the binary (made by a different Mesa build?)");
    }
    return (int)prog;
}

GLREF_API void glref_delete_program(int prog) { if (g_ready && prog > 0) p_glDeleteProgram((GLuint)prog); }

/* ---- scene buffers: kept bound between dispatches (the reference uploads its buffers once and only pushes the
 * 128 constant bytes per frame, ComputePipeline.zig:488-505) ---- */
static GLuint g_bufs[24];
static int g_nbufs, g_nubo;
static int g_ubo_binding[8];
static GLuint g_tex;
static int g_tex_w, g_tex_h, g_tex_float = -1;

GLREF_API void glref_buffers_clear(void) {
    if (!g_ready) return;
    if (g_nbufs) p_glDeleteBuffers(g_nbufs, g_bufs);
    g_nbufs = g_nubo = 0;
    if (g_tex) p_glDeleteTextures(1, &g_tex);
    g_tex = 0;
    g_tex_float = -1;
}

/*   ubo[i]  : nubo uniform blocks   -> UBO binding point ubo_binding[i]
 *   ssbo[i] : nssbo storage blocks  -> SSBO binding point ssbo_binding[i]   (sizes in bytes; padded to 4 here) */
GLREF_API int glref_buffers_set(int nubo, const int *ubo_binding, const void *const *ubo, const int64_t *ubo_bytes, int nssbo,
                                const int *ssbo_binding, const void *const *ssbo, const int64_t *ssbo_bytes) {
    if (!g_ready) return fail("%s", "glref_init not called");
    if (nubo > 8 || nssbo > 16) return fail("%s", "too many buffers");
    glref_buffers_clear();
    while (p_glGetError() != GL_NO_ERROR) {}
    GLint max_ssbo = 0;
    p_glGetIntegerv(GL_MAX_SHADER_STORAGE_BLOCK_SIZE, &max_ssbo);
    for (int i = 0; i < nssbo; i++)
        if (ssbo_bytes[i] > (int64_t)max_ssbo) {
            snprintf(g_err, sizeof g_err, "storage block %d has %lld bytes, GL_MAX_SHADER_STORAGE_BLOCK_SIZE is %d", ssbo_binding[i],
                     (long long)ssbo_bytes[i], max_ssbo);
            return -2;
        }
    g_nbufs = nubo + nssbo;
    g_nubo = nubo;
    p_glGenBuffers(g_nbufs, g_bufs);
    for (int i = 0; i < nubo; i++) {
        g_ubo_binding[i] = ubo_binding[i];
        p_glBindBuffer(GL_UNIFORM_BUFFER, g_bufs[i]);
        p_glBufferData(GL_UNIFORM_BUFFER, (GLsizeiptr)ubo_bytes[i], ubo[i], GL_DYNAMIC_DRAW);
        p_glBindBufferBase(GL_UNIFORM_BUFFER, (GLuint)ubo_binding[i], g_bufs[i]);
    }
    for (int i = 0; i < nssbo; i++) {
        int64_t n = ssbo_bytes[i], padded = (n + 3) & ~(int64_t)3;
        if (padded == 0) padded = 4;
        p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, g_bufs[nubo + i]);
        p_glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)padded, NULL, GL_STATIC_DRAW);
        if (n) p_glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)n, ssbo[i]);
        if (padded > n) { const uint32_t z = 0; p_glBufferSubData(GL_SHADER_STORAGE_BUFFER, (GLintptr)n, (GLsizeiptr)(padded - n), &z); }
        p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)ssbo_binding[i], g_bufs[nubo + i]);
    }
    if (p_glGetError() != GL_NO_ERROR) return fail("%s", "GL error while creating the scene buffers");
    return 0;
}

/* New contents for the uniform block at `binding` (the per-frame push constants). */
GLREF_API int glref_ubo_update(int binding, const void *data, int64_t bytes) {
    for (int i = 0; i < g_nubo; i++)
        if (g_ubo_binding[i] == binding) {
            p_glBindBuffer(GL_UNIFORM_BUFFER, g_bufs[i]);
            p_glBufferSubData(GL_UNIFORM_BUFFER, 0, (GLsizeiptr)bytes, data);
            return 0;
        }
    return fail("%s", "no uniform block at that binding");
}

/* One dispatch over a width x height image with the buffers of glref_buffers_set:
 *   image unit 0 : a width x height texture, RGBA8 (out_float = 0) or RGBA32F (out_float = 1), write-only
 *   groups = ceil(width / wg_x) x ceil(height / wg_y) x 1   (ComputePipeline.zig:547-550)
 * Returns when the dispatch has finished (glFinish).  `out` (may be NULL: timing runs) receives the image rows top to
 * bottom (y ascending), 4 or 16 bytes per pixel.  0 on success. */
GLREF_API int glref_run(int prog, int width, int height, int wg_x, int wg_y, int out_float, void *out) {
    if (!g_ready) return fail("%s", "glref_init not called");
    while (p_glGetError() != GL_NO_ERROR) {}
    if (!g_tex || g_tex_w != width || g_tex_h != height || g_tex_float != out_float) {
        if (g_tex) p_glDeleteTextures(1, &g_tex);
        p_glGenTextures(1, &g_tex);
        p_glBindTexture(GL_TEXTURE_2D, g_tex);
        p_glTexStorage2D(GL_TEXTURE_2D, 1, out_float ? GL_RGBA32F : GL_RGBA8, width, height);
        g_tex_w = width, g_tex_h = height, g_tex_float = out_float;
    }
    p_glBindTexture(GL_TEXTURE_2D, g_tex);
    p_glBindImageTexture(0, g_tex, 0, GL_FALSE, 0, GL_WRITE_ONLY, out_float ? GL_RGBA32F : GL_RGBA8);
    p_glUseProgram((GLuint)prog);
    p_glDispatchCompute((GLuint)((width + wg_x - 1) / wg_x), (GLuint)((height + wg_y - 1) / wg_y), 1);
    p_glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    if (out) {
        p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
        p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, out_float ? GL_FLOAT : GL_UNSIGNED_BYTE, out);
    }
    GLenum e = p_glGetError();
    p_glUseProgram(0);
    if (e != GL_NO_ERROR) {
        snprintf(g_err, sizeof g_err, "GL error 0x%x during dispatch", e);
        return -1;
    }
    return 0;
}

/* buffers_set + run + buffers_clear in one call */
GLREF_API int glref_dispatch(int prog, int width, int height, int wg_x, int wg_y, int nubo, const int *ubo_binding,
                             const void *const *ubo, const int64_t *ubo_bytes, int nssbo, const int *ssbo_binding,
                             const void *const *ssbo, const int64_t *ssbo_bytes, int out_float, void *out) {
    int rc = glref_buffers_set(nubo, ubo_binding, ubo, ubo_bytes, nssbo, ssbo_binding, ssbo, ssbo_bytes);
    if (rc == 0) rc = glref_run(prog, width, height, wg_x, wg_y, out_float, out);
    glref_buffers_clear();
    return rc;
}

/* number of llvmpipe worker threads of this process (threads named "llvmpipe-N"), for the baseline's `cores` */
GLREF_API int glref_worker_threads(void) {
    int n = 0;
    DIR *d = opendir("/proc/self/task");
    if (!d) return 0;
    struct dirent *e;
    while ((e = readdir(d)) != NULL) {
        if (e->d_name[0] == '.') continue;
        char path[320], name[64] = {0};
        snprintf(path, sizeof path, "/proc/self/task/%s/comm", e->d_name);
        FILE *f = fopen(path, "r");
        if (!f) continue;
        if (fgets(name, sizeof name, f) && !strncmp(name, "llvmpipe-", 9)) n++;
        fclose(f);
    }
    closedir(d);
    return n;
}
