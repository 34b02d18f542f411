/*
 * harness.c — runs the REFERENCE's GLSL (shader/camera.fs, shader/tracer.fs,
 * read from /root/reference at run time, never copied) on the SwiftShader
 * GLES 3.0 software rasteriser that ships inside the `kaleido` wheel.
 *
 * Build-container tooling only: it produces the golden vectors under
 * tests/golden/ that pin oracle/ (tools/make_glsl_goldens.py drives it through
 * ctypes).  Mirrors the reference's resource setup and draw calls:
 *   createTexture / initBVH uploads   main.js:408-437, 562-577
 *   createEnvironmentMapImg           main.js:170-180
 *   initAtlas                         main.js:548-560
 *   initBuffers                       main.js:598-617
 *   drawCamera / drawTracer           main.js:741-807
 * No GL headers are installed: entry points are declared by hand and resolved
 * with dlsym.  Work-around for the SwiftShader 4.1 quad-divergence defect
 * (SURVEY.md App. B.4): screen targets are 2W x 2H and all 4 lanes of a 2x2
 * quad trace the same logical pixel; read-back checks the replicas agree.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned int GLenum, GLuint, GLbitfield;
typedef int GLint, GLsizei;
typedef float GLfloat;
typedef unsigned char GLboolean;
typedef char GLchar;
typedef ptrdiff_t GLsizeiptr;
typedef void *EGLDisplay, *EGLConfig, *EGLSurface, *EGLContext;
typedef int EGLint;
typedef unsigned int EGLBoolean;

#define GL_TEXTURE_2D 0x0DE1
#define GL_TEXTURE_2D_ARRAY 0x8C1A
#define GL_TEXTURE_WRAP_S 0x2802
#define GL_TEXTURE_WRAP_T 0x2803
#define GL_TEXTURE_MIN_FILTER 0x2801
#define GL_TEXTURE_MAG_FILTER 0x2800
#define GL_NEAREST 0x2600
#define GL_LINEAR 0x2601
#define GL_CLAMP_TO_EDGE 0x812F
#define GL_REPEAT 0x2901
#define GL_RGB32F 0x8815
#define GL_RGBA32F 0x8814
#define GL_RG32F 0x8230
#define GL_RGB 0x1907
#define GL_RGBA 0x1908
#define GL_RG 0x8227
#define GL_FLOAT 0x1406
#define GL_UNSIGNED_BYTE 0x1401
#define GL_TEXTURE0 0x84C0
#define GL_FRAMEBUFFER 0x8D40
#define GL_COLOR_ATTACHMENT0 0x8CE0
#define GL_FRAMEBUFFER_COMPLETE 0x8CD5
#define GL_VERTEX_SHADER 0x8B31
#define GL_FRAGMENT_SHADER 0x8B30
#define GL_COMPILE_STATUS 0x8B81
#define GL_LINK_STATUS 0x8B82
#define GL_ARRAY_BUFFER 0x8892
#define GL_STATIC_DRAW 0x88E4
#define GL_TRIANGLES 0x0004
#define GL_COLOR_BUFFER_BIT 0x4000
#define GL_UNPACK_ALIGNMENT 0x0CF5
#define GL_PACK_ALIGNMENT 0x0D05
#define GL_BLEND 0x0BE2
#define GL_RENDERER 0x1F01
#define GL_VERSION 0x1F02
#define GL_EXTENSIONS 0x1F03

#define EGL_SURFACE_TYPE 0x3033
#define EGL_PBUFFER_BIT 0x0001
#define EGL_RENDERABLE_TYPE 0x3040
#define EGL_OPENGL_ES3_BIT 0x0040
#define EGL_NONE 0x3038
#define EGL_WIDTH 0x3057
#define EGL_HEIGHT 0x3056
#define EGL_CONTEXT_CLIENT_VERSION 0x3098
#define EGL_OPENGL_ES_API 0x30A0

#define FN(ret, name, args) static ret(*name) args
FN(EGLDisplay, eglGetDisplay, (void *));
FN(EGLBoolean, eglInitialize, (EGLDisplay, EGLint *, EGLint *));
FN(EGLBoolean, eglBindAPI, (unsigned int));
FN(EGLBoolean, eglChooseConfig, (EGLDisplay, const EGLint *, EGLConfig *, EGLint, EGLint *));
FN(EGLSurface, eglCreatePbufferSurface, (EGLDisplay, EGLConfig, const EGLint *));
FN(EGLContext, eglCreateContext, (EGLDisplay, EGLConfig, EGLContext, const EGLint *));
FN(EGLBoolean, eglMakeCurrent, (EGLDisplay, EGLSurface, EGLSurface, EGLContext));
FN(EGLint, eglGetError, (void));
FN(const unsigned char *, glGetString, (GLenum));
FN(GLenum, glGetError, (void));
FN(void, glPixelStorei, (GLenum, GLint));
FN(void, glGenTextures, (GLsizei, GLuint *));
FN(void, glBindTexture, (GLenum, GLuint));
FN(void, glTexParameteri, (GLenum, GLenum, GLint));
FN(void, glTexImage2D, (GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *));
FN(void, glTexImage3D, (GLenum, GLint, GLint, GLsizei, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *));
FN(void, glActiveTexture, (GLenum));
FN(void, glGenFramebuffers, (GLsizei, GLuint *));
FN(void, glBindFramebuffer, (GLenum, GLuint));
FN(void, glFramebufferTexture2D, (GLenum, GLenum, GLenum, GLuint, GLint));
FN(void, glDrawBuffers, (GLsizei, const GLenum *));
FN(void, glReadBuffer, (GLenum));
FN(GLenum, glCheckFramebufferStatus, (GLenum));
FN(GLuint, glCreateShader, (GLenum));
FN(void, glShaderSource, (GLuint, GLsizei, const GLchar *const *, const GLint *));
FN(void, glCompileShader, (GLuint));
FN(void, glGetShaderiv, (GLuint, GLenum, GLint *));
FN(void, glGetShaderInfoLog, (GLuint, GLsizei, GLsizei *, GLchar *));
FN(GLuint, glCreateProgram, (void));
FN(void, glAttachShader, (GLuint, GLuint));
FN(void, glLinkProgram, (GLuint));
FN(void, glGetProgramiv, (GLuint, GLenum, GLint *));
FN(void, glGetProgramInfoLog, (GLuint, GLsizei, GLsizei *, GLchar *));
FN(void, glUseProgram, (GLuint));
FN(GLint, glGetUniformLocation, (GLuint, const GLchar *));
FN(GLint, glGetAttribLocation, (GLuint, const GLchar *));
FN(void, glUniform1i, (GLint, GLint));
FN(void, glUniform1ui, (GLint, GLuint));
FN(void, glUniform1f, (GLint, GLfloat));
FN(void, glUniform2fv, (GLint, GLsizei, const GLfloat *));
FN(void, glUniform3fv, (GLint, GLsizei, const GLfloat *));
FN(void, glUniform4uiv, (GLint, GLsizei, const GLuint *));
FN(void, glGenBuffers, (GLsizei, GLuint *));
FN(void, glBindBuffer, (GLenum, GLuint));
FN(void, glBufferData, (GLenum, GLsizeiptr, const void *, GLenum));
FN(void, glVertexAttribPointer, (GLuint, GLint, GLenum, GLboolean, GLsizei, const void *));
FN(void, glEnableVertexAttribArray, (GLuint));
FN(void, glViewport, (GLint, GLint, GLsizei, GLsizei));
FN(void, glDrawArrays, (GLenum, GLint, GLsizei));
FN(void, glClearColor, (GLfloat, GLfloat, GLfloat, GLfloat));
FN(void, glClear, (GLbitfield));
FN(void, glFinish, (void));
FN(void, glReadPixels, (GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *));
FN(void, glDisable, (GLenum));
FN(void, glDeleteTextures, (GLsizei, const GLuint *));
FN(void, glDeleteProgram, (GLuint));

static char g_err[8192];
const char *gh_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return -1; } while (0)
#define GLCHK(what) do { GLenum e_ = glGetError(); if (e_) FAIL("%s: GL error 0x%x", what, e_); } while (0)

static EGLDisplay g_dpy;
static int g_W, g_H, g_rep = 2; /* g_rep: quad replication factor of the screen targets */
static GLuint t_bvh, t_tri, t_mat, t_norm, t_uv, t_light, t_env, t_atlas;
static GLuint t_screen[2], t_campos, t_camdir;
static GLuint f_screen[2], f_camera;
static GLuint p_camera, p_tracer;
static GLuint g_vbo;

#define LOAD(lib, name) do { *(void **)(&name) = dlsym(lib, #name); if (!name) FAIL("missing symbol %s", #name); } while (0)

int gh_init(const char *dir) {
  char path[1024];
  snprintf(path, sizeof(path), "%s/libGLESv2.so", dir);
  void *gles = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!gles) FAIL("dlopen %s: %s", path, dlerror());
  snprintf(path, sizeof(path), "%s/libEGL.so", dir);
  void *egl = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!egl) FAIL("dlopen %s: %s", path, dlerror());
  LOAD(egl, eglGetDisplay); LOAD(egl, eglInitialize); LOAD(egl, eglBindAPI); LOAD(egl, eglChooseConfig);
  LOAD(egl, eglCreatePbufferSurface); LOAD(egl, eglCreateContext); LOAD(egl, eglMakeCurrent); LOAD(egl, eglGetError);
  LOAD(gles, glGetString); LOAD(gles, glGetError); LOAD(gles, glPixelStorei); LOAD(gles, glGenTextures);
  LOAD(gles, glBindTexture); LOAD(gles, glTexParameteri); LOAD(gles, glTexImage2D); LOAD(gles, glTexImage3D);
  LOAD(gles, glActiveTexture); LOAD(gles, glGenFramebuffers); LOAD(gles, glBindFramebuffer);
  LOAD(gles, glFramebufferTexture2D); LOAD(gles, glDrawBuffers); LOAD(gles, glReadBuffer);
  LOAD(gles, glCheckFramebufferStatus); LOAD(gles, glCreateShader); LOAD(gles, glShaderSource);
  LOAD(gles, glCompileShader); LOAD(gles, glGetShaderiv); LOAD(gles, glGetShaderInfoLog); LOAD(gles, glCreateProgram);
  LOAD(gles, glAttachShader); LOAD(gles, glLinkProgram); LOAD(gles, glGetProgramiv); LOAD(gles, glGetProgramInfoLog);
  LOAD(gles, glUseProgram); LOAD(gles, glGetUniformLocation); LOAD(gles, glGetAttribLocation); LOAD(gles, glUniform1i);
  LOAD(gles, glUniform1ui); LOAD(gles, glUniform1f); LOAD(gles, glUniform2fv); LOAD(gles, glUniform3fv);
  LOAD(gles, glUniform4uiv); LOAD(gles, glGenBuffers); LOAD(gles, glBindBuffer); LOAD(gles, glBufferData);
  LOAD(gles, glVertexAttribPointer); LOAD(gles, glEnableVertexAttribArray); LOAD(gles, glViewport);
  LOAD(gles, glDrawArrays); LOAD(gles, glClearColor); LOAD(gles, glClear); LOAD(gles, glFinish);
  LOAD(gles, glReadPixels); LOAD(gles, glDisable); LOAD(gles, glDeleteTextures); LOAD(gles, glDeleteProgram);

  g_dpy = eglGetDisplay((void *)0xFACE1E55); /* SwiftShader headless display */
  if (!g_dpy) g_dpy = eglGetDisplay(NULL);
  EGLint maj, min;
  if (!eglInitialize(g_dpy, &maj, &min)) FAIL("eglInitialize failed 0x%x", eglGetError());
  eglBindAPI(EGL_OPENGL_ES_API);
  EGLint cfg_attr[] = {EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_NONE};
  EGLConfig cfg; EGLint ncfg = 0;
  if (!eglChooseConfig(g_dpy, cfg_attr, &cfg, 1, &ncfg) || ncfg < 1) FAIL("eglChooseConfig failed");
  EGLint pb_attr[] = {EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE};
  EGLSurface surf = eglCreatePbufferSurface(g_dpy, cfg, pb_attr);
  if (!surf) FAIL("eglCreatePbufferSurface failed 0x%x", eglGetError());
  EGLint ctx_attr[] = {EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE};
  EGLContext ctx = eglCreateContext(g_dpy, cfg, NULL, ctx_attr);
  if (!ctx) FAIL("eglCreateContext failed 0x%x", eglGetError());
  if (!eglMakeCurrent(g_dpy, surf, surf, ctx)) FAIL("eglMakeCurrent failed 0x%x", eglGetError());
  glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
  glPixelStorei(GL_PACK_ALIGNMENT, 1);
  glDisable(GL_BLEND); /* main.js:909 */
  float verts[] = {-1.0f, 3.0f, 0.0f, 3.0f, -1.0f, 0.0f, -1.0f, -1.0f, 0.0f}; /* main.js:601-605 */
  glGenBuffers(1, &g_vbo);
  glBindBuffer(GL_ARRAY_BUFFER, g_vbo);
  glBufferData(GL_ARRAY_BUFFER, sizeof(verts), verts, GL_STATIC_DRAW);
  GLCHK("init");
  return 0;
}

const char *gh_renderer(void) {
  static char buf[512];
  snprintf(buf, sizeof(buf), "%s | %s", (const char *)glGetString(GL_RENDERER), (const char *)glGetString(GL_VERSION));
  return buf;
}
const char *gh_extensions(void) { return (const char *)glGetString(GL_EXTENSIONS); }

/* createTexture's sampler state (main.js:570-574) */
static GLuint data_texture(GLint internal, GLenum fmt, int w, int h, const void *data) {
  GLuint t;
  glGenTextures(1, &t);
  glBindTexture(GL_TEXTURE_2D, t);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, GL_FLOAT, data);
  return t;
}

/* All data arrays are already padded by the caller exactly as padBuffer does (main.js:143-154). */
int gh_scene(const float *bvh, int bvh_w, int bvh_h, const float *tri, int tri_w, int tri_h, const float *mat,
             int mat_w, int mat_h, const float *norm, int norm_w, int norm_h, const float *uv, int uv_w, int uv_h,
             const uint8_t *atlas, int atlas_res, int atlas_layers, const uint8_t *env, int env_w, int env_h) {
  t_bvh = data_texture(GL_RGB32F, GL_RGB, bvh_w, bvh_h, bvh);
  t_mat = data_texture(GL_RGB32F, GL_RGB, mat_w, mat_h, mat);
  t_tri = data_texture(GL_RGB32F, GL_RGB, tri_w, tri_h, tri);
  t_norm = data_texture(GL_RGB32F, GL_RGB, norm_w, norm_h, norm);
  float dummy[3] = {-1, -1, -1};
  t_light = data_texture(GL_RGB32F, GL_RGB, 1, 1, dummy);
  t_uv = data_texture(GL_RG32F, GL_RG, uv_w, uv_h, uv);
  GLCHK("data textures");
  /* createEnvironmentMapImg (main.js:170-180) */
  glGenTextures(1, &t_env);
  glBindTexture(GL_TEXTURE_2D, t_env);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, env_w, env_h, 0, GL_RGBA, GL_UNSIGNED_BYTE, env);
  GLCHK("env texture");
  /* initAtlas (main.js:548-560) */
  glGenTextures(1, &t_atlas);
  glBindTexture(GL_TEXTURE_2D_ARRAY, t_atlas);
  glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_WRAP_S, GL_REPEAT);
  glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_WRAP_T, GL_REPEAT);
  glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
  glTexParameteri(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
  glTexImage3D(GL_TEXTURE_2D_ARRAY, 0, GL_RGBA, atlas_res, atlas_res, atlas_layers, 0, GL_RGBA, GL_UNSIGNED_BYTE, atlas);
  GLCHK("atlas texture");
  return 0;
}

static int compile(GLenum type, const char *src, GLuint *out) {
  GLuint s = glCreateShader(type);
  glShaderSource(s, 1, &src, NULL);
  glCompileShader(s);
  GLint ok = 0;
  glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[6000]; GLsizei n = 0;
    glGetShaderInfoLog(s, sizeof(log), &n, log);
    FAIL("shader compile failed: %.*s", (int)n, log);
  }
  *out = s;
  return 0;
}
static int link_program(const char *vs, const char *fs, GLuint *out) {
  GLuint v, f;
  if (compile(GL_VERTEX_SHADER, vs, &v)) return -1;
  if (compile(GL_FRAGMENT_SHADER, fs, &f)) return -1;
  GLuint p = glCreateProgram();
  glAttachShader(p, v);
  glAttachShader(p, f);
  glLinkProgram(p);
  GLint ok = 0;
  glGetProgramiv(p, GL_LINK_STATUS, &ok);
  if (!ok) {
    char log[6000]; GLsizei n = 0;
    glGetProgramInfoLog(p, sizeof(log), &n, log);
    FAIL("program link failed: %.*s", (int)n, log);
  }
  *out = p;
  return 0;
}

int gh_camera_program(const char *vs, const char *fs) { return link_program(vs, fs, &p_camera); }
/* may be called repeatedly with different (instrumented) fragment sources */
int gh_tracer_program(const char *vs, const char *fs) {
  /* old programs are leaked on purpose: SwiftShader 4.1 crashes when a program that was
     current is deleted and a new one is drawn with */
  p_tracer = 0;
  return link_program(vs, fs, &p_tracer);
}

static GLuint rgba32f_target(int w, int h) { return data_texture(GL_RGBA32F, GL_RGBA, w, h, NULL); }

/* initBuffers (main.js:598-617); rep = 2 enables the quad-replication work-around, 1 = native */
int gh_target(int W, int H, int rep) {
  g_W = W; g_H = H; g_rep = rep;
  for (int i = 0; i < 2; ++i) {
    t_screen[i] = rgba32f_target(W * rep, H * rep);
    glGenFramebuffers(1, &f_screen[i]);
    glBindFramebuffer(GL_FRAMEBUFFER, f_screen[i]);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, t_screen[i], 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("screen FBO incomplete");
  }
  t_campos = rgba32f_target(W, H);
  t_camdir = rgba32f_target(W, H);
  glGenFramebuffers(1, &f_camera);
  glBindFramebuffer(GL_FRAMEBUFFER, f_camera);
  glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, t_campos, 0);
  glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + 1, GL_TEXTURE_2D, t_camdir, 0);
  GLenum bufs[2] = {GL_COLOR_ATTACHMENT0, GL_COLOR_ATTACHMENT0 + 1};
  glDrawBuffers(2, bufs);
  if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("camera FBO incomplete");
  GLCHK("targets");
  return 0;
}

/* clear() (main.js:826-836) */
int gh_clear(void) {
  glClearColor(0.0f, 0.0f, 0.0f, 1.0f); /* main.js:908 */
  for (int i = 0; i < 2; ++i) {
    glBindFramebuffer(GL_FRAMEBUFFER, f_screen[i]);
    glClearColor(0.0f, 0.0f, 0.0f, 0.0f);
    glClear(GL_COLOR_BUFFER_BIT);
  }
  GLCHK("clear");
  return 0;
}

static void bind_corner(GLuint prog) {
  GLint loc = glGetAttribLocation(prog, "corner");
  glBindBuffer(GL_ARRAY_BUFFER, g_vbo);
  glVertexAttribPointer((GLuint)loc, 3, GL_FLOAT, 0, 0, 0);
  glEnableVertexAttribArray((GLuint)loc);
}

/* drawCamera (main.js:741-756) */
int gh_draw_camera(const float P[3], const float I[3], float fovScale, const float lens[2], float randBase) {
  glUseProgram(p_camera);
  glViewport(0, 0, g_W, g_H);
  bind_corner(p_camera);
  glUniform1f(glGetUniformLocation(p_camera, "fovScale"), fovScale);
  glUniform1f(glGetUniformLocation(p_camera, "randBase"), randBase);
  glUniform2fv(glGetUniformLocation(p_camera, "lensFeatures"), 1, lens);
  float res[2] = {(float)g_W, (float)g_H};
  glUniform2fv(glGetUniformLocation(p_camera, "resolution"), 1, res);
  glUniform3fv(glGetUniformLocation(p_camera, "P"), 1, P);
  glUniform3fv(glGetUniformLocation(p_camera, "I"), 1, I);
  glBindTexture(GL_TEXTURE_2D, 0);
  glBindFramebuffer(GL_FRAMEBUFFER, f_camera);
  glDrawArrays(GL_TRIANGLES, 0, 3);
  glFinish();
  GLCHK("drawCamera");
  return 0;
}

static int read_tex(GLuint fbo, int att, int w, int h, float *out) {
  glBindFramebuffer(GL_FRAMEBUFFER, fbo);
  glReadBuffer(GL_COLOR_ATTACHMENT0 + att);
  glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out);
  GLCHK("readPixels");
  return 0;
}
int gh_read_camera(float *pos, float *dir) {
  if (read_tex(f_camera, 0, g_W, g_H, pos)) return -1;
  return read_tex(f_camera, 1, g_W, g_H, dir);
}
/* inject ray textures (for staged parity: feed both sides the same rays) */
int gh_set_camera(const float *pos, const float *dir) {
  glBindTexture(GL_TEXTURE_2D, t_campos);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, g_W, g_H, 0, GL_RGBA, GL_FLOAT, pos);
  glBindTexture(GL_TEXTURE_2D, t_camdir);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, g_W, g_H, 0, GL_RGBA, GL_FLOAT, dir);
  GLCHK("set_camera");
  return 0;
}

/* extra int uniform of an instrumented probe main */
int gh_set_int(const char *name, int value) {
  glUseProgram(p_tracer);
  GLint loc = glGetUniformLocation(p_tracer, name);
  if (loc < 0) FAIL("uniform %s not active", name);
  glUniform1i(loc, value);
  GLCHK("set_int");
  return 0;
}

/* drawTracer(i) (main.js:758-807) */
int gh_draw_tracer(unsigned int tick, float randBase, float envTheta, const unsigned int *bins, int n_bins) {
  GLuint p = p_tracer;
  glUseProgram(p);
  glViewport(0, 0, g_W * g_rep, g_H * g_rep);
  bind_corner(p);
  const char *names[11] = {"fbTex", "triTex", "bvhTex", "matTex", "normTex", "lightTex", "uvTex", "envTex",
                           "cameraPosTex", "cameraDirTex", "texArray"};
  GLint loc;
  for (int i = 0; i < 11; ++i) { loc = glGetUniformLocation(p, names[i]); if (loc >= 0) glUniform1i(loc, i); }
  /* instrumented probe mains leave some uniforms inactive (location -1) */
  if ((loc = glGetUniformLocation(p, "tick")) >= 0) glUniform1ui(loc, tick);
  if ((loc = glGetUniformLocation(p, "numLights")) >= 0) glUniform1f(loc, 0.0f);
  if ((loc = glGetUniformLocation(p, "randBase")) >= 0) glUniform1f(loc, randBase);
  if ((loc = glGetUniformLocation(p, "envTheta")) >= 0) glUniform1f(loc, envTheta);
  if ((loc = glGetUniformLocation(p, "radianceBins")) >= 0) glUniform4uiv(loc, n_bins, bins);
  GLuint tex2d[10] = {t_screen[(tick + 1) % 2], t_tri, t_bvh, t_mat, t_norm, t_light, t_uv, t_env, t_campos, t_camdir};
  for (int i = 0; i < 10; ++i) {
    glActiveTexture(GL_TEXTURE0 + i);
    glBindTexture(GL_TEXTURE_2D, tex2d[i]);
  }
  glActiveTexture(GL_TEXTURE0 + 10);
  glBindTexture(GL_TEXTURE_2D_ARRAY, t_atlas);
  glBindFramebuffer(GL_FRAMEBUFFER, f_screen[tick % 2]);
  glDrawArrays(GL_TRIANGLES, 0, 3);
  glFinish();
  glActiveTexture(GL_TEXTURE0);
  GLCHK("drawTracer");
  return 0;
}

/* Reads screen[which]; with replication, subsamples and returns the number of
 * logical pixels whose 4 replicas are not bit-equal (expected 0). */
int gh_read_screen(int which, float *out /* W*H*4 */) {
  int rw = g_W * g_rep, rh = g_H * g_rep;
  float *tmp = (float *)malloc((size_t)rw * rh * 16);
  if (read_tex(f_screen[which], 0, rw, rh, tmp)) { free(tmp); return -1; }
  int mismatches = 0;
  for (int y = 0; y < g_H; ++y)
    for (int x = 0; x < g_W; ++x) {
      const float *p0 = tmp + ((size_t)(y * g_rep) * rw + x * g_rep) * 4;
      memcpy(out + ((size_t)y * g_W + x) * 4, p0, 16);
      int bad = 0;
      for (int dy = 0; dy < g_rep; ++dy)
        for (int dx = 0; dx < g_rep; ++dx) {
          const float *q = tmp + ((size_t)(y * g_rep + dy) * rw + x * g_rep + dx) * 4;
          if (memcmp(p0, q, 16) != 0) bad = 1;
        }
      mismatches += bad;
    }
  free(tmp);
  return mismatches;
}

/* drawQuad (main.js:809-824): draw.fs over an RGBA32F buffer into an RGBA8 target (the canvas) */
static GLuint p_draw;
int gh_draw_program(const char *vs, const char *fs) { return link_program(vs, fs, &p_draw); }
int gh_draw(const float *acc, int W, int H, float exposure, float saturation, float scale, float maxSigma, int denoise,
            unsigned char *out) {
  GLuint src = data_texture(GL_RGBA32F, GL_RGBA, W, H, acc);
  GLuint dst, fbo;
  glGenTextures(1, &dst);
  glBindTexture(GL_TEXTURE_2D, dst);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, W, H, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
  glGenFramebuffers(1, &fbo);
  glBindFramebuffer(GL_FRAMEBUFFER, fbo);
  glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, dst, 0);
  if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("draw FBO incomplete");
  glUseProgram(p_draw);
  glViewport(0, 0, W, H);
  bind_corner(p_draw);
  GLint loc;
  if ((loc = glGetUniformLocation(p_draw, "maxSigma")) >= 0) glUniform1f(loc, maxSigma);
  if ((loc = glGetUniformLocation(p_draw, "saturation")) >= 0) glUniform1f(loc, saturation);
  if ((loc = glGetUniformLocation(p_draw, "exposure")) >= 0) glUniform1f(loc, exposure);
  if ((loc = glGetUniformLocation(p_draw, "denoise")) >= 0) glUniform1i(loc, denoise);
  if ((loc = glGetUniformLocation(p_draw, "scale")) >= 0) glUniform1f(loc, scale);
  if ((loc = glGetUniformLocation(p_draw, "fbTex")) >= 0) glUniform1i(loc, 0);
  glActiveTexture(GL_TEXTURE0);
  glBindTexture(GL_TEXTURE_2D, src);
  glDrawArrays(GL_TRIANGLES, 0, 3);
  glFinish();
  glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, out);
  GLCHK("draw");
  return 0;
}

/* WebGLTextureWriter.setAndDrawTexture + getPixels (texture_packer.js:159-185): resample one source image to
 * res x res through the writer's own shader. */
#define GL_SRGB8_ALPHA8 0x8C43
static GLuint p_writer;
int gh_writer_program(const char *vs, const char *fs) { return link_program(vs, fs, &p_writer); }
int gh_write_texture(const unsigned char *src, int w, int h, int corrected, const unsigned int swizzle[4], int res,
                     unsigned char *out) {
  GLuint tex, dst, fbo;
  glGenTextures(1, &tex);
  glBindTexture(GL_TEXTURE_2D, tex);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);          /* texture_packer.js:91-94 */
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
  glTexImage2D(GL_TEXTURE_2D, 0, corrected ? GL_SRGB8_ALPHA8 : GL_RGBA, w, h, 0, GL_RGBA, GL_UNSIGNED_BYTE, src);
  glGenTextures(1, &dst);
  glBindTexture(GL_TEXTURE_2D, dst);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, res, res, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
  glGenFramebuffers(1, &fbo);
  glBindFramebuffer(GL_FRAMEBUFFER, fbo);
  glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, dst, 0);
  if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("writer FBO incomplete");
  glUseProgram(p_writer);
  glClearColor(0.0f, 0.0f, 0.0f, 1.0f);
  glClear(GL_COLOR_BUFFER_BIT);
  glViewport(0, 0, res, res);
  bind_corner(p_writer);
  GLint loc;
  if ((loc = glGetUniformLocation(p_writer, "tex")) >= 0) glUniform1i(loc, 0);
  float dims[2] = {(float)res, (float)res};
  if ((loc = glGetUniformLocation(p_writer, "dims")) >= 0) glUniform2fv(loc, 1, dims);
  if ((loc = glGetUniformLocation(p_writer, "swizzle")) >= 0) glUniform4uiv(loc, 1, swizzle);
  glActiveTexture(GL_TEXTURE0);
  glBindTexture(GL_TEXTURE_2D, tex);
  glDrawArrays(GL_TRIANGLES, 0, 3);
  glFinish();
  glReadPixels(0, 0, res, res, GL_RGBA, GL_UNSIGNED_BYTE, out);
  GLCHK("write_texture");
  return 0;
}
