/* libfspt_mock.c - a stand-in for libfspt.so that needs no GPU, for tests/test_napi_handles.py only.
 * The N-API addon (fspt_amd/csrc/fspt_napi.c) is built against it so that the addon's OWN logic - handle kinds,
 * finalizers, the renderAsync busy flag - can be tested here on the CPU: every object is a malloc'ed counter block,
 * fspt_render sleeps (so a job is observably "in flight"), and live-object counts are exported through
 * fspt_device_count() (scenes + targets + multis + builders currently alive).  Nothing renders. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "fspt.h"
#include "fspt_tuning.h"

static int g_live = 0;
static int g_render_ms = 300;
struct fspt_scene { int tag; };
struct fspt_target { int tag; fspt_scene *scene; uint32_t W, H; int renders; };
struct fspt_multi { int tag; uint32_t W, H; fspt_target *t0; };
struct fspt_builder { int tag; };

const char *fspt_last_error(void) { return "mock"; }
int fspt_abi_version(void) { return FSPT_ABI_VERSION; }
int fspt_device_count(void) { return g_live; } /* (the mock's hook: objects alive) */

int fspt_scene_create(const fspt_scene_desc *d, int device, fspt_scene **out) { (void)d; (void)device; *out = calloc(1, sizeof **out); (*out)->tag = 1; g_live++; return 0; }
int fspt_scene_destroy(fspt_scene *s) { if (s) { if (s->tag != 1) abort(); s->tag = 0; free(s); g_live--; } return 0; }
int fspt_target_create(fspt_scene *s, uint32_t W, uint32_t H, fspt_target **out) {
  if (!s || s->tag != 1) abort();
  *out = calloc(1, sizeof **out); (*out)->tag = 2; (*out)->scene = s; (*out)->W = W; (*out)->H = H; g_live++; return 0;
}
int fspt_target_destroy(fspt_target *t) { if (t) { if (t->tag != 2 || t->scene->tag != 1) abort(); /* the scene must outlive it */ t->tag = 0; free(t); g_live--; } return 0; }
int fspt_target_size(fspt_target *t, uint32_t *W, uint32_t *H) { if (t->tag != 2) abort(); *W = t->W; *H = t->H; return 0; }
int fspt_render(fspt_target *t, const fspt_camera_params *c, uint32_t first, uint32_t n, uint64_t seed) {
  (void)c; (void)first; (void)n; (void)seed;
  if (t->tag != 2) abort();
  usleep(g_render_ms * 1000);
  if (t->tag != 2) abort(); /* destroyed while "rendering": the race the busy flag exists to prevent */
  t->renders++;
  return 0;
}
int fspt_sync(fspt_target *t) { if (t->tag != 2) abort(); return 0; }
int fspt_read_radiance(fspt_target *t, float *out) { if (t->tag != 2) abort(); out[0] = (float)t->renders; return 0; }
int fspt_clear(fspt_target *t) { if (t->tag != 2) abort(); t->renders = 0; return 0; }
int fspt_camera(fspt_target *t, const float P[3], const float I[3], float f, const float l[2], float rb) { (void)P; (void)I; (void)f; (void)l; (void)rb; if (t->tag != 2) abort(); return 0; }
int fspt_trace(fspt_target *t, uint32_t tick, float rb, float th, uint32_t nb) { (void)tick; (void)rb; (void)th; (void)nb; if (t->tag != 2) abort(); return 0; }
int fspt_trace_test(fspt_target *t, uint32_t tick) { (void)tick; if (t->tag != 2) abort(); return 0; }
int fspt_draw_scaled(fspt_target *t, float a, float b, int c, float d, float e, uint8_t *o) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)o; if (t->tag != 2) abort(); return 0; }

int fspt_multi_create(const fspt_scene_desc *d, const int *dev, uint32_t n, uint32_t W, uint32_t H, fspt_multi **out) {
  (void)d; (void)dev; (void)n;
  *out = calloc(1, sizeof **out); (*out)->tag = 3; (*out)->W = W; (*out)->H = H; g_live++;
  fspt_scene *s; fspt_scene_create(NULL, 0, &s); fspt_target_create(s, W, H, &(*out)->t0);
  return 0;
}
int fspt_multi_destroy(fspt_multi *m) { if (m) { if (m->tag != 3) abort(); fspt_scene *s = m->t0->scene; fspt_target_destroy(m->t0); fspt_scene_destroy(s); m->tag = 0; free(m); g_live--; } return 0; }
int fspt_multi_target(fspt_multi *m, uint32_t i, fspt_target **out) { (void)i; if (m->tag != 3) abort(); *out = m->t0; return 0; }
int fspt_multi_size(fspt_multi *m, uint32_t *W, uint32_t *H) { if (m->tag != 3) abort(); *W = m->W; *H = m->H; return 0; }
int fspt_multi_render(fspt_multi *m, const fspt_camera_params *c, uint32_t f, uint32_t n, uint64_t s) { if (m->tag != 3) abort(); return fspt_render(m->t0, c, f, n, s); }
int fspt_multi_sync(fspt_multi *m) { if (m->tag != 3) abort(); return 0; }
int fspt_multi_clear(fspt_multi *m) { if (m->tag != 3) abort(); return 0; }
int fspt_multi_camera(fspt_multi *m, const float P[3], const float I[3], float f, const float l[2], float rb) { (void)P; (void)I; (void)f; (void)l; (void)rb; if (m->tag != 3) abort(); return 0; }
int fspt_multi_trace(fspt_multi *m, uint32_t t, float rb, float th, uint32_t nb) { (void)t; (void)rb; (void)th; (void)nb; if (m->tag != 3) abort(); return 0; }
int fspt_multi_read_radiance(fspt_multi *m, float *out) { if (m->tag != 3) abort(); return fspt_read_radiance(m->t0, out); }
int fspt_multi_draw(fspt_multi *m, float a, float b, int c, float d, uint8_t *o) { (void)a; (void)b; (void)c; (void)d; (void)o; if (m->tag != 3) abort(); return 0; }
int fspt_multi_set_exchange(fspt_multi *m, int mode) { (void)mode; if (m->tag != 3) abort(); return 0; }
int fspt_multi_get_exchange(fspt_multi *m, int *mode, int *v) { if (m->tag != 3) abort(); if (mode) *mode = 0; if (v) *v = 0; return 0; }
int fspt_multi_last_stage_ms(fspt_multi *m, float *ms, uint32_t n) { if (m->tag != 3) abort(); for (uint32_t i = 0; i < 4 * n; ++i) ms[i] = -1.0f; return 0; }

int fspt_builder_create(fspt_builder **out) { *out = calloc(1, sizeof **out); (*out)->tag = 4; g_live++; return 0; }
int fspt_builder_destroy(fspt_builder *b) { if (b) { if (b->tag != 4) abort(); b->tag = 0; free(b); g_live--; } return 0; }
int fspt_builder_normalize(fspt_builder *b, double s) { (void)s; if (b->tag != 4) abort(); return 0; }

float fspt_rand_base_next(uint64_t *s) { *s += 1; return 0.5f; }
