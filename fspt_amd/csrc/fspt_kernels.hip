// fspt_kernels.hip — gfx950 kernels of libfspt.
//
// The reference's hot path is one fragment-shader invocation per pixel
// (shader/tracer.fs:436-518).  Two execution strategies run the same per-path
// arithmetic (advance_path / shade_hit / trace loops below) and give
// bit-identical results:
//
//  * wavefront pipeline (default; k_wf_primary / k_wf_trace / k_wf_logic [/ k_wf_tail] / k_wf_resolve):
//    path state in HBM, compacted every round into consecutive indices of one of two state sets, one kernel per
//    kind of work; the first launch of a batch does ray generation, the primary ray's traversal and its shading
//    in one kernel.  k_wf_trace is a persistent while-while traversal with per-lane dynamic refill from the
//    round's path list (one item per path: its shadow ray, then its extension ray), its deferred-child stack in
//    LDS; k_wf_logic classifies the live paths, reserves the survivors' output range with one atomic per 4 096
//    paths, finishes the others in place and shades the survivors with dense waves.  Two schedulers drive these
//    kernels (fspt_api.cpp): BATCHES - every (pixel, tick) of up to 128 ticks at once, rounds that drain to nothing - and
//    a STREAM over a fixed pool of live paths that k_wf_plan keeps full (fspt_device.hpp: WfStreamCtl).  A trace wave
//    that can get no more rays parks its unfinished traversals in records; k_wf_carry moves those paths on and the next
//    trace launch resumes them first (fspt_device.hpp: suspended traversals).
//  * megakernel (k_trace): one persistent kernel per tick; a lane whose path has
//    terminated pulls the next pixel in place (path regeneration); the wave alternates
//      S  consume traversal results (tracer.fs:501-512), finish / regenerate
//         (camera.fs:37-46, tracer.fs:515-517) or shade (tracer.fs:447-499),
//      T  trace the lane's one or two pending rays (shadow, then extension).
//
// Arithmetic is "fspt-math" (fspt_math.hpp); traversal order, pruning rule and the
// 4-triangle leaf over-read are the reference's, so results equal the CPU restatement
// (oracle/) bit for bit.
#include "fspt_device.hpp"
#include "fspt_math.hpp"

namespace fspt {
using namespace fm;

#define WAVE 64
#ifndef FSPT_TAP2
#define FSPT_TAP2 1
#endif
#ifndef WF_LOGIC_LDSTAB
#define WF_LOGIC_LDSTAB 1
#endif
#ifndef WF_TRACE_LDS_TOP
#define WF_TRACE_LDS_TOP 31 // top-of-tree nodes (breadth-first) k_wf_trace keeps in LDS: measured best of {0, 31, 77} (profiles/r01)
#endif
#define WF_TRACE_BLOCKS_PER_CU 7u // most 256-thread blocks per CU the LDS split is computed for (the registers - 80 since the suspended traversals - allow 6)
#define BLOCK_THREADS 256
#define WAVES_PER_BLOCK (BLOCK_THREADS / WAVE)
#define WORK_CHUNK 256u

// ---------------------------------------------------------------------------
// rayBoxIntersect (tracer.fs:317-326) for BOTH child boxes of a node; 1/dir hoisted (same value every call).
// Packed FP32: gfx950 issues v_pk_add_f32 / v_pk_mul_f32 (two IEEE binary32 operations per lane) at the rate of one
// scalar operation, and the traversal kernels are bound by VALU issue (profiles/r02: 0.85-0.98 of the issue rate), so
// the node stores the slab bounds in the pairs the arithmetic wants (fspt_device.hpp): the 12 subtractions and 12
// multiplications of a step are 6 + 6 instructions.  Same operations on the same operands as the scalar form:
// bit-identical results.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4)));
FM_DEV f2 mk2(float a, float b) { return (f2){a, b}; }
FM_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
FM_DEV float slab(f2 txy1, f2 txy2, f2 tz) {
  float tMax = min_(min_(max_(txy1.x, txy2.x), max_(txy1.y, txy2.y)), max_(tz.x, tz.y));
  float tMin = max_(max_(min_(txy1.x, txy2.x), min_(txy1.y, txy2.y)), min_(tz.x, tz.y));
  return (tMax >= tMin && tMax > 0.0f) ? tMin : MAX_T;
}
FM_DEV void node_test(float4 n0, float4 n1, float4 n2, V3 o, V3 inv, float &tl, float &tr) {
  const f2 oxy = mk2(o.x, o.y), ixy = mk2(inv.x, inv.y), oz = mk2(o.z, o.z), iz = mk2(inv.z, inv.z);
  const f2 l1 = (mk2(n0.x, n0.y) - oxy) * ixy, l2 = (mk2(n0.z, n0.w) - oxy) * ixy; // left: (t1x, t1y), (t2x, t2y)
  const f2 r1 = (mk2(n1.x, n1.y) - oxy) * ixy, r2 = (mk2(n1.z, n1.w) - oxy) * ixy; // right
  const f2 lz = (mk2(n2.x, n2.y) - oz) * iz, rz = (mk2(n2.z, n2.w) - oz) * iz;     // (t1z, t2z) left, right
  tl = slab(l1, l2, lz);
  tr = slab(r1, r2, rz);
}
// The two child references of a node (f4[3].xy): ONE 8-byte load.  (Written as an int2 / int4 access the compiler
// widens it to 16 bytes.)
FM_DEV int2 node_refs(const float4 *n) {
  const long long rr = *reinterpret_cast<const long long *>(n + 3);
  return make_int2((int)(rr & 0xffffffffll), (int)(rr >> 32));
}

// One step of intersectScene at an interior node (tracer.fs:382-401) once the two entry distances are known: descend
// into the nearer child that is hit (the other one, if hit too, goes on the stack), or pop.  Returns which child the
// walk descends into: 0 left, 1 right, -1 neither (cur then holds the popped reference).
FM_DEV int node_decide(float tl, float tr, int refL, int refR, float t, int *stack, int &sp, int &cur) {
  const bool hl = tl < t, hr = tr < t;
  const bool swap = tl > tr; // tracer.fs:384: right first only when strictly nearer
  if (hl && hr) {
    stack[sp * WAVE] = swap ? refL : refR;
    sp++;
    cur = swap ? refR : refL;
    return swap ? 1 : 0;
  }
  if (hl) { cur = refL; return 0; }
  if (hr) { cur = refR; return 1; }
  if (sp > 0) { sp--; cur = stack[sp * WAVE]; }
  else cur = REF_SENTINEL;
  return -1;
}
// TWO steps on one two-level node (fspt_device.hpp "quad"; the eight loads are the caller's: global or LDS): the step at
// the node itself - its children's boxes are the unions of the stored grandchildren's, derived exactly - and, when the
// walk descends into a child that is an interior node, that child's step right away: result.t has not changed in
// between (no leaf was visited), so it is the test the one-level walk makes after its next fetch.  `steps` counts the
// reference's loop iterations exactly as the one-level walk does.
template <bool COUNT>
FM_DEV void quad_step(float4 a0, float4 a1, float4 a2, int4 ar, float4 b0, float4 b1, float4 b2, int2 br, V3 o, V3 inv, float t,
                      int *stack, int &sp, int &cur, uint32_t &steps) {
  const float4 n0 = make_float4(min_(a0.x, a1.x), min_(a0.y, a1.y), max_(a0.z, a1.z), max_(a0.w, a1.w)); // box(L) = box(LL) U box(LR)
  const float4 n1 = make_float4(min_(b0.x, b1.x), min_(b0.y, b1.y), max_(b0.z, b1.z), max_(b0.w, b1.w)); // box(R)
  const float4 n2 = make_float4(min_(a2.x, a2.z), max_(a2.y, a2.w), min_(b2.x, b2.z), max_(b2.y, b2.w));
  float tl, tr;
  node_test(n0, n1, n2, o, inv, tl, tr);
  const int side = node_decide(tl, tr, ar.z, ar.w, t, stack, sp, cur);
  if (side < 0 || cur < 0) return; // popped, or the child is a leaf (its record is the next fetch)
  if (COUNT) steps++;
  const bool right = side != 0;
  const float4 c0 = right ? b0 : a0, c1 = right ? b1 : a1, c2 = right ? b2 : a2;
  node_test(c0, c1, c2, o, inv, tl, tr);
  node_decide(tl, tr, right ? br.x : ar.x, right ? br.y : ar.y, t, stack, sp, cur);
}
FM_DEV void quad_load(const float4 *q, float4 &a0, float4 &a1, float4 &a2, int4 &ar, float4 &b0, float4 &b1, float4 &b2, int2 &br) {
  a0 = q[0]; a1 = q[1]; a2 = q[2];
  const float4 r = q[3];
  ar = make_int4(__float_as_int(r.x), __float_as_int(r.y), __float_as_int(r.z), __float_as_int(r.w));
  b0 = q[4]; b1 = q[5]; b2 = q[6];
  br = node_refs(q + 4);
}

// rayTriangleIntersect (tracer.fs:300-315) on a pre-edged triangle; the early
// returns become one predicate with the same NaN behaviour.
FM_DEV float ray_tri(V3 o, V3 d, V3 v1, V3 e1, V3 e2) {
  V3 p = cross(d, e2);
  float det = dot(e1, p);
  float invDet = 1.0f / det;
  V3 t = o - v1;
  float u = dot(t, p) * invDet;
  V3 q = cross(t, e1);
  float v = dot(d, q) * invDet;
  float dist = dot(e2, q) * invDet;
  bool miss = (abs_(det) < EPSILON) || (u < 0.0f) || (u > 1.0f) || (v < 0.0f) || (u + v > 1.0f) || !(dist > EPSILON);
  return miss ? MAX_T : dist;
}

// Two triangles at once (packed FP32, see node_test): component c of the pair is (tri_a.c, tri_b.c).  Every
// operation is ray_tri's, on the same operands, in the same order.
FM_DEV f2 ray_tri2(V3 o, V3 d, f2 v1x, f2 v1y, f2 v1z, f2 e1x, f2 e1y, f2 e1z, f2 e2x, f2 e2y, f2 e2z) {
  const f2 dx = mk2(d.x, d.x), dy = mk2(d.y, d.y), dz = mk2(d.z, d.z);
  const f2 px = fma2(dy, e2z, -(dz * e2y)), py = fma2(dz, e2x, -(dx * e2z)), pz = fma2(dx, e2y, -(dy * e2x)); // cross(d, e2)
  const f2 det = fma2(e1z, pz, fma2(e1y, py, e1x * px));
  const f2 invDet = mk2(1.0f / det.x, 1.0f / det.y);
  const f2 tx = mk2(o.x, o.x) - v1x, ty = mk2(o.y, o.y) - v1y, tz = mk2(o.z, o.z) - v1z;
  const f2 u = fma2(tz, pz, fma2(ty, py, tx * px)) * invDet;
  const f2 qx = fma2(ty, e1z, -(tz * e1y)), qy = fma2(tz, e1x, -(tx * e1z)), qz = fma2(tx, e1y, -(ty * e1x)); // cross(t, e1)
  const f2 v = fma2(dz, qz, fma2(dy, qy, dx * qx)) * invDet;
  const f2 dist = fma2(e2z, qz, fma2(e2y, qy, e2x * qx)) * invDet;
  const f2 uv = u + v;
  const bool m0 = (abs_(det.x) < EPSILON) || (u.x < 0.0f) || (u.x > 1.0f) || (v.x < 0.0f) || (uv.x > 1.0f) || !(dist.x > EPSILON);
  const bool m1 = (abs_(det.y) < EPSILON) || (u.y < 0.0f) || (u.y > 1.0f) || (v.y < 0.0f) || (uv.y > 1.0f) || !(dist.y > EPSILON);
  return mk2(m0 ? MAX_T : dist.x, m1 ? MAX_T : dist.y);
}

// ---------------------------------------------------------------------------
// processLeaf (tracer.fs:355-364): always LEAF_SIZE triangles from the leaf's first one, strict `<` update in
// triangle order.  Leaf record `leaf` (fspt_device.hpp): the LEAF_SIZE pre-edged triangles the reference would read,
// component-major - for LEAF_SIZE = 4 nine 16-byte loads whose halves are the operand pairs of ray_tri2.
// `hit` becomes the leaf SLOT (leaf * LEAF_SIZE + i): the index the hit records are stored under; slot_to_tri gives
// the reference's triangle index where one is reported (fspt_intersect).
// ---------------------------------------------------------------------------
FM_DEV void process_leaf(const float *__restrict__ leaves, uint32_t leaf_size, int leaf, V3 o, V3 d, float &t, int &hit) {
  if (leaf_size == 4) {
    const f4u *q = reinterpret_cast<const f4u *>(leaves) + (size_t)leaf * TRI_FLOATS;
    const f4u c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3], c4 = q[4], c5 = q[5], c6 = q[6], c7 = q[7], c8 = q[8];
    const f2 a = ray_tri2(o, d, c0.xy, c1.xy, c2.xy, c3.xy, c4.xy, c5.xy, c6.xy, c7.xy, c8.xy);
    const f2 b = ray_tri2(o, d, c0.zw, c1.zw, c2.zw, c3.zw, c4.zw, c5.zw, c6.zw, c7.zw, c8.zw);
    const int s0 = leaf * 4;
    if (a.x < t) { t = a.x; hit = s0; }
    if (a.y < t) { t = a.y; hit = s0 + 1; }
    if (b.x < t) { t = b.x; hit = s0 + 2; }
    if (b.y < t) { t = b.y; hit = s0 + 3; }
  } else {
    const float *a = leaves + (size_t)leaf * leaf_size * TRI_FLOATS;
    for (uint32_t i = 0; i < leaf_size; ++i) {
      float res = ray_tri(o, d, v3(a[i], a[leaf_size + i], a[2 * leaf_size + i]),
                          v3(a[3 * leaf_size + i], a[4 * leaf_size + i], a[5 * leaf_size + i]),
                          v3(a[6 * leaf_size + i], a[7 * leaf_size + i], a[8 * leaf_size + i]));
      if (res < t) { t = res; hit = leaf * (int)leaf_size + (int)i; }
    }
  }
}
// leaf slot -> the reference's triangle index (first triangle of the leaf + i, tracer.fs:360); -1 stays -1
FM_DEV int slot_to_tri(const DScene &S, int slot) { return slot < 0 ? -1 : (int)S.slot_tri[slot]; }

struct Counters {
  uint32_t samples, rays, steps, leaves, shades, envs;
};

// ---------------------------------------------------------------------------
// intersectScene (tracer.fs:366-404) for up to two rays sharing an origin:
// ray A (optional: the NEE shadow ray, tracer.fs:501) then ray B (primary or
// extension ray, tracer.fs:440,507).  `stack` points at this lane's column of
// the wave's LDS stack (entry k at stack[k*64]).
// ---------------------------------------------------------------------------
// ANYHIT: ray A stops at its first hit (only `shadow.index == -1` is consumed, tracer.fs:502).
// WIDE: walk the two-level nodes (S.quads, must be non-NULL): two steps per memory round trip, same walk.
template <bool COUNT, bool ANYHIT, bool WIDE = false>
FM_DEV void trace_rays(const DScene &S, int *stack, V3 o, bool hasA, V3 dA, V3 dB, int &hitA, float &tB, int &hitB,
                       Counters &cnt) {
  int slot = hasA ? 0 : 1;
  V3 d = hasA ? dA : dB;
  V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  float t = MAX_T;
  int hit = -1;
  int cur = S.root_ref;
  int sp = 0;
  hitA = -1;
  if (COUNT) cnt.rays++;
  const float4 *__restrict__ nodes = WIDE ? S.quads : S.nodes;
  const float *__restrict__ leaves = S.leaves;
  const uint32_t leaf_size = S.leaf_size;
  while (true) {
    // ---- interior nodes ----------------------------------------------------
    while (cur >= 0) {
      if (COUNT) cnt.steps++;
      if constexpr (WIDE) {
        float4 a0, a1, a2, b0, b1, b2; int4 ar; int2 br;
        quad_load(nodes + (size_t)cur * QUAD_F4, a0, a1, a2, ar, b0, b1, b2, br);
        quad_step<COUNT>(a0, a1, a2, ar, b0, b1, b2, br, o, inv, t, stack, sp, cur, cnt.steps);
        continue;
      }
      const float4 *n = nodes + (size_t)cur * NODE_F4;
      float4 n0 = n[0], n1 = n[1], n2 = n[2];
      const int2 n3 = node_refs(n);
      float tl, tr;
      node_test(n0, n1, n2, o, inv, tl, tr);
      bool hl = tl < t, hr = tr < t;
      bool swap = tl > tr; // tracer.fs:384: right first only when strictly nearer
      int nearRef = swap ? n3.y : n3.x;
      int farRef = swap ? n3.x : n3.y;
      if (hl && hr) {
        stack[sp * WAVE] = farRef;
        sp++;
        cur = nearRef;
      } else if (hl) {
        cur = n3.x;
      } else if (hr) {
        cur = n3.y;
      } else if (sp > 0) {
        sp--;
        cur = stack[sp * WAVE];
      } else {
        cur = REF_SENTINEL;
      }
    }
    if (cur == REF_SENTINEL) {
      if (slot == 0) { // shadow ray done: only its hit/miss is consumed (tracer.fs:502)
        hitA = hit;
        slot = 1;
        d = dB;
        inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        t = MAX_T;
        hit = -1;
        cur = S.root_ref;
        if (COUNT) cnt.rays++;
        continue;
      }
      break;
    }
    // ---- leaf: processLeaf (tracer.fs:355-364), always leaf_size triangles ----
    {
      if (COUNT) { cnt.steps++; cnt.leaves++; }
      int ts = ~cur;
      process_leaf(leaves, leaf_size, ts, o, d, t, hit);
      if (sp > 0) { sp--; cur = stack[sp * WAVE]; }
      else cur = REF_SENTINEL;
      if (ANYHIT && slot == 0 && hit != -1) cur = REF_SENTINEL;
    }
  }
  tB = t;
  hitB = hit;
}

// intersectScene (tracer.fs:366-404) for ONE ray per lane, in SLICES: the traversal state (node reference, stack depth,
// t, hit; the stack itself is the lane's LDS column) is the caller's and survives the call.  A lane walks on from
// where it stands until its ray is finished (cur == REF_SENTINEL) or it has done `budget` loop iterations in this call;
// the call returns when no lane has anything left to do within its budget.  `anyhit`: stop at the first hit (NEE
// shadow rays, tracer.fs:502).  Same node sequence and arithmetic as trace_rays, whatever the slicing.
// WIDE: 0 the 64-byte nodes, 1 the two-level nodes, 2 the caller says per call (`wide_now`, wave-uniform): the traversal
// state - node reference, stack entries - means the same in both walks (the two arrays are indexed alike), so a ray can
// change form between any two steps.
template <bool COUNT, int WIDE = 0>
FM_DEV void trace_slice(const DScene &S, int *stack, V3 o, V3 d, V3 inv /* 1 / d */, bool anyhit, int &cur, int &sp, float &t, int &hit,
                        uint32_t budget, uint32_t &n /* loop iterations (memory round trips) this lane has used of the budget */, Counters &cnt,
                        bool wide_now = false) {
  const float4 *__restrict__ nodes = S.nodes;
  const float4 *__restrict__ quads = S.quads;
  const float *__restrict__ leaves = S.leaves;
  const uint32_t leaf_size = S.leaf_size;
  while (cur != REF_SENTINEL && n < budget) {
    while (cur >= 0 && n < budget) {
      if (COUNT) cnt.steps++;
      ++n;
      if (WIDE == 1 || (WIDE == 2 && wide_now)) {
        float4 a0, a1, a2, b0, b1, b2; int4 ar; int2 br;
        quad_load(quads + (size_t)cur * QUAD_F4, a0, a1, a2, ar, b0, b1, b2, br);
        quad_step<COUNT>(a0, a1, a2, ar, b0, b1, b2, br, o, inv, t, stack, sp, cur, cnt.steps);
        continue;
      }
      const float4 *nd = nodes + (size_t)cur * NODE_F4;
      float4 n0 = nd[0], n1 = nd[1], n2 = nd[2];
      const int2 n3 = node_refs(nd);
      float tl, tr;
      node_test(n0, n1, n2, o, inv, tl, tr);
      bool hl = tl < t, hr = tr < t;
      bool swap = tl > tr; // tracer.fs:384: right first only when strictly nearer
      int nearRef = swap ? n3.y : n3.x;
      int farRef = swap ? n3.x : n3.y;
      if (hl && hr) {
        stack[sp * WAVE] = farRef;
        sp++;
        cur = nearRef;
      } else if (hl) {
        cur = n3.x;
      } else if (hr) {
        cur = n3.y;
      } else if (sp > 0) {
        sp--;
        cur = stack[sp * WAVE];
      } else {
        cur = REF_SENTINEL;
      }
    }
    if (cur >= 0 || cur == REF_SENTINEL) break; // out of budget on an interior node / finished
    if (COUNT) { cnt.steps++; cnt.leaves++; }
    ++n;
    process_leaf(leaves, leaf_size, ~cur, o, d, t, hit);
    if (sp > 0) { sp--; cur = stack[sp * WAVE]; }
    else cur = REF_SENTINEL;
    if (anyhit && hit != -1) cur = REF_SENTINEL;
  }
}

// ---------------------------------------------------------------------------
// Texture fetches (sampler state: main.js:170-180, 548-559)
// ---------------------------------------------------------------------------
// (float)b / 255.0f, exactly (verified for all 256 bytes: one Newton step on
// the rounded reciprocal is correctly rounded for these operands)
FM_DEV float unorm8(uint32_t b) {
  const float rc = 1.0f / 255.0f;
  float x = (float)b;
  float q = x * rc;
  float r = fma_(-q, 255.0f, x);
  return fma_(r, rc, q);
}
FM_DEV int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
FM_DEV int wrap_clamp(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
FM_DEV float safe_floor_coord(float u) {
  float f = floor_(u);
  if (!(f > -1.0e9f && f < 1.0e9f)) f = 0.0f;
  return f;
}
struct Tap4 { uint32_t t00, t10, t01, t11; float a, b; };
// Texel (i, j) of a w-wide image stored in tiles of 2^WL2 x 2^HL2 texels (fspt_device.hpp): one tile = one 128-byte
// cache line, so the 2 x 2 footprint of a bilinear fetch lies in 1.4 lines on average (8 x 4 tiles) instead of 2 rows = 2 lines.
template <int WL2, int HL2>
FM_DEV uint32_t tile_offset(int i, int j, int tiles_x) {
  return (uint32_t)(((j >> HL2) * tiles_x + (i >> WL2)) << (WL2 + HL2)) + (uint32_t)(((j & ((1 << HL2) - 1)) << WL2) + (i & ((1 << WL2) - 1)));
}
// Footprint of a bilinear fetch at (s, t): the wrapped texel coordinates and the two weights.  Shared by the four atlas
// layers a shading event reads at the same uv (tracer.fs:453-456): the wrap / floor arithmetic is done once.
struct TexCoord { int i0, i1, j0, j1; float a, b; };
FM_DEV TexCoord bilinear_coord(int w, int h, float s, float t, bool repeat_t) {
  TexCoord c;
  float u = fma_(s, (float)w, -0.5f), v = fma_(t, (float)h, -0.5f);
  float fu = safe_floor_coord(u), fv = safe_floor_coord(v);
  float a = u - fu, b = v - fv;
  if (!(a >= 0.0f && a <= 1.0f)) a = 0.0f;
  if (!(b >= 0.0f && b <= 1.0f)) b = 0.0f;
  int i0 = (int)fu, j0 = (int)fv;
  c.i1 = wrap_repeat(i0 + 1, w);
  c.i0 = wrap_repeat(i0, w);
  if (repeat_t) { c.j1 = wrap_repeat(j0 + 1, h); c.j0 = wrap_repeat(j0, h); }
  else { c.j1 = wrap_clamp(j0 + 1, h); c.j0 = wrap_clamp(j0, h); }
  c.a = a; c.b = b;
  return c;
}
// ... and its four texel offsets in a single-layer image (8 x 4-texel tiles)
struct TapGeom { uint32_t o00, o10, o01, o11; float a, b; bool pair; };
FM_DEV TapGeom tap_geom(const TexCoord &c, int w) {
  TapGeom g;
  const int tiles_x = (w + TEX_TILE_W - 1) >> TEX_TILE_W_LOG2;
  g.o00 = tile_offset<TEX_TILE_W_LOG2, TEX_TILE_H_LOG2>(c.i0, c.j0, tiles_x);
  g.o01 = tile_offset<TEX_TILE_W_LOG2, TEX_TILE_H_LOG2>(c.i0, c.j1, tiles_x);
  g.o10 = tile_offset<TEX_TILE_W_LOG2, TEX_TILE_H_LOG2>(c.i1, c.j0, tiles_x);
  g.o11 = tile_offset<TEX_TILE_W_LOG2, TEX_TILE_H_LOG2>(c.i1, c.j1, tiles_x);
  // the two taps of a row are neighbours unless the column wraps or leaves the tile: one 8-byte load (dword-aligned)
  // instead of two 4-byte loads - half the lane-requests on the vector-memory pipeline, same texel values
  g.pair = FSPT_TAP2 && c.i1 == c.i0 + 1 && (TEX_TILE_W == 1 || (c.i0 & (TEX_TILE_W - 1)) != TEX_TILE_W - 1);
  g.a = c.a; g.b = c.b;
  return g;
}
FM_DEV Tap4 fetch_taps(const uint32_t *texels, const TapGeom &g) {
  Tap4 r;
  typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(4)));
  if (g.pair) {
    u2a q0 = *reinterpret_cast<const u2a *>(texels + g.o00), q1 = *reinterpret_cast<const u2a *>(texels + g.o01);
    r.t00 = q0.x; r.t10 = q0.y; r.t01 = q1.x; r.t11 = q1.y;
  } else {
    r.t00 = texels[g.o00]; r.t10 = texels[g.o10];
    r.t01 = texels[g.o01]; r.t11 = texels[g.o11];
  }
  r.a = g.a; r.b = g.b;
  return r;
}
FM_DEV Tap4 bilinear_taps(const uint32_t *texels, int w, int h, float s, float t, bool repeat_t) {
  return fetch_taps(texels, tap_geom(bilinear_coord(w, h, s, t, repeat_t), w));
}
// The environment map (REPEAT in s, CLAMP in t) in OVERLAPPING 8 x 4-texel tiles: tile (a, b) holds texels
// [7a, 7a + 8) x [3b, 3b + 4), columns wrapped and rows clamped when it was built, so the 2 x 2 footprint of EVERY lookup
// lies in one 128-byte line (1.4 lines with disjoint tiles; the lookups are random, each line it touches is a miss more
// often than not) and its two rows are always two 8-byte loads.  Same texels, same weights: bit-identical.
FM_DEV Tap4 env_taps(const uint32_t *texels, int w, int h, float s, float t) {
  float u = fma_(s, (float)w, -0.5f), v = fma_(t, (float)h, -0.5f);
  float fu = safe_floor_coord(u), fv = safe_floor_coord(v);
  float a = u - fu, b = v - fv;
  if (!(a >= 0.0f && a <= 1.0f)) a = 0.0f;
  if (!(b >= 0.0f && b <= 1.0f)) b = 0.0f;
  const int i0 = wrap_repeat((int)fu, w);
  int j0 = (int)fv;
  // rows (clamp(j0), clamp(j0 + 1)): below row 0 both are row 0 - read rows (0, 1) with weight 0 (lerp(x, y, 0) = x);
  // beyond the last row both are the last row, which is what the clamped overlap row of the last tile holds
  if (j0 < 0) { j0 = 0; b = 0.0f; }
  if (j0 > h - 1) j0 = h - 1;
  const uint32_t ta = (uint32_t)i0 / 7u, tb = (uint32_t)j0 / 3u;
  const uint32_t la = (uint32_t)i0 - ta * 7u, lb = (uint32_t)j0 - tb * 3u;
  const uint32_t tiles_x = ((uint32_t)w + 6u) / 7u;
  const uint32_t off = ((tb * tiles_x + ta) << 5) + (lb << 3) + la;
  typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(4)));
  const u2a q0 = *reinterpret_cast<const u2a *>(texels + off), q1 = *reinterpret_cast<const u2a *>(texels + off + 8u);
  return Tap4{q0.x, q0.y, q1.x, q1.y, a, b};
}
FM_DEV float tap_channel(const Tap4 &tp, int ch) {
  float t00 = unorm8((tp.t00 >> (8 * ch)) & 255u), t10 = unorm8((tp.t10 >> (8 * ch)) & 255u);
  float t01 = unorm8((tp.t01 >> (8 * ch)) & 255u), t11 = unorm8((tp.t11 >> (8 * ch)) & 255u);
  return lerp_(lerp_(t00, t10, tp.a), lerp_(t01, t11, tp.a), tp.b);
}
// One atlas layer of a material texture set in SEPARATE form: its tiled image, or - a layer whose texels are all equal
// (TexturePacker fills a whole res x res layer for every flat colour once one image is in the atlas,
// texture_packer.js:36-42) is not stored - its one texel: lerp(x, x, a) = fma(a, 0, x) = x exactly, so four equal taps
// give the same bits as the fetch would.
FM_DEV Tap4 layer_taps(const DScene &S, const TapGeom &g, uint32_t base_tiles, uint32_t texel) {
  if (base_tiles == LAYER_CONST) {
    Tap4 r;
    r.t00 = r.t10 = r.t01 = r.t11 = texel;
    r.a = g.a; r.b = g.b;
    return r;
  }
  return fetch_taps(S.atlas + (size_t)base_tiles * (TEX_TILE_W * TEX_TILE_H), g);
}
// envSample + envColor (tracer.fs:410-419)
template <bool COUNT>
FM_DEV V3 env_sample(const DScene &S, V3 dir, float envTheta, Counters &cnt) {
  if (COUNT) cnt.envs++;
  if (!S.env) return v3(0.0f, 0.0f, 0.0f);
  float cx = envTheta + atan2_(dir.z, dir.x) / M_TAU_F;
  float cy = fma_(asin_(-dir.y), INV_PI_F, 0.5f);
#if FSPT_ENV_APRON
  Tap4 tp = env_taps(S.env, (int)S.env_w, (int)S.env_h, cx, cy);
#else
  Tap4 tp = bilinear_taps(S.env, (int)S.env_w, (int)S.env_h, cx, cy, false);
#endif
  float r = tap_channel(tp, 0), g = tap_channel(tp, 1), b = tap_channel(tp, 2), e = tap_channel(tp, 3);
  float sc = exp2_(fma_(e, 255.0f, -128.0f));
  return v3(r * sc, g * sc, b * sc);
}
// sampleEnv (tracer.fs:421-434)
FM_DEV void sample_env(const DScene &S, float envTheta, float &seed, V3 &dir, float &pdf) {
  float nb = (float)S.n_bins;
  int idx = (int)(nb * rnd(seed));
  if (idx > (int)S.n_bins - 1) idx = (int)S.n_bins - 1;
  if (idx < 0) idx = 0;
  uint4 bn = S.bins[idx];
  float bx = (float)bn.x, by = (float)bn.y, bz = (float)bn.z, bw = (float)bn.w;
  float dx = (float)S.env_w, dy = (float)S.env_h;
  if (!S.env) { dx = 1.0f; dy = 2048.0f; }
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float ux = -envTheta + fma_(bz - bx, r1, bx) / dx;
  float uy = 0.0f + fma_(bw - by, r2, by) / dy;
  float theta = ux * M_TAU_F;
  float phi = uy * M_PI_F;
  float sinPhi, cosPhi, sinTheta, cosTheta;
  sincos_(phi, sinPhi, cosPhi);
  sincos_(theta, sinTheta, cosTheta);
  dir = v3(cosTheta * sinPhi, cosPhi, sinTheta * sinPhi);
  float nominal = (dx * dy) / nb;
  pdf = nominal / (((((bz - bx) * (bw - by)) * M_TAU_F) * M_PI_F) * sinPhi);
}

// ---------------------------------------------------------------------------
// BRDF helpers (tracer.fs:194-298)
// ---------------------------------------------------------------------------
FM_DEV float gtr2(float ndh, float a) {
  float a2 = a * a;
  float t = fma_((a2 - 1.0f) * ndh, ndh, 1.0f);
  return a2 / ((M_PI_F * t) * t);
}
FM_DEV float smithG(float ndv, float alphaG) {
  float a = alphaG * alphaG, b = ndv * ndv;
  return 1.0f / (ndv + sqrt_(fma_(-a, b, a + b)));
}
FM_DEV float gtr2_pdf(V3 incident, V3 normal, float rough, V3 bsdfDir) {
  float alpha = max_(0.001f, rough);
  V3 h = normalize(bsdfDir + incident);
  float cosTheta = abs_(dot(h, normal));
  float pdf = gtr2(cosTheta, alpha) * cosTheta;
  return pdf / (4.0f * abs_(dot(bsdfDir, h)));
}
FM_DEV float schlick(V3 incident, V3 normal, float nx, float ny) {
  float r0 = (nx - ny) / (nx + ny);
  r0 *= r0;
  float cosTheta = dot(normal, incident);
  if (nx > ny) {
    float n = nx / ny;
    float sinTheta2 = (n * n) * fma_(-cosTheta, cosTheta, 1.0f);
    if (sinTheta2 > 1.0f) return 1.0f;
    cosTheta = sqrt_(1.0f - sinTheta2);
  }
  float x = 1.0f - cosTheta;
  float q = ((((1.0f - r0) * x) * x) * x) * x;
  return fma_(q, x, r0);
}
FM_DEV void local_frame(V3 n, V3 &tangent, V3 &bitangent) {
  V3 up = (abs_(n.z) < 0.999f) ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
  tangent = normalize(cross(up, n));
  bitangent = cross(n, tangent);
}
FM_DEV V3 frame_combine(V3 t, V3 b, V3 n, V3 h) {
  return v3(fma_(n.x, h.z, fma_(b.x, h.y, t.x * h.x)), fma_(n.y, h.z, fma_(b.y, h.y, t.y * h.x)),
            fma_(n.z, h.z, fma_(b.z, h.y, t.z * h.x)));
}
FM_DEV V3 sample_microfacet(V3 normal, float rough, float &seed) {
  float r1 = rnd(seed), r2 = rnd(seed);
  V3 t, b;
  local_frame(normal, t, b);
  float a = max_(0.001f, rough);
  float phi = r1 * M_TAU_F;
  float cosTheta = sqrt_((1.0f - r2) / fma_(fma_(a, a, -1.0f), r2, 1.0f));
  float sinTheta = clamp_(sqrt_(fma_(-cosTheta, cosTheta, 1.0f)), 0.0f, 1.0f);
  float sinPhi, cosPhi;
  sincos_(phi, sinPhi, cosPhi);
  V3 h = v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
  return frame_combine(t, b, normal, h);
}
FM_DEV V3 sample_lambert(V3 normal, float &seed) {
  float r1 = rnd(seed), r2 = rnd(seed);
  V3 t, b;
  local_frame(normal, t, b);
  float r = sqrt_(r1);
  float phi = M_TAU_F * r2;
  float sp, cp;
  sincos_(phi, sp, cp);
  V3 d;
  d.x = r * cp;
  d.y = r * sp;
  d.z = sqrt_(max_(0.0f, fma_(-d.y, d.y, fma_(-d.x, d.x, 1.0f))));
  return frame_combine(t, b, normal, d);
}
FM_DEV V3 eval_specular(V3 incident, V3 normal, V3 diffuse, float metallic, float rough, V3 bsdfDir) {
  float ndl = dot(normal, bsdfDir);
  float ndv = dot(normal, incident);
  V3 H = normalize(bsdfDir + incident);
  float ndh = dot(normal, H);
  float a = max_(0.001f, rough);
  float Ds = gtr2(ndh, a);
  float om = 1.0f - metallic;
  V3 Fs = v3(fma_(diffuse.x, metallic, om), fma_(diffuse.y, metallic, om), fma_(diffuse.z, metallic, om));
  float roughg = fma_(rough, 0.5f, 0.5f);
  roughg = roughg * roughg;
  float Gs = smithG(ndl, roughg) * smithG(ndv, roughg);
  return v3((Gs * Fs.x) * Ds, (Gs * Fs.y) * Ds, (Gs * Fs.z) * Ds);
}

// ---------------------------------------------------------------------------
// camera.fs main (37-46) for pixel (x, y)
// ---------------------------------------------------------------------------
FM_DEV void camera_ray(uint32_t x, uint32_t y, uint32_t W, uint32_t H, const CameraP &cam, float randBase, V3 &o,
                       V3 &d) {
  float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
  float resx = (float)W, resy = (float)H;
  float uvx = fma_(fx / resx, 2.0f, -1.0f), uvy = fma_(fy / resy, 2.0f, -1.0f);
  float seed = fma_(fx, resy, randBase) + fy;
  V3 Iv = v3(cam.I[0], cam.I[1], cam.I[2]), Pv = v3(cam.P[0], cam.P[1], cam.P[2]);
  V3 basisX = normalize(cross(Iv, v3(0.0f, 1.0f, 0.0f)));
  V3 basisY = normalize(cross(basisX, Iv));
  float fov = cam.fov_scale;
  float icx = uvx * (resx / resy), icy = uvy * 1.0f;
  V3 screen;
  screen.x = (fma_(icy * basisY.x, fov, (icx * basisX.x) * fov) + Iv.x) + Pv.x;
  screen.y = (fma_(icy * basisY.y, fov, (icx * basisX.y) * fov) + Iv.y) + Pv.y;
  screen.z = (fma_(icy * basisY.z, fov, (icx * basisX.z) * fov) + Iv.z) + Pv.z;
  float theta = (rnd(seed) * M_PI_F) * 2.0f;
  float r = sqrt_(rnd(seed)) * 1.414f;
  float st, ct;
  sincos_(theta, st, ct);
  V3 aa;
  aa.x = (r * ((basisX.x * ct) / resx + (basisY.x * st) / resy)) * fov;
  aa.y = (r * ((basisX.y * ct) / resx + (basisY.y * st) / resy)) * fov;
  aa.z = (r * ((basisX.z * ct) / resx + (basisY.z * st) / resy)) * fov;
  float theta2 = (rnd(seed) * M_PI_F) * 2.0f;
  float s2, c2;
  sincos_(theta2, s2, c2);
  float sq = sqrt_(rnd(seed));
  float lx = cam.lens[0], ly = cam.lens[1];
  V3 dof;
  dof.x = (fma_(s2, basisY.x, c2 * basisX.x) * ly) * sq;
  dof.y = (fma_(s2, basisY.y, c2 * basisX.y) * ly) * sq;
  dof.z = (fma_(s2, basisY.z, c2 * basisX.z) * ly) * sq;
  o = Pv + dof;
  V3 tgt = v3(fma_(dof.x, lx, screen.x + aa.x), fma_(dof.y, lx, screen.y + aa.y), fma_(dof.z, lx, screen.z + aa.z));
  d = normalize(tgt - o);
}

// ---------------------------------------------------------------------------
// Per-lane path state
// ---------------------------------------------------------------------------
struct Path {
  V3 ro, rd;      // pending extension / primary ray
  V3 thr, color;  // accumulatedReflectance, color (tracer.fs:441,445)
  V3 envDir;      // pending NEE shadow ray direction
  V3 pend;        // accumulatedReflectance * envThroughput (tracer.fs:503)
  float wx, wy;   // misWeights (tracer.fs:499)
  int bounce;     // tracer.fs:446 `i`
  int iters;
  int pix;        // y*W + x, -1: lane idle
  bool hasShadow;
  bool primary;   // pending ray is the camera ray
  uint32_t lag;   // rounds the path has lagged behind its generation because a traversal of it was suspended (<= WF_LAG_MAX)
};

// tracer.fs:447-499: shade the hit (t, tri) of ray (ro, rd); sets up the next
// shadow + extension rays in `ps`.
template <bool COUNT>
FM_DEV void shade_hit(const DScene &S, Path &ps, float tHit, int ti, float randBase, float envTheta, Counters &cnt) {
  if (COUNT) cnt.shades++;
  const float4 *hp = S.hitrec + (size_t)ti * HITREC_F4; // 192 B = 3 whole cache lines
  const float4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3], h4 = hp[4], h5 = hp[5], h6 = hp[6], h7 = hp[7], h8 = hp[8],
               h9 = hp[9], h10 = hp[10], h11 = hp[11];
  V3 v1 = v3(h0.x, h0.y, h0.z), e1 = v3(h0.w, h1.x, h1.y), e2 = v3(h1.z, h1.w, h2.x);
  V3 n1 = v3(h2.y, h2.z, h2.w), t1 = v3(h3.x, h3.y, h3.z), b1 = v3(h3.w, h4.x, h4.y);
  V3 n2 = v3(h4.z, h4.w, h5.x), t2 = v3(h5.y, h5.z, h5.w), b2 = v3(h6.x, h6.y, h6.z);
  V3 n3 = v3(h6.w, h7.x, h7.y), t3 = v3(h7.z, h7.w, h8.x), b3 = v3(h8.y, h8.z, h8.w);
  float uv0x = h9.x, uv0y = h9.y, uv1x = h9.z, uv1y = h9.w, uv2x = h10.x, uv2y = h10.y;
  float layDiffuse = h10.z; // (bits) the triangle's material texture set
  float ior = h11.z, dielectric = h11.w;

  V3 rd = ps.rd;
  V3 origin = vfma(rd, tHit, ps.ro);
  // barycentricWeights (tracer.fs:339-353); v0 = e1, v1 = e2
  V3 w;
  {
    V3 vv2 = origin - v1;
    float d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2);
    float d20 = dot(vv2, e1), d21 = dot(vv2, e2);
    float invDenom = 1.0f / fma_(d00, d11, -(d01 * d01));
    float bv = fma_(d11, d20, -(d01 * d21)) * invDenom;
    float bw = fma_(d00, d21, -(d01 * d20)) * invDenom;
    w = v3((1.0f - bv) - bw, bv, bw);
  }
  float tcx = fma_(w.z, uv2x, fma_(w.y, uv1x, w.x * uv0x));
  float tcy = fma_(w.z, uv2y, fma_(w.y, uv1y, w.x * uv0y));
  V3 texDiffuse, texEmissive, texNormal;
  float metallic, rough;
  // texture(texArray, vec3(uv, layer)) x 4 (tracer.fs:453-456) through the triangle's material texture set (hit
  // record word 42; fspt_device.hpp TexSet): the four layers' texels of ONE footprint
  const uint4 *tset = S.tex_sets + (size_t)__float_as_uint(layDiffuse) * 3;
  const uint4 ts0 = tset[0], ts1 = tset[1];
  if (ts0.x == TEXSET_CONST) {
    // four flat colours (texture_packer.js:36-42; every material of a colours-only atlas): the four bilinear taps are
    // the same texel and lerp(x, x, a) = fma(a, 0, x) = x exactly, so the filter arithmetic is skipped (bit-identical)
    uint32_t q = ts1.x;
    texDiffuse = v3(unorm8(q & 255u), unorm8((q >> 8) & 255u), unorm8((q >> 16) & 255u));
    q = ts1.y;
    texEmissive = v3(unorm8(q & 255u), unorm8((q >> 8) & 255u), unorm8((q >> 16) & 255u));
    q = ts1.z;
    metallic = unorm8(q & 255u);
    rough = unorm8((q >> 8) & 255u);
    q = ts1.w;
    texNormal = v3((unorm8(q & 255u) - 0.5f) * 2.0f, (unorm8((q >> 8) & 255u) - 0.5f) * 2.0f,
                   (unorm8((q >> 16) & 255u) - 0.0f) * 1.0f);
  } else {
    const TexCoord tc = bilinear_coord((int)S.atlas_res, (int)S.atlas_res, tcx, tcy, true);
    Tap4 qd, qe, qr, qn;
    if (ts0.x == TEXSET_QUAD) {
      // the four layers interleaved texel by texel (4 x 2-texel tiles of 16-byte texels): one 16-byte load per tap
      // brings all four layers, and the footprint lies in 1.9 lines instead of 4 x 1.4
      const uint4 *img = S.atlas4 + (size_t)ts0.y * 8u;
      const int tiles_x = ((int)S.atlas_res + 3) >> 2;
      const uint4 t00 = img[tile_offset<2, 1>(tc.i0, tc.j0, tiles_x)], t10 = img[tile_offset<2, 1>(tc.i1, tc.j0, tiles_x)];
      const uint4 t01 = img[tile_offset<2, 1>(tc.i0, tc.j1, tiles_x)], t11 = img[tile_offset<2, 1>(tc.i1, tc.j1, tiles_x)];
      qd = Tap4{t00.x, t10.x, t01.x, t11.x, tc.a, tc.b};
      qe = Tap4{t00.y, t10.y, t01.y, t11.y, tc.a, tc.b};
      qr = Tap4{t00.z, t10.z, t01.z, t11.z, tc.a, tc.b};
      qn = Tap4{t00.w, t10.w, t01.w, t11.w, tc.a, tc.b};
    } else {
      // separate single-layer images (a set with one image layer, or sets beyond the interleaving budget): all loads
      // of the image layers are issued before the first texel is decoded
      const uint4 ts2 = tset[2];
      const TapGeom tg = tap_geom(tc, (int)S.atlas_res);
      qd = layer_taps(S, tg, ts2.x, ts1.x);
      qe = layer_taps(S, tg, ts2.y, ts1.y);
      qr = layer_taps(S, tg, ts2.z, ts1.z);
      qn = layer_taps(S, tg, ts2.w, ts1.w);
    }
    texDiffuse = v3(tap_channel(qd, 0), tap_channel(qd, 1), tap_channel(qd, 2));
    texEmissive = v3(tap_channel(qe, 0), tap_channel(qe, 1), tap_channel(qe, 2));
    metallic = tap_channel(qr, 0);
    rough = tap_channel(qr, 1);
    texNormal = v3((tap_channel(qn, 0) - 0.5f) * 2.0f, (tap_channel(qn, 1) - 0.5f) * 2.0f,
                   (tap_channel(qn, 2) - 0.0f) * 1.0f);
  }
  rough = rough * rough;
  float seed = fma_(origin.z, 4761.52835f, ((origin.x * randBase) * origin.y) * 1.396529836f);
  V3 baryNormal = bary3(w, n1, n2, n3);
  V3 baryTangent = bary3(w, t1, t2, t3);
  V3 baryBitangent = bary3(w, b1, b2, b3);
  V3 macroNormal = normalize(
      v3(fma_(texNormal.z, baryNormal.x, fma_(texNormal.y, baryBitangent.x, texNormal.x * baryTangent.x)),
         fma_(texNormal.z, baryNormal.y, fma_(texNormal.y, baryBitangent.y, texNormal.x * baryTangent.y)),
         fma_(texNormal.z, baryNormal.z, fma_(texNormal.y, baryBitangent.z, texNormal.x * baryTangent.z))));
  bool inside = dot(-rd, baryNormal) < 0.0f;
  float nsx = inside ? ior : 1.0f, nsy = inside ? 1.0f : ior;
  if (inside) macroNormal = -macroNormal;
  V3 off = (macroNormal * EPSILON) * 2.0f;
  V3 ro = origin + off;

  V3 thr = ps.thr;
  ps.color = v3(fma_((thr.x * texEmissive.x) * texDiffuse.x, 30.0f, ps.color.x),
                fma_((thr.y * texEmissive.y) * texDiffuse.y, 30.0f, ps.color.y),
                fma_((thr.z * texEmissive.z) * texDiffuse.z, 30.0f, ps.color.z));
  V3 incident = -rd;
  V3 envThroughput, bsdfThroughput;
  float bsdfPdf;
  V3 microNormal = sample_microfacet(macroNormal, rough, seed);
  V3 envDir;
  float envPdf;
  sample_env(S, envTheta, seed, envDir, envPdf);
  float cosEnv = dot(macroNormal, envDir);
  float F = schlick(incident, microNormal, nsx, nsy);
  bool specular = fma_(1.0f, metallic, F * (1.0f - metallic)) > rnd(seed);
  bool refracted = false;
  // bsdfThroughput = numB * clamp(n . rd) / bsdfPdf and envThroughput = numE * clamp(n . envDir) / envPdf in the reflect
  // and the Lambert branch (tracer.fs:476-480, 490-495): the branches leave the numerators, the six divisions happen once
  // behind them (a wave usually holds lanes of both branches and used to run both copies)
  V3 numB = v3(1.0f, 1.0f, 1.0f), numE = v3(0.0f, 0.0f, 0.0f);
  if (specular) {
    V3 I = -incident;
    float k = 2.0f * dot(microNormal, I);
    rd = v3(fma_(-k, microNormal.x, I.x), fma_(-k, microNormal.y, I.y), fma_(-k, microNormal.z, I.z));
    bsdfPdf = gtr2_pdf(incident, macroNormal, rough, rd);
    numB = eval_specular(incident, macroNormal, texDiffuse, metallic, rough, rd);
    numE = eval_specular(incident, macroNormal, texDiffuse, metallic, rough, envDir);
  } else if (dielectric >= 0.0f) {
    bsdfPdf = 1.0f;
    ro = origin - off;
    V3 I = -incident;
    float eta = nsx / nsy;
    float dNI = dot(microNormal, I);
    float kk = 1.0f - (eta * eta) * (1.0f - dNI * dNI);
    if (kk < 0.0f) rd = v3(0.0f, 0.0f, 0.0f);
    else {
      float sc = fma_(eta, dNI, sqrt_(kk));
      rd = v3(fma_(eta, I.x, -(sc * microNormal.x)), fma_(eta, I.y, -(sc * microNormal.y)),
              fma_(eta, I.z, -(sc * microNormal.z)));
    }
    refracted = true; // tracer.fs:488 `i--`
  } else {
    rd = sample_lambert(macroNormal, seed);
    bsdfPdf = abs_(dot(rd, macroNormal)) * INV_PI_F;
    numB = numE = v3(texDiffuse.x * INV_PI_F, texDiffuse.y * INV_PI_F, texDiffuse.z * INV_PI_F);
  }
  if (refracted) {
    bsdfThroughput = v3(1.0f, 1.0f, 1.0f);
    envThroughput = v3(0.0f, 0.0f, 0.0f);
  } else {
    const float cl = clamp_(dot(macroNormal, rd), 0.0f, 1.0f), ce = clamp_(cosEnv, 0.0f, 1.0f);
    bsdfThroughput = v3((numB.x * cl) / bsdfPdf, (numB.y * cl) / bsdfPdf, (numB.z * cl) / bsdfPdf);
    envThroughput = v3((numE.x * ce) / envPdf, (numE.y * ce) / envPdf, (numE.z * ce) / envPdf);
  }
  if (inside) { // tracer.fs:497
    bsdfThroughput = v3(max_(1.0f - (((1.0f - texDiffuse.x) * tHit) * dielectric), 0.0f),
                        max_(1.0f - (((1.0f - texDiffuse.y) * tHit) * dielectric), 0.0f),
                        max_(1.0f - (((1.0f - texDiffuse.z) * tHit) * dielectric), 0.0f));
  }
  // misWeights (tracer.fs:194-203)
  if (envPdf > EPSILON && bsdfPdf > EPSILON) {
    float a2 = envPdf * envPdf, b2 = bsdfPdf * bsdfPdf, sum = a2 + b2;
    ps.wx = a2 / sum;
    ps.wy = b2 / sum;
  } else {
    ps.wx = 1.0f;
    ps.wy = 0.0f;
  }
  ps.hasShadow = (dielectric < 0.0f && cosEnv > 0.0f); // tracer.fs:500
  ps.envDir = envDir;
  ps.pend = thr * envThroughput;
  ps.ro = ro;
  ps.rd = rd;
  // tracer.fs:508: accumulatedReflectance *= bsdfThroughput happens after both traces; the
  // shadow contribution above already captured the pre-update value in `pend`.
  ps.thr = thr * bsdfThroughput;
  if (!refracted) ps.bounce++;
  ps.iters++;
  ps.primary = false;
}

// One S step of a live path (tracer.fs:500-512 + the loop bound of 446): consume the
// shadow result hitA and the primary/extension result (tB, hitB); either shade the next
// bounce (new rays in ps) or finish.  Returns true when the path is finished; ps.color
// then holds the un-clamped sample colour.  Shared by the megakernel and the wavefront
// pipeline so both run the same arithmetic in the same order.
// ... its first half: everything but the shading of a live hit.  Returns true when the path is finished.
template <bool COUNT>
FM_DEV bool consume_rays(const DScene &S, Path &ps, int hitA, int hitB, float envTheta, uint32_t numBounces, Counters &cnt) {
  // NEE result (tracer.fs:500-505)
  if (ps.hasShadow && hitA == -1) {
    V3 es = env_sample<COUNT>(S, ps.envDir, envTheta, cnt);
    ps.color = v3(fma_(ps.pend.x * es.x, ps.wx, ps.color.x), fma_(ps.pend.y * es.y, ps.wx, ps.color.y),
                  fma_(ps.pend.z * es.z, ps.wx, ps.color.z));
  }
  ps.hasShadow = false;
  if (hitB == -1) {
    // tracer.fs:442-443 (primary: weight 1, reflectance 1) / 509-512
    V3 es = env_sample<COUNT>(S, ps.rd, envTheta, cnt);
    float wgt = ps.primary ? 1.0f : ps.wy;
    V3 thr = ps.primary ? v3(1.0f, 1.0f, 1.0f) : ps.thr;
    ps.color = v3(fma_(thr.x * es.x, wgt, ps.color.x), fma_(thr.y * es.y, wgt, ps.color.y),
                  fma_(thr.z * es.z, wgt, ps.color.z));
    return true;
  }
  if (ps.bounce >= (int)numBounces || ps.iters >= MAX_PATH_ITERS) return true; // tracer.fs:446 bound, live hit
  return false;
}
template <bool COUNT>
FM_DEV bool advance_path(const DScene &S, Path &ps, int hitA, float tB, int hitB, float randBase, float envTheta,
                         uint32_t numBounces, Counters &cnt) {
  if (consume_rays<COUNT>(S, ps, hitA, hitB, envTheta, numBounces, cnt)) return true;
  shade_hit<COUNT>(S, ps, tB, hitB, randBase, envTheta, cnt);
  return false;
}

// tracer.fs:515-517: clamp + running mean with weight tick
FM_DEV float4 accumulate_sample(float4 prev, V3 color, uint32_t tick) {
  float ft = (float)tick;
  float den = ft + 1.0f;
  float cr = clamp_(color.x, 0.0f, 1024.0f), cg = clamp_(color.y, 0.0f, 1024.0f), cb = clamp_(color.z, 0.0f, 1024.0f);
  float4 o4;
  o4.x = fma_(prev.x, ft, cr) / den;
  o4.y = fma_(prev.y, ft, cg) / den;
  o4.z = fma_(prev.z, ft, cb) / den;
  o4.w = 1.0f;
  return o4;
}

// Work index -> pixel.  The frame is cut into tile x tile pixel tiles dealt
// round-robin to shards; inside a tile pixels are enumerated in 8x8 blocks so
// that the 64 lanes of a wave start on a compact screen patch.
template <class P>
FM_DEV bool work_to_pixel(const P &p, uint32_t idx, uint32_t &x, uint32_t &y) {
  uint32_t tile = p.tile;
  uint32_t per_tile = tile * tile;
  uint32_t k = idx / per_tile, local = idx - k * per_tile;
  uint32_t g = p.shard + k * p.n_shards;
  if (g >= p.tiles_x * p.tiles_y) return false;
  uint32_t tx = g % p.tiles_x, ty = g / p.tiles_x;
  uint32_t sub = local >> 6, l = local & 63u;
  uint32_t subs_x = tile >> 3;
  uint32_t sx = sub % subs_x, sy = sub / subs_x;
  x = tx * tile + sx * 8 + (l & 7u);
  y = ty * tile + sy * 8 + (l >> 3);
  return x < p.vw && y < p.vh;
}

// Sample number g of a run (pixel-major: n_batch ticks per work index) -> work index.
FM_DEV uint32_t wf_work_index(const WfP &p, uint32_t g) { return g / p.n_batch; }

// ---------------------------------------------------------------------------
// The path-trace kernel: tracer.fs main() (436-518) over the whole frame.
// ---------------------------------------------------------------------------
template <bool GEN_RAYS, bool COUNT>
__global__ __launch_bounds__(BLOCK_THREADS) void k_trace(const TraceP p) {
  extern __shared__ int lds_stack[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const DScene &S = p.scene;
  int *stack = lds_stack + (size_t)wave * S.stack_n * WAVE + lane;

  Counters cnt = {0, 0, 0, 0, 0, 0};
  Path ps;
  ps.pix = -1;
  ps.hasShadow = false;
  ps.primary = true;
  ps.bounce = 0;
  ps.iters = 0;
  ps.wx = ps.wy = 0.0f;
  ps.lag = 0u;
  ps.ro = ps.rd = ps.thr = ps.color = ps.envDir = ps.pend = v3(0.0f, 0.0f, 0.0f);

  // traversal results of the previous T phase
  int hitA = -1, hitB = -1;
  float tB = MAX_T;

  const uint32_t total_work = p.n_owned_tiles * p.tile * p.tile;
  uint32_t pool_next = 0, pool_end = 0; // wave-uniform
  bool exhausted = false;               // wave-uniform

  while (true) {
    // ================= S phase =================
    bool need_pixel = (ps.pix < 0);
    if (ps.pix >= 0) {
      if (advance_path<COUNT>(S, ps, hitA, tB, hitB, p.rand_base, p.env_theta, p.num_bounces, cnt)) {
        p.accum[ps.pix] = accumulate_sample(p.accum[ps.pix], ps.color, p.tick);
        ps.pix = -1;
        need_pixel = true;
      }
    }
    // ---- refill idle lanes from the wave's pixel pool ---------------------
    while (true) {
      unsigned long long need = __ballot(need_pixel);
      if (need == 0ull) break;
      uint32_t avail = pool_end - pool_next;
      if (avail == 0u) {
        if (exhausted) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(p.work_counter, WORK_CHUNK);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= total_work) { exhausted = true; break; }
        pool_next = base;
        pool_end = min(base + WORK_CHUNK, total_work);
        continue;
      }
      uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      uint32_t want = (uint32_t)__popcll(need);
      uint32_t take = want < avail ? want : avail;
      if (need_pixel && rank < take) {
        uint32_t x, y;
        if (work_to_pixel(p, pool_next + rank, x, y)) {
          ps.pix = (int)(y * p.W + x);
          if (COUNT) cnt.samples++;
          if (GEN_RAYS) {
            camera_ray(x, y, p.W, p.H, p.cam, p.rand_base_cam, ps.ro, ps.rd);
          } else {
            float4 po = p.ray_pos[ps.pix], di = p.ray_dir[ps.pix];
            ps.ro = v3(po.x, po.y, po.z);
            ps.rd = v3(di.x, di.y, di.z);
          }
          ps.thr = v3(1.0f, 1.0f, 1.0f);
          ps.color = v3(0.0f, 0.0f, 0.0f);
          ps.bounce = 0;
          ps.iters = 0;
          ps.primary = true;
          ps.hasShadow = false;
          need_pixel = false;
        }
      }
      pool_next += take;
    }
    if (__ballot(ps.pix >= 0) == 0ull) break; // wave drained and no work left

    // ================= T phase =================
    if (ps.pix >= 0) {
      trace_rays<COUNT, false>(S, stack, ps.ro, ps.hasShadow, ps.envDir, ps.rd, hitA, tB, hitB, cnt);
    }
  }

  if (COUNT) {
    unsigned long long v[6] = {cnt.samples, cnt.rays, cnt.steps, cnt.leaves, cnt.shades, cnt.envs};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      unsigned long long x = v[i];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
      if (lane == 0) atomicAdd(p.counters + i, x);
    }
  }
}

// ===========================================================================
// Wavefront pipeline: primary -> [trace <-> logic] x rounds [-> tail] -> resolve
// (layout and slot numbering: fspt_device.hpp)
// ===========================================================================
#ifndef WF_TRACE_CHUNK
#define WF_TRACE_CHUNK 128u // cap of the pool chunk (paths).  With the pool heads in memory lines of their own a draw is cheap
// (profiles/r02/ab_trace_pool_heads.log): caps 64 / 128 / 256 -> 3 810 / 3 828 / 3 750 Msamples/s at 20-tick batches,
// 4 750 / 4 775 / 4 795 at 128.  (With all heads in ONE 64-byte line - same-line atomics serialise at the memory side -
// 128 cost 20 % and 512 was the optimum.)
#endif
// measured on C2 (profiles/r01): 1 -> 0.462, 8 -> 0.348, 16 -> 0.338, 24 -> 0.336, 32 -> 0.343 ms per tick
#ifndef WF_INTERIOR_MIN
#define WF_INTERIOR_MIN 24 // (round 6, on the trace kernel without its scratch reload: 16 -> 24: +0.9 % on the 70 k scene at 20 ticks, +3 % on the 1 M one; 8 / 20 / 32 lose or tie - profiles/r06/scan_constants*.log)
#endif
#ifndef WF_LOGIC_THREADS
#define WF_LOGIC_THREADS 512
#endif
// paths per thread of one block iteration of the logic kernel (classification is cheap; a larger chunk gives the dense
// shading phase more whole waves of work)
#ifndef WF_LOGIC_U
#define WF_LOGIC_U 8
#endif
// Block iterations of the primary and the logic launch by TICKET instead of a static stride (see k_wf_primary): 1 / 0
// (measurement hooks: profiles/r05/ab_primary_tickets_*.log, ab_logic_tickets_*.log)
#ifndef WF_PRIMARY_TICKETS
#define WF_PRIMARY_TICKETS 1
#endif
#ifndef WF_LOGIC_TICKETS
#define WF_LOGIC_TICKETS 1
#endif
// Grid sizes (measurement hooks, profiles/r05/ab_grid_sizes_*.log): 256-thread blocks per CU of a trace launch (8: 6 are
// resident at 80 registers; 6 / 7: trace launch -2 % / +-0 on the 70 k scene, +-0 on the 1 M one, whole job +-0), and the
// factor over what is resident for the primary / logic launches (2; 1: +-0 with tickets)
#ifndef WF_TRACE_GRID_PER_CU
#define WF_TRACE_GRID_PER_CU 8u
#endif
// ... and the trace grid capped at what is resident (blocks per CU the registers allow; 0 = no cap): trace launches -2.5 %
// on the 70 k scene (20 ticks), -0.7 % (128 ticks), +-0 on the 1 M one (profiles/r05/ab_trace_grid_resident_*.log)
#ifndef WF_TRACE_GRID_RESIDENT
#define WF_TRACE_GRID_RESIDENT 6
#endif
#ifndef WF_PL_OVERSUB
#define WF_PL_OVERSUB 2u
#endif
// k_wf_primary takes ONE path per thread and block iteration: r01: 1 / 2 / 4 / 8 -> 0.327 / 0.320 / 0.331 / 0.357 ms per tick;
// r02: 1 / 2 / 4 -> 15.0 / 15.4 / 16.9 ms per 128 ticks (one is the only count without register spills at 4 waves/SIMD;
// profiles/r02/ab_primary_paths_per_thread.log).  (The top of the tree in LDS, as in k_wf_trace, makes this launch 10 %
// SLOWER: its rays are coherent, their node fetches hit L1 anyway; profiles/r02/ab_primary_lds_top.log)

// Path state is streamed (touched once per round) through plain loads and stores: non-temporal variants were measured
// slower (logic kernel +5 %, stores box-dependent; profiles/r01, profiles/r02/ab_tunables.log).
FM_DEV float4 ld4(const float4 *p) { return *p; }
FM_DEV void st4(float4 *p, float4 v) { *p = v; }
FM_DEV float2 ld2(const float2 *p) { return *p; }
FM_DEV void st2(float2 *p, float2 v) { *p = v; }
// 12-byte (RGB) elements of the finished-sample array: dword-aligned three-dword accesses
typedef float f3a __attribute__((ext_vector_type(3), aligned(4)));
FM_DEV void st3(float *p, V3 v) {
  f3a w = {v.x, v.y, v.z};
  *reinterpret_cast<f3a *>(p) = w;
}
FM_DEV V3 ld3(const float *p) {
  const f3a v = *reinterpret_cast<const f3a *>(p);
  return v3(v.x, v.y, v.z);
}
FM_DEV int ldi(const int *p) { return *p; }
FM_DEV void sti(int *p, int v) { *p = v; }

// The wave's index in its block as a SCALAR (round 6): threadIdx.x / 64 is the same for the 64 lanes of a wave, but the
// compiler does not know that, so everything derived from it - the wave's pool-head address, its LDS stack base - lived in
// vector registers.  In k_wf_trace the 64-bit head address was spilled to scratch and RELOADED AFTER EVERY LEAF VISIT (the
// leaf code needs the registers): one more vector-memory instruction per leaf visit in the kernel that is bound by exactly
// those; k_wf_tail spilled three such values.  With the readfirstlane both kernels use no scratch at all: trace launches
// -5 % (70 k triangles), -4 % (1 M), whole job +1.7 ... +2.2 % (profiles/r06/ab_small3_*.log, isa_trace_spill.txt).
FM_DEV int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)); }
FM_DEV uint32_t lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- path state <-> registers ---------------------------------------------------------------
FM_DEV uint32_t pack_flags(const Path &ps, bool col_zero) {
  return ((uint32_t)ps.bounce & 255u) | (((uint32_t)ps.iters & 255u) << 8) | (ps.primary ? WF_FLAG_PRIMARY : 0u) |
         (ps.hasShadow ? WF_FLAG_SHADOW : 0u) | (col_zero ? WF_FLAG_COLZERO : 0u) | ((ps.lag & WF_LAG_MASK) << WF_LAG_SHIFT);
}
// state of a surviving path -> index k of `o`; the colour array is only written while the colour is non-zero
// (it is +0 until the first light arrives), D / P only when the path has a NEE shadow ray
FM_DEV void store_path(const WfSet &o, uint32_t k, const Path &ps, uint32_t slot) {
  const bool col_zero = ps.color.x == 0.0f && ps.color.y == 0.0f && ps.color.z == 0.0f;
  st4(o.A + k, make_float4(ps.ro.x, ps.ro.y, ps.ro.z, __uint_as_float(slot)));
  st4(o.B + k, make_float4(ps.rd.x, ps.rd.y, ps.rd.z, __uint_as_float(pack_flags(ps, col_zero))));
  st4(o.C + k, make_float4(ps.thr.x, ps.thr.y, ps.thr.z, ps.wy));
  if (!col_zero) st4(o.E + k, make_float4(ps.color.x, ps.color.y, ps.color.z, 0.0f));
  if (ps.hasShadow) {
    st4(o.D + k, make_float4(ps.envDir.x, ps.envDir.y, ps.envDir.z, ps.wx));
    st4(o.P + k, make_float4(ps.pend.x, ps.pend.y, ps.pend.z, 0.0f));
  }
}
// path k of `in` with the traversal results of its rays; returns the slot
FM_DEV uint32_t load_path(const WfSet &in, uint32_t k, Path &ps, const int *shadow_hit, int &hitA) {
  const float4 ro = ld4(in.A + k), rd = ld4(in.B + k), th = ld4(in.C + k);
  const uint32_t flags = __float_as_uint(rd.w);
  float4 co = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (!(flags & WF_FLAG_COLZERO)) co = ld4(in.E + k);
  ps.ro = v3(ro.x, ro.y, ro.z);
  ps.rd = v3(rd.x, rd.y, rd.z);
  ps.thr = v3(th.x, th.y, th.z);
  ps.wy = th.w;
  ps.color = v3(co.x, co.y, co.z);
  ps.bounce = (int)(flags & 255u);
  ps.iters = (int)((flags >> 8) & 255u);
  ps.primary = (flags & WF_FLAG_PRIMARY) != 0u;
  ps.hasShadow = (flags & WF_FLAG_SHADOW) != 0u;
  ps.lag = (flags >> WF_LAG_SHIFT) & WF_LAG_MASK;
  ps.pix = 0;
  ps.wx = 0.0f;
  ps.envDir = v3(0.0f, 0.0f, 0.0f);
  ps.pend = v3(0.0f, 0.0f, 0.0f);
  hitA = -1;
  if (ps.hasShadow) {
    const float4 sd = ld4(in.D + k), pe = ld4(in.P + k);
    ps.envDir = v3(sd.x, sd.y, sd.z);
    ps.wx = sd.w;
    ps.pend = v3(pe.x, pe.y, pe.z);
    if (shadow_hit) hitA = ldi(shadow_hit + k);
  }
  return __float_as_uint(ro.w);
}

// a path whose traversal was suspended: its state, bit for bit, from index i of `in` to index k of `o`, flagged and
// with one more round of lag in the flags word
FM_DEV void carry_path(const WfSet &in, const WfSet &o, uint32_t i, uint32_t k) {
  const float4 a = ld4(in.A + i), c = ld4(in.C + i);
  float4 b = ld4(in.B + i);
  uint32_t fl = __float_as_uint(b.w);
  const uint32_t lag = (fl >> WF_LAG_SHIFT) & WF_LAG_MASK;
  fl = (fl & ~(WF_LAG_MASK << WF_LAG_SHIFT)) | ((lag < WF_LAG_MASK ? lag + 1u : lag) << WF_LAG_SHIFT) | WF_FLAG_SUSP;
  st4(o.A + k, a);
  st4(o.C + k, c);
  if (!(fl & WF_FLAG_COLZERO)) st4(o.E + k, ld4(in.E + i));
  if (fl & WF_FLAG_SHADOW) { st4(o.D + k, ld4(in.D + i)); st4(o.P + k, ld4(in.P + i)); }
  b.w = __uint_as_float(fl);
  st4(o.B + k, b);
}

// ---- carry: the paths whose traversal the previous trace launch suspended move on to the next state set ---------------
// One thread per record (a handful to a few thousand per launch): same state at a new index of the set this round's
// logic launch writes (same counter), marked - the next trace launch resumes the record instead of starting the path's
// rays afresh - one round of lag counted, and the record learns the new index.  A kernel of its own, launched in front of
// k_wf_logic: inside the logic kernel the few lines cost the shading loop registers (9 spilled, profiles/r03).
// (Round 5: the same work is also done by WfP::carry_blocks trailing blocks of the logic LAUNCH - a branch at the top of
// k_wf_logic that returns before the shading code, so its registers are not the shading loop's - which saves a launch
// and its gap per round; the stand-alone kernel remains for hosts that ask for it.)
FM_DEV void carry_records(const WfP &p, uint32_t first, uint32_t stride) {
  const uint32_t n = p.counts[p.cnt_in].n_susp;
  int *rec = p.susp[p.cnt_in & 1u];
  // one reservation per WAVE (same-address atomics serialise at the memory side, ~15 ns each: a few thousand records,
  // one atomic each, were most of this launch's 11-27 us, profiles/r05/launch_list_c2.txt)
  for (uint32_t r = first;; r += stride) {
    const bool act = r < n;
    const unsigned long long m = __ballot(act);
    if (m == 0ull) break;
    uint32_t base = 0;
    if ((threadIdx.x & (WAVE - 1)) == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&p.counts[p.cnt_out].n_ext, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), WAVE);
    if (act) {
      int *q = rec + (size_t)r * p.susp_stride;
      const uint32_t k_new = base + lane_rank(m);
      carry_path(p.set[p.set_in], p.set[p.set_out], (uint32_t)q[0], k_new);
      q[0] = (int)k_new;
    }
  }
}
__global__ __launch_bounds__(BLOCK_THREADS) void k_wf_carry(const WfP p) {
  carry_records(p, blockIdx.x * BLOCK_THREADS + threadIdx.x, gridDim.x * BLOCK_THREADS);
}

// ---- trace: intersectScene for the rays of one round; persistent waves, per-lane refill -------
// One item per live path k of the round: its NEE shadow ray (if it has one) and then its extension ray, traced by the
// same lane one after the other (they share their origin: one state fetch, and no items that turn out to be empty).
// A lane whose path is done writes its results and takes the next item from the wave's pool, so all 64 lanes keep
// traversing.
// The pool: device-scope atomics execute at the memory side on this part (the XCDs' L2s are not coherent with each
// other) - one address sustains ~65 M atomics/s, and device-scope loads are as slow (measured: a peek before every
// draw cost more than it saved, profiles/r02).  So: every wave's first chunk is its own (no atomic); the chunks beyond
// are dealt round-robin to WF_HEADS stripes (chunk c belongs to stripe c % WF_HEADS), every wave draws from the one
// stripe it is assigned to and never from another: the stripes hold the same mix of cheap and expensive image regions,
// so they run dry together, each head takes 1/WF_HEADS of the atomics, and a wave ends on ONE failed atomic.
// 6 waves/SIMD minimum -> <= 80 VGPRs: measured best of {5, 7, 8(spills)} waves/SIMD (profiles/r01)
#ifndef WF_TRACE_WAVES
#define WF_TRACE_WAVES 6
#endif
#ifndef WF_TRACE_RAY_ITEMS
#define WF_TRACE_RAY_ITEMS 4u // x resident lanes: the last paths of a launch whose two rays are separate items
#endif
#ifndef WF_TRACE_FINE
#define WF_TRACE_FINE 8u // the last 1/8 of a launch's paths are dealt out in 64-path chunks
#endif
// WIDE: the interior loop walks the two-level nodes (fspt_device.hpp "quad"): two of the reference's steps per memory
// round trip at twice the lane-requests per fetch - for the launches that are bound by the LATENCY of their rays' dependent
// chains (few paths; a scene that does not fit the L2), not by the request rate of the vector-memory pipeline.
template <bool COUNT, bool ANYHIT, bool WIDE = false>
__global__ __launch_bounds__(BLOCK_THREADS, WF_TRACE_WAVES) void k_wf_trace(const WfP p) {
  extern __shared__ int lds_stack[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = wave_index();
  const DScene &S = p.scene;
  // one more LDS entry per lane than the tree needs: the lane's finished shadow result while its extension ray is traced
  // (a suspended traversal's record carries it along)
  const uint32_t sn = S.stack_n + 4u; // + 3: the extension ray's direction while the shadow ray is traced
  int *stack = lds_stack + (size_t)wave * sn * WAVE + lane;
  constexpr int REC_F4 = WIDE ? QUAD_F4 : NODE_F4;
  const float4 *__restrict__ nodes = WIDE ? S.quads : S.nodes;
  const float *__restrict__ leaves = S.leaves;
  const uint32_t leaf_size = S.leaf_size;
  const WfSet st = p.set[p.set_out];
  WfCounts *cn = p.counts + p.cnt_out;
  const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
  // suspended traversals (fspt_device.hpp): the records the previous trace launch wrote are this launch's FIRST items
  // (a launch with budget 0 - the last one of a batch - still RESUMES what the previous launch suspended)
  const bool susp_on = !COUNT && p.susp_budget != 0u;
  const uint32_t n_res = (!COUNT && p.susp[0] != nullptr) ? p.counts[p.cnt_in].n_susp : 0u;
  const int *__restrict__ rec_in = p.susp[p.cnt_in & 1u];
  int *rec_out = p.susp[p.cnt_out & 1u];
  const uint32_t total = cn->n_ext + n_res;

  // top of the tree in LDS (behind the waves' stacks): every ray walks these nodes, and a fetch from LDS does not
  // occupy the vector-memory pipeline that bounds this kernel
  const int n_top = (int)p.lds_top;
  float4 *top = reinterpret_cast<float4 *>(lds_stack + (size_t)WAVES_PER_BLOCK * sn * WAVE);
  if (n_top > 0) {
    for (int i = threadIdx.x; i < n_top * REC_F4; i += BLOCK_THREADS) top[i] = nodes[i];
    __syncthreads();
  }

  // The LAST paths of a launch are dealt out as RAY items (item 2i = the extension ray of path i, item 2i + 1 = its NEE
  // shadow ray, empty when it has none): a launch ends on the dependent chain of its longest-running item, and a path
  // item is two rays one after the other.  Split, the two rays of the late paths run side by side on two lanes (any two:
  // their results go to hit[] and shadow_hit[] independently), and the launch ends after ONE long ray, not two.  Small
  // launches consist of ray items only.  (Costs a second state fetch per split path; WF_TRACE_RAY_ITEMS x the resident lanes.)
  // (with suspended traversals the end of a launch is short anyway, and a record describes a whole path item)
  const uint32_t split_paths = (susp_on || n_res != 0u) ? 0u : min(total, (uint32_t)WF_TRACE_RAY_ITEMS * n_waves * (uint32_t)WAVE);
  const uint32_t path_items = total - split_paths; // paths [0, path_items): one item per path
  // pool chunk: large while paths are plentiful (few atomics), one wave-load when they are scarce
  // (late rounds), so that every resident wave gets work
  uint32_t chunk = (total / (n_waves * 4u)) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > WF_TRACE_CHUNK ? WF_TRACE_CHUNK : chunk);
  // Three kinds of chunk, handed out in this order: the first (1 - 1/WF_TRACE_FINE) of the path items in chunks of
  // `chunk`, the rest of them in chunks of 64, then the ray items in chunks of 64 - the launch ends on fine-grained work:
  // the imbalance at its end is one chunk's worth of time (512 paths on one wave = 8 per lane, ~300 us) unless the last
  // chunks are small.
  const uint32_t big_paths = chunk > 64u ? ((path_items - path_items / WF_TRACE_FINE) / chunk) * chunk : 0u;
  const uint32_t n_big = chunk > 64u ? big_paths / chunk : 0u;
  const uint32_t n_fine = (path_items - big_paths + 63u) / 64u;
  const uint32_t item_end = path_items + 2u * split_paths; // items [path_items, item_end): ray items
  const uint32_t n_chunks = n_big + n_fine + (2u * split_paths + 63u) / 64u;
  auto chunk_range = [&](uint32_t c, uint32_t &lo, uint32_t &hi) {
    if (c < n_big) { lo = c * chunk; hi = lo + chunk; }
    else if (c < n_big + n_fine) { lo = big_paths + (c - n_big) * 64u; hi = min(lo + 64u, path_items); }
    else { lo = path_items + (c - n_big - n_fine) * 64u; hi = min(lo + 64u, item_end); }
  };
  const uint32_t wave_id = blockIdx.x * WAVES_PER_BLOCK + wave;
  const uint32_t stripe = wave_id % WF_HEADS;
  uint32_t pool_next = 0, pool_end = 0;
  if (wave_id < n_chunks) chunk_range(wave_id, pool_next, pool_end); // chunk `wave_id` is the wave's own
  bool exhausted = n_chunks <= n_waves; // nothing beyond the waves' own chunks

  uint32_t c_rays = 0, c_steps = 0, c_leaves = 0, c_lds = 0;
  uint32_t starve = 0; // traversal steps since the wave found its pool empty (wave-uniform)
  bool idle = true;
  bool is_shadow = false;
  // state index of the lane's path; bit 31: the path's bounce budget is used up (a hit of its extension ray will not be
  // shaded); bits 29-30: what the item covers - 0 both rays, 1 the extension ray, 2 the shadow ray, 3 nothing (the shadow
  // item of a path without one; the lane still goes through the finish step below)
  uint32_t path = 0;
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), inv = v3(0, 0, 0);
  float t = MAX_T;
  int hit = -1, cur = REF_SENTINEL, sp = 0;

  while (true) {
    // ---- refill idle lanes ----
    while (true) {
      unsigned long long need = __ballot(idle);
      if (need == 0ull) break;
      uint32_t avail = pool_end - pool_next;
      if (avail == 0u) {
        if (exhausted) break;
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(&p.heads[((size_t)p.cnt_out * WF_HEADS + stripe) * WF_HEAD_STRIDE], 1u);
        j = __builtin_amdgcn_readfirstlane(j);
        const unsigned long long c = (unsigned long long)n_waves + (unsigned long long)j * WF_HEADS + stripe;
        if (c >= n_chunks) { exhausted = true; break; }
        chunk_range((uint32_t)c, pool_next, pool_end);
        continue;
      }
      uint32_t rank = lane_rank(need);
      uint32_t want = (uint32_t)__popcll(need);
      uint32_t take = want < avail ? want : avail;
      if (idle && rank < take) {
        uint32_t k = pool_next + rank, mode = 0u;
        if (k >= path_items) { // ray item
          const uint32_t j = k - path_items;
          k = path_items + (j >> 1);
          mode = 1u + (j & 1u);
        }
        // the first n_res items are suspended traversals: the record names the path's state index
        const bool resume = k < n_res;
        const int *rec = rec_in + (size_t)k * p.susp_stride;
        if (resume) k = (uint32_t)rec[0]; else k -= n_res;
        const float4 ro = ld4(st.A + k), rd = ld4(st.B + k);
        const float4 sd = ld4(st.D + k); // fetched alongside (only meaningful when the path has a shadow ray)
        o = v3(ro.x, ro.y, ro.z);
        // the extension ray's direction waits in the lane's LDS column while the shadow ray is traced (3 registers that
        // the leaf code needs: with them in registers it spilled 16 bytes per lane and leaf visit)
        stack[(S.stack_n + 1u) * WAVE] = __float_as_int(rd.x); stack[(S.stack_n + 2u) * WAVE] = __float_as_int(rd.y); stack[(S.stack_n + 3u) * WAVE] = __float_as_int(rd.z);
        const uint32_t fl = __float_as_uint(rd.w);
        const bool has_shadow = (fl & WF_FLAG_SHADOW) != 0u;
        if (mode == 2u && !has_shadow) mode = 3u;
        // a path the logic launch carried over with a suspended traversal is this launch's resume item, not a new one
        if (!resume && (fl & WF_FLAG_SUSP)) mode = 3u;
        is_shadow = has_shadow && mode != 1u;
        t = MAX_T;
        hit = -1;
        cur = mode == 3u ? REF_SENTINEL : S.root_ref;
        sp = 0;
        if (resume) { // node, t, hit | sp, ray, finished shadow result | stack
          const int3 r0 = make_int3(rec[1], rec[2], rec[3]);
          const int2 r1 = *reinterpret_cast<const int2 *>(rec + 4);
          cur = r0.x; t = __int_as_float(r0.y); hit = r0.z;
          sp = r1.x & 255;
          is_shadow = ((r1.x >> 8) & 1) != 0;
          for (int i = 0; i < sp; ++i) stack[i * WAVE] = rec[WF_SUSP_HEADER + i];
          if (!is_shadow) { stack[S.stack_n * WAVE] = r1.y; if (has_shadow) sti(p.shadow_hit + k, r1.y); } // its shadow ray had finished before
        }
        d = is_shadow ? v3(sd.x, sd.y, sd.z) : v3(rd.x, rd.y, rd.z);
        inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool live = (fl & 255u) < p.num_bounces && ((fl >> 8) & 255u) < (uint32_t)MAX_PATH_ITERS;
        // bit 29 of a path item (mode 0): the traversal may be suspended (the path has not lagged WF_LAG_MAX rounds yet)
        const bool may_susp = susp_on && ((fl >> WF_LAG_SHIFT) & WF_LAG_MASK) < WF_LAG_MAX;
        path = k | (mode << 29) | (may_susp ? (1u << 29) : 0u) | (live ? 0u : 0x80000000u);
        idle = false;
        if (COUNT && mode != 3u) c_rays++;
      }
      pool_next += take;
    }
    // ---- a wave that can get no more work walks on for susp_budget steps, then parks its unfinished traversals ----
    const bool starved = susp_on && exhausted && pool_next == pool_end; // (wave-uniform)
    if (starved && starve >= p.susp_budget) {
      const bool park = !idle && cur != REF_SENTINEL && ((path >> 29) & 3u) == 1u; // a path item that may still lag
      const unsigned long long m = __ballot(park);
      if (m != 0ull) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cn->n_susp, (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (park) {
          const uint32_t k = path & 0x1fffffffu, rid = base + lane_rank(m);
          int *rec = rec_out + (size_t)rid * p.susp_stride;
          *reinterpret_cast<int4 *>(rec) = make_int4((int)k, cur, __float_as_int(t), hit);
          *reinterpret_cast<int2 *>(rec + 4) = make_int2(sp | (is_shadow ? 256 : 0), stack[S.stack_n * WAVE]);
          for (int i = 0; i < sp; ++i) rec[WF_SUSP_HEADER + i] = stack[i * WAVE];
          st2(p.hit + k, make_float2(__uint_as_float(rid), __int_as_float(WF_HIT_PENDING)));
          idle = true;
          cur = REF_SENTINEL;
        }
      }
      starve = 0u; // the lanes that had to stay (lagged paths) get another budget before the next look
    }
    if (__ballot(!idle) == 0ull) break;

    // ---- interior nodes ----
    // Leave the loop once fewer than WF_INTERIOR_MIN lanes are still descending, so the lanes that
    // already wait at a leaf (or for a refill) are not held up by a few long descents.
    while (true) {
      unsigned long long in = __ballot(cur >= 0);
      if (in == 0ull) break;
#if WF_INTERIOR_MIN > 1
      if ((uint32_t)__popcll(in) < WF_INTERIOR_MIN && __popcll(__ballot(!idle)) > __popcll(in)) break;
#endif
      if (cur >= 0) {
      if (COUNT) c_steps++;
      if constexpr (WIDE) {
        float4 a0, a1, a2, b0, b1, b2; int4 ar; int2 br;
        if (cur < n_top) {
          if (COUNT) c_lds++;
          typedef float lds_f4 __attribute__((ext_vector_type(4)));
          typedef int lds_i4 __attribute__((ext_vector_type(4)));
          typedef int lds_i2 __attribute__((ext_vector_type(2)));
          const __attribute__((address_space(3))) lds_f4 *n =
              (const __attribute__((address_space(3))) lds_f4 *)(top + cur * QUAD_F4);
          const lds_f4 x0 = n[0], x1 = n[1], x2 = n[2], y0 = n[4], y1 = n[5], y2 = n[6];
          const lds_i4 xr = *(const __attribute__((address_space(3))) lds_i4 *)(n + 3);
          const lds_i2 yr = *(const __attribute__((address_space(3))) lds_i2 *)(n + 7);
          a0 = make_float4(x0.x, x0.y, x0.z, x0.w); a1 = make_float4(x1.x, x1.y, x1.z, x1.w); a2 = make_float4(x2.x, x2.y, x2.z, x2.w);
          b0 = make_float4(y0.x, y0.y, y0.z, y0.w); b1 = make_float4(y1.x, y1.y, y1.z, y1.w); b2 = make_float4(y2.x, y2.y, y2.z, y2.w);
          ar = make_int4(xr.x, xr.y, xr.z, xr.w); br = make_int2(yr.x, yr.y);
        } else {
          quad_load(nodes + (size_t)cur * QUAD_F4, a0, a1, a2, ar, b0, b1, b2, br);
        }
        quad_step<COUNT>(a0, a1, a2, ar, b0, b1, b2, br, o, inv, t, stack, sp, cur, c_steps);
      } else {
      float4 n0, n1, n2;
      int2 n3;
      if (cur < n_top) {
        if (COUNT) c_lds++;
        // explicit LDS address space: keeps these ds_read_b128 from being merged with the global path into flat loads
        typedef float lds_f4 __attribute__((ext_vector_type(4)));
        typedef int lds_i2 __attribute__((ext_vector_type(2)));
        const __attribute__((address_space(3))) lds_f4 *n =
            (const __attribute__((address_space(3))) lds_f4 *)(top + cur * NODE_F4);
        lds_f4 a = n[0], b = n[1], c = n[2];
        lds_i2 r = *(const __attribute__((address_space(3))) lds_i2 *)(n + 3);
        n0 = make_float4(a.x, a.y, a.z, a.w); n1 = make_float4(b.x, b.y, b.z, b.w); n2 = make_float4(c.x, c.y, c.z, c.w);
        n3 = make_int2(r.x, r.y);
      } else {
        const float4 *n = nodes + (size_t)cur * NODE_F4;
        n0 = n[0]; n1 = n[1]; n2 = n[2];
        n3 = node_refs(n);
      }
      float tl, tr;
      node_test(n0, n1, n2, o, inv, tl, tr);
      bool hl = tl < t, hr = tr < t;
      bool swap = tl > tr;
      int nearRef = swap ? n3.y : n3.x;
      int farRef = swap ? n3.x : n3.y;
      if (hl && hr) {
        stack[sp * WAVE] = farRef;
        sp++;
        cur = nearRef;
      } else if (hl) {
        cur = n3.x;
      } else if (hr) {
        cur = n3.y;
      } else if (sp > 0) {
        sp--;
        cur = stack[sp * WAVE];
      } else {
        cur = REF_SENTINEL;
      }
      } // !WIDE
      }
      // (checked AFTER the step: every pass of the outer loop moves its rays on, also the ones that may not be parked)
      if (starved && ++starve >= p.susp_budget) break;
    }
    // ---- leaf ----
    if (!idle && cur < 0 && cur != REF_SENTINEL) {
      if (COUNT) { c_steps++; c_leaves++; }
      int ts = ~cur;
      process_leaf(leaves, leaf_size, ts, o, d, t, hit);
      if (sp > 0) { sp--; cur = stack[sp * WAVE]; }
      else cur = REF_SENTINEL;
      // NEE shadow rays: only `shadow.index == -1` is consumed (tracer.fs:502), so the first hit settles
      // the ray.  The ANYHIT = false counting variant keeps the reference's full closest-hit traversal, so its
      // work counters equal the oracle's (the algorithmic work of the reference algorithm).
      if (ANYHIT && is_shadow && hit != -1) cur = REF_SENTINEL;
    }
    // ---- finished rays: write the result; after the shadow ray of a path item comes its extension ray, then the lane is idle ----
    if (!idle && cur == REF_SENTINEL) {
      const uint32_t k = path & 0x1fffffffu, mode = (path >> 29) & 3u;
      if (mode == 3u) {
        idle = true; // empty item
      } else if (is_shadow) {
        sti(p.shadow_hit + k, hit);
        stack[S.stack_n * WAVE] = hit; // kept for a record, should the extension ray be suspended
        if (mode == 2u) {
          idle = true; // the extension ray is another lane's item
        } else {
          is_shadow = false;
          d = v3(__int_as_float(stack[(S.stack_n + 1u) * WAVE]), __int_as_float(stack[(S.stack_n + 2u) * WAVE]), __int_as_float(stack[(S.stack_n + 3u) * WAVE]));
          inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          t = MAX_T;
          hit = -1;
          cur = S.root_ref;
          sp = 0;
          if (COUNT) c_rays++;
        }
      } else {
        // index < -1 = hit, but the path ends here (tracer.fs:446 bound): the logic kernel classifies from this word alone
        st2(p.hit + k, make_float2(t, __int_as_float(hit != -1 && (path >> 31) ? WF_HIT_TERMINAL : hit)));
        idle = true;
      }
    }
  }
  if (COUNT) {
    unsigned long long v[4] = {c_rays, c_steps, c_leaves, c_lds};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned long long x = v[i];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
      if (lane == 0 && x) atomicAdd(p.counters + (i < 3 ? 1 + i : 6), x); // [6]: interior steps served from the LDS copy of the top of the tree
    }
  }
}

// LDS-staged tables of the shading kernels: the small read-only tables every shading event gathers from - the
// material texture sets (fspt_device.hpp), the environment's importance bins and the batch's randBase values - are
// staged in LDS once per block, so those gathers go through the LDS pipeline instead of the vector-memory pipeline.
// The shading kernels stage the scene's small tables in LDS: material texture sets (48 B each), importance bins (16 B each)
// and the batch's randBase values, in DYNAMIC shared memory sized for the scene at launch (C2: 6 sets + 86 bins + 20
// ticks = 1.7 KB; rounds 1-3 reserved 28.7 KB for 256 sets / 1 024 bins whatever the scene, which capped the primary
// launch at 2 blocks per CU by LDS alone).  Scenes whose tables exceed WF_LDS_TABLE_MAX read them from memory.
#ifndef WF_LDS_TABLE_MAX
#define WF_LDS_TABLE_MAX (32u * 1024u)
#endif
static __host__ __device__ inline uint32_t wf_table_bytes(uint32_t n_tex_sets, uint32_t n_bins, uint32_t n_batch) {
  return n_tex_sets * 48u + n_bins * 16u + ((n_batch * 4u + 15u) & ~15u);
}
struct LdsTables { uint4 *sets; uint4 *bins; float *rb; };
// carve the tables out of dynamic LDS at `base` (16-byte aligned) and fill them; the caller synchronises
FM_DEV LdsTables stage_tables(void *base, const DScene &S, const float *rb_trace, uint32_t n_batch, uint32_t nthreads) {
  LdsTables t;
  t.sets = reinterpret_cast<uint4 *>(base);
  t.bins = t.sets + S.n_tex_sets * 3u;
  t.rb = reinterpret_cast<float *>(t.bins + S.n_bins);
  for (uint32_t i = threadIdx.x; i < S.n_tex_sets * 3u; i += nthreads) t.sets[i] = S.tex_sets[i];
  for (uint32_t i = threadIdx.x; i < S.n_bins; i += nthreads) t.bins[i] = S.bins[i];
  for (uint32_t i = threadIdx.x; i < n_batch; i += nthreads) t.rb[i] = rb_trace[i];
  return t;
}
#ifndef WF_LOGIC_WAVES
#define WF_LOGIC_WAVES 4 // 4 waves/SIMD (<= 128 VGPRs): measured best of {3, 4, 5(39 spills)} (profiles/r01)
#endif

template <bool COUNT>
FM_DEV void flush_counters(const Counters &cnt, unsigned long long *counters, int first, int lane) {
  if (!COUNT) return;
  unsigned long long v[6] = {cnt.samples, cnt.rays, cnt.steps, cnt.leaves, cnt.shades, cnt.envs};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (i < first) continue;
    unsigned long long x = v[i];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
    if (lane == 0 && x) atomicAdd(counters + i, x);
  }
}

// ---- primary: round 1 of a batch ---------------------------------------------------------------
// camera.fs (or the injected ray buffers), the camera ray's intersectScene (in place, stack in LDS, no refill: the 64
// lanes of a wave are 64 ticks of one pixel, so their traversals have similar lengths) and its shading, for every
// slot of the batch: the primary ray never travels through HBM, thr / colour / flags are constants, and the VALU-bound
// shading of some waves overlaps the memory-bound traversal of others on the same SIMD (measured +5.6 % over
// separate gen / trace / logic launches, profiles/r01).  Survivors go to consecutive indices of state set 1.
//
// The survivors' indices are reserved per BLOCK iteration with ONE device-scope atomic, before the shading (same-line
// atomics serialise at the memory side, ~15 ns each: 128-thread blocks already cost 40 %).  256-thread blocks: the four
// waves of a SIMD then come from four different blocks and sit in different phases - traversal (memory latency) and
// shading (VALU) overlap - where the two waves a 512-thread block puts on a SIMD run in step: primary 0.135 -> 0.126 ms
// per tick at 20-tick batches, 0.240 -> 0.207 on the 1 M-triangle scene (profiles/r04/ab_primary_block_ticket_uniform*.log).
// Measured with it and NOT adopted: a barrier-free reservation (waves take an LDS ticket, the last arriver does the
// global atomic, a wave only waits for the published base when it stores its survivors: +-0 at 256 threads - it is the
// phase mix, not the barrier); a wave-uniform fast path that fetches the node once through the scalar cache
// (s_load_dwordx16) when all descending lanes stand on the same node (+-0); 96 registers / 5 waves per SIMD (-11 %);
// tables in dynamic LDS alone (no change: the registers hold the kernel at 4 waves per SIMD).
#ifndef WF_PRIMARY_THREADS
#define WF_PRIMARY_THREADS 256
#endif
// Two forms of the traversal phase, chosen per launch (WfP::primary_r; same samples, same values):
//   R = 1  one traversal per lane (trace_rays): the wave waits for the longest of its 64 rays;
//   R = 2  per-lane refill over the wave's 2 x 64 samples (below).
// Which one is faster depends on the scene and the batch: on the 70 k-triangle scene a wave's rays (ticks of three
// neighbouring pixels) have similar lengths and the plain loop's tighter code wins (primary 0.128 vs 0.134 ms per tick at
// 20-tick batches, 0.28 vs 0.35 for a single tick); on the 1 M-triangle scene - sub-pixel triangles, ray lengths all over
// the place - the refill wins (0.215 -> 0.180).  The host measures both on the target's own batches and keeps the faster
// (fspt_api.cpp: primary-form tuner; profiles/r04/ab_primary_refill*.log).
#define WF_PRIMARY_R_MAX 2
#ifndef WF_PRIMARY_SLICE
#define WF_PRIMARY_SLICE 8u // traversal steps between two looks at the wave's sample counter
#endif
template <bool COUNT, bool LDSTAB, int R, bool WIDE = false>
__global__ __launch_bounds__(WF_PRIMARY_THREADS, WF_LOGIC_WAVES) void k_wf_primary(const WfP p) {
  extern __shared__ int lds_dyn[]; // the waves' traversal stacks | [tables] | [camera rays and hits of the block iteration]
  constexpr int NW = WF_PRIMARY_THREADS / WAVE;
  constexpr uint32_t SPAN = (uint32_t)R * WF_PRIMARY_THREADS; // samples of a block iteration
  static_assert(R >= 1 && R <= WF_PRIMARY_R_MAX, "k_wf_primary: R");
  __shared__ uint32_t s_off[2], s_base[2], s_next[2];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = wave_index();
  DScene S = p.scene;
  const float *s_rb = p.rb_trace;
  const WfSet out = p.set[p.set_out];
  WfCounts *cn = p.counts + p.cnt_out;
  // the samples of this launch: the whole batch, or (stream) the units plan(i) took from the cursor.  g = first + i is
  // the sample's number in the run (pixel-major: work index g / n_batch, tick g % n_batch), ring0 + i (mod ring_slots)
  // its place in the fin ring = the slot id the path carries.
  const uint32_t unit_slots = 64u * p.n_batch;
  uint32_t first = 0, n_in = p.n_batch * p.work_total, ring0 = 0;
  if (p.ctl) {
    const uint32_t u0 = p.ctl->plan_start[p.round % WF_RING], nu = p.ctl->plan_units[p.round % WF_RING];
    if (nu == 0u) return; // nothing to generate in this iteration (cursor exhausted, or no room in the pool)
    first = u0 * unit_slots;
    n_in = nu * unit_slots;
    ring0 = (uint32_t)(((unsigned long long)u0 * unit_slots) % p.ring_slots);
  }
  if (threadIdx.x < 2) s_off[threadIdx.x] = 0u;
  int *lds_tab = lds_dyn + (size_t)NW * S.stack_n * WAVE;
  if (LDSTAB) { // behind the stacks
    const LdsTables tb = stage_tables(lds_tab, p.scene, p.rb_trace, p.n_batch, WF_PRIMARY_THREADS);
    S.tex_sets = tb.sets;
    S.bins = tb.bins;
    s_rb = tb.rb;
  }
  __syncthreads();
  Counters cnt = {0, 0, 0, 0, 0, 0};
  int *stack = lds_dyn + (size_t)wave * S.stack_n * WAVE + lane;
  // R > 1: camera ray + hit of every sample of the block iteration: 8 floats each (o.xyz, t | d.xyz, hit), in a piece of
  // LDS per wave that only this wave touches
  float4 *s_ray = reinterpret_cast<float4 *>(lds_tab + (LDSTAB ? wf_table_bytes(S.n_tex_sets, S.n_bins, p.n_batch) / 4u : 0u));

  uint32_t par = 0;
  // Block iterations are handed out by TICKET (the counter is the second line of pool-head slot 1 of this round /
  // iteration - the trace launch uses the first word of each slot -, cleared with the heads: by the resolve launch behind
  // every batch, by k_wf_plan one iteration ahead in a stream run): a block's first iteration is its own index, every further one comes from the
  // counter - blocks whose pixels were cheap take more of them, and the launch ends within one iteration's time of its
  // last block instead of with the slowest block's whole static share (the grid is twice what is resident: with static
  // shares the second half only started when blocks of the first had finished theirs).  70 k triangles: 0.126 -> 0.113
  // ms per tick, 1 M: 0.180 -> 0.145 (profiles/r05/ab_primary_tickets_*.log).  (The first iteration by ticket as well:
  // no different at 20 ticks, single-tick primary 0.23 -> 0.27, ab_first_ticket_*.log.)
  const bool tickets = WF_PRIMARY_TICKETS != 0;
  uint32_t *ticket = p.heads + ((size_t)p.cnt_out * WF_HEADS + 1u) * WF_HEAD_STRIDE + WF_HEAD_STRIDE / 2;
  uint32_t it = blockIdx.x, it_next = 0;
  for (uint32_t base; (unsigned long long)it * SPAN < (unsigned long long)n_in; it = it_next, par ^= 1u) {
    base = it * SPAN;
    it_next = it + gridDim.x;
    unsigned long long m_surv[R];
    // R == 1: this thread's one sample, in registers
    V3 o1 = v3(0.0f, 0.0f, 0.0f), d1 = v3(0.0f, 0.0f, 1.0f);
    float t1 = MAX_T;
    int hit1 = -1;
    bool valid1 = false;
    if constexpr (R == 1) {
      const uint32_t i = base + threadIdx.x;
      const uint32_t g = first + i;
      uint32_t fx = 0, fy = 0;
      valid1 = i < n_in && work_to_pixel(p, wf_work_index(p, g), fx, fy);
      if (valid1) {
        if (p.gen_rays) {
          camera_ray(fx, fy, p.W, p.H, p.cam, p.rb_cam[g % p.n_batch], o1, d1);
        } else {
          float4 po = p.ray_pos[fy * p.W + fx], di = p.ray_dir[fy * p.W + fx];
          o1 = v3(po.x, po.y, po.z);
          d1 = v3(di.x, di.y, di.z);
        }
        if (COUNT) cnt.samples++;
        int hitA;
        trace_rays<COUNT, false, WIDE>(S, stack, o1, false, d1, d1, hitA, t1, hit1, cnt);
      }
    } else {
      // ---- T: the wave's R x 64 samples of this block iteration, with per-lane refill ---------------------------------
      // A lane whose ray is done writes t and hit to LDS and takes the wave's next sample, so the wave is through when
      // the WORK is through, not when its longest ray is.  The wave then shades the same R x 64 samples (lane l: samples
      // l, l + 64, ... of the wave's range, whoever traced them): ray and hit wait in a piece of LDS only this wave
      // touches - no barrier between the two phases, and a sample's ray is read right before it is shaded.  Slot ids
      // and every value are those of the plain form; only the ORDER of the survivors in the state set differs.
      const uint32_t w_lo = base + (uint32_t)wave * (R * WAVE), w_hi = min(w_lo + (uint32_t)(R * WAVE), n_in);
      // the wave's camera rays first, all lanes at work (camera.fs main: 150 instructions a lane should not run alone)
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t loc = (uint32_t)wave * (R * WAVE) + (uint32_t)u * WAVE + (uint32_t)lane;
        const uint32_t i = base + loc, g = first + i;
        uint32_t fx = 0, fy = 0;
        V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
        int h = -2; // -2: the sample does not exist (outside the viewport / beyond the launch)
        if (i < n_in && work_to_pixel(p, wf_work_index(p, g), fx, fy)) {
          if (p.gen_rays) {
            camera_ray(fx, fy, p.W, p.H, p.cam, p.rb_cam[g % p.n_batch], o, d);
          } else {
            const float4 po = p.ray_pos[fy * p.W + fx], di = p.ray_dir[fy * p.W + fx];
            o = v3(po.x, po.y, po.z);
            d = v3(di.x, di.y, di.z);
          }
          h = -1;
          if (COUNT) { cnt.samples++; cnt.rays++; }
        }
        s_ray[2u * loc] = make_float4(o.x, o.y, o.z, MAX_T);
        s_ray[2u * loc + 1u] = make_float4(d.x, d.y, d.z, __int_as_float(h));
      }
      uint32_t next = min(w_lo, w_hi); // (wave-uniform)
      bool have = false;
      uint32_t my_loc = 0;
      V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f), inv = v3(0.0f, 0.0f, 0.0f);
      int cur = REF_SENTINEL, sp = 0, hit = -1;
      float t = MAX_T;
      while (true) {
        // finished rays -> LDS (t and hit: the ray is there already)
        if (have && cur == REF_SENTINEL) {
          s_ray[2u * my_loc].w = t;
          s_ray[2u * my_loc + 1u].w = __int_as_float(hit);
          have = false;
        }
        // refill: a free lane takes the wave's next sample
        const unsigned long long need = __ballot(!have);
        const uint32_t avail = w_hi - next;
        if (avail != 0u && need != 0ull) {
          const uint32_t rank = lane_rank(need), take = min((uint32_t)__popcll(need), avail);
          if (!have && rank < take) {
            my_loc = next + rank - base;
            const float4 ra = s_ray[2u * my_loc], rb = s_ray[2u * my_loc + 1u];
            if (__float_as_int(rb.w) != -2) { // (a sample that does not exist stays as it is)
              o = v3(ra.x, ra.y, ra.z); d = v3(rb.x, rb.y, rb.z);
              inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
              have = true;
              t = MAX_T; hit = -1; cur = S.root_ref; sp = 0;
            }
          }
          next += take;
        }
        if (__ballot(have) == 0ull) { if (next >= w_hi) break; else continue; }
        uint32_t used = 0;
        trace_slice<COUNT, WIDE ? 1 : 0>(S, stack, o, d, inv, false, cur, sp, t, hit, WF_PRIMARY_SLICE, used, cnt);
      }
    }
    // advance_path: a hit is shaded (and the path lives on) unless the bounce budget is already used up
    uint32_t n_mine = 0;
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if constexpr (R == 1) {
        m_surv[u] = __ballot(valid1 && hit1 != -1 && p.num_bounces > 0u);
      } else {
        const uint32_t loc = (uint32_t)wave * (R * WAVE) + (uint32_t)u * WAVE + (uint32_t)lane;
        const int h = base + loc < n_in ? __float_as_int(s_ray[2u * loc + 1u].w) : -2; // -2: no such sample
        m_surv[u] = __ballot(h >= 0 && p.num_bounces > 0u);
      }
      n_mine += (uint32_t)__popcll(m_surv[u]);
    }
    // block-aggregated reservation: the waves' offsets from an LDS counter, ONE global atomic per block iteration.
    // The two LDS words alternate between iterations (parity), so one barrier pair per iteration is enough.
    uint32_t my_off = 0;
    if (lane == 0) my_off = atomicAdd(&s_off[par], n_mine);
    my_off = __builtin_amdgcn_readfirstlane(my_off);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = s_off[par];
      s_base[par] = tot ? atomicAdd(&cn->n_ext, tot) : 0u;
      s_off[par ^ 1u] = 0u;
      if (tickets) s_next[par] = gridDim.x + atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (tickets) it_next = __builtin_amdgcn_readfirstlane(s_next[par]);
#pragma unroll
    for (int u = 0; u < R; ++u) {
      uint32_t i;
      bool valid;
      V3 ro, rd;
      float tB;
      int hitB;
      if constexpr (R == 1) {
        i = base + threadIdx.x;
        valid = valid1; ro = o1; rd = d1; tB = t1; hitB = hit1;
      } else {
        const uint32_t loc = (uint32_t)wave * (R * WAVE) + (uint32_t)u * WAVE + (uint32_t)lane;
        i = base + loc;
        const float4 ra = s_ray[2u * loc], rb = s_ray[2u * loc + 1u];
        valid = i < n_in && __float_as_int(rb.w) != -2;
        ro = v3(ra.x, ra.y, ra.z); rd = v3(rb.x, rb.y, rb.z);
        tB = ra.w;
        hitB = __float_as_int(rb.w);
      }
      if (valid) {
        uint32_t slot = ring0 + i; // < 2 * ring_slots: a launch is shorter than the ring
        if (slot >= p.ring_slots) slot -= p.ring_slots;
        Path ps;
        ps.ro = ro; ps.rd = rd;
        ps.thr = v3(1.0f, 1.0f, 1.0f);
        ps.color = v3(0.0f, 0.0f, 0.0f);
        ps.envDir = v3(0.0f, 0.0f, 0.0f);
        ps.pend = v3(0.0f, 0.0f, 0.0f);
        ps.wx = ps.wy = 0.0f;
        ps.bounce = 0; ps.iters = 0; ps.pix = 0; ps.lag = 0u;
        ps.hasShadow = false; ps.primary = true;
        const uint32_t j = (first + i) % p.n_batch;
        const bool finished = advance_path<COUNT>(S, ps, -1, tB, hitB, s_rb[j], p.env_theta, p.num_bounces, cnt);
        if (finished) st3(p.fin + 3 * (size_t)slot, ps.color);
        else store_path(out, s_base[par] + my_off + lane_rank(m_surv[u]), ps, slot);
      }
      my_off += (uint32_t)__popcll(m_surv[u]);
    }
  }
  flush_counters<COUNT>(cnt, p.counters, 0, lane);
}

// ---- logic: one S step for every live path of the round (rounds >= 2) ---------------------------
// Input: the n paths of state set (round-1)&1 with the results of their rays.  Per block iteration (U*512 paths):
//   1  classify from ONE word per path, the hit index (the trace kernel stores WF_HIT_TERMINAL for a hit whose path has
//      no bounce left): a path is shaded - and survives - iff its extension ray hit something and its bounce budget is
//      not used up (tracer.fs:446,509); everything else finishes here.  The shaded
//      ones are listed in LDS, and ONE atomic reserves their consecutive output indices BEFORE any shading;
//   2a every thread finishes its own non-shaded paths (NEE result, environment on a miss -> fin[slot]);
//   2b the listed paths are shaded by consecutive threads - whole waves of shading work, instead of the 1-in-5 lanes
//      a round-2 wave has when every thread keeps its own path (VALU lane utilisation 0.34 in round 1's profile).
template <bool COUNT, bool LDSTAB>
__global__ __launch_bounds__(WF_LOGIC_THREADS, WF_LOGIC_WAVES) void k_wf_logic(const WfP p) {
  constexpr int U = WF_LOGIC_U;
  extern __shared__ int lds_dyn[]; // the staged tables
  __shared__ uint16_t s_list[U * WF_LOGIC_THREADS];
  __shared__ uint32_t s_n, s_total, s_gbase, s_next;
  // the launch's last carry_blocks blocks move the suspended traversals' paths on (k_wf_carry's work) and are done
  const uint32_t n_blocks = gridDim.x - p.carry_blocks;
  if (blockIdx.x >= n_blocks) {
    carry_records(p, (blockIdx.x - n_blocks) * WF_LOGIC_THREADS + threadIdx.x, p.carry_blocks * WF_LOGIC_THREADS);
    return;
  }
  const int lane = threadIdx.x & (WAVE - 1);
  DScene S = p.scene;
  const float *s_rb = p.rb_trace;
  if (threadIdx.x == 0) s_n = 0;
  if (LDSTAB) {
    const LdsTables tb = stage_tables(lds_dyn, p.scene, p.rb_trace, p.n_batch, WF_LOGIC_THREADS);
    S.tex_sets = tb.sets;
    S.bins = tb.bins;
    s_rb = tb.rb;
  }
  __syncthreads();
  const WfSet in = p.set[p.set_in], out = p.set[p.set_out];
  const uint32_t n_in = p.counts[p.cnt_in].n_ext;
  WfCounts *cn = p.counts + p.cnt_out;
  Counters cnt = {0, 0, 0, 0, 0, 0};

  // paths per thread and iteration: as many as keep every block busy, at most U
  uint32_t u_eff = (n_in + n_blocks * WF_LOGIC_THREADS - 1) / (n_blocks * WF_LOGIC_THREADS);
  u_eff = u_eff < 1u ? 1u : (u_eff > (uint32_t)U ? (uint32_t)U : u_eff);
  const uint32_t span = u_eff * WF_LOGIC_THREADS;
  // block iterations by ticket, as in k_wf_primary: the counter is the second line of this round's (iteration's) first
  // pool-head slot, cleared with the heads
  const bool tickets = WF_LOGIC_TICKETS != 0;
  uint32_t *ticket = p.heads + (size_t)p.cnt_out * WF_HEADS * WF_HEAD_STRIDE + WF_HEAD_STRIDE / 2;
  uint32_t it_next = 0;
  for (uint32_t it = blockIdx.x; (unsigned long long)it * span < (unsigned long long)n_in; it = it_next) {
    const uint32_t base = it * span;
    it_next = it + n_blocks;
    // ---- 1: classify ----
    uint32_t own_fin = 0; // bit u: own path u finishes in this round (handled in 2a)
    for (uint32_t u = 0; u < u_eff; ++u) {
      const uint32_t loc = u * WF_LOGIC_THREADS + threadIdx.x;
      const uint32_t i = base + loc;
      bool shade = false;
      if (i < n_in) {
        const int idx = __float_as_int(ld2(p.hit + i).y);
        shade = idx >= 0; // -1: miss; WF_HIT_TERMINAL: hit with the bounce budget used up
        // (WF_HIT_PENDING: its traversal was suspended - k_wf_carry has moved the path on, nothing to do here)
        if (!shade && idx != WF_HIT_PENDING) own_fin |= 1u << u;
      }
      const unsigned long long m = __ballot(shade);
      uint32_t wb = 0;
      if (lane == 0 && m) wb = atomicAdd(&s_n, (uint32_t)__popcll(m));
      wb = __builtin_amdgcn_readfirstlane(wb);
      if (shade) s_list[wb + lane_rank(m)] = (uint16_t)loc;

    }
    __syncthreads();
    uint32_t my_gbase = 0;
    if (threadIdx.x == 0) {
      const uint32_t tot = s_n;
      s_total = tot;
      s_n = 0;
      my_gbase = tot ? atomicAdd(&cn->n_ext, tot) : 0u; // consumed after phase 2a, which covers its round trip
      if (tickets) s_next = n_blocks + atomicAdd(ticket, 1u);
    }
    // ---- 2a (own finishing paths) then 2b (listed paths, dense) through ONE copy of the path code ----
    uint32_t n_shade = 0, gbase = 0;
    for (uint32_t jt = 0;; ++jt) {
      uint32_t i = 0, k_out = 0;
      bool active = false;
      if (jt < u_eff) {
        i = base + jt * WF_LOGIC_THREADS + threadIdx.x;
        active = (own_fin >> jt) & 1u;
      } else {
        if (jt == u_eff) {
          if (threadIdx.x == 0) s_gbase = my_gbase;
          __syncthreads();
          n_shade = s_total;
          gbase = s_gbase;
          if (tickets) it_next = __builtin_amdgcn_readfirstlane(s_next);
        }
        const uint32_t tsh = (jt - u_eff) * WF_LOGIC_THREADS + threadIdx.x;
        if ((jt - u_eff) * WF_LOGIC_THREADS >= n_shade) break; // block-uniform
        active = tsh < n_shade;
        if (active) { i = base + s_list[tsh]; k_out = gbase + tsh; }
      }
      if (active) {
        Path ps;
        int hitA;
        const uint32_t slot = load_path(in, i, ps, p.shadow_hit, hitA);
        const float2 h = ld2(p.hit + i);
        const uint32_t j = slot % p.n_batch;
        const bool finished = advance_path<COUNT>(S, ps, hitA, h.x, __float_as_int(h.y), s_rb[j], p.env_theta, p.num_bounces, cnt);
        if (finished) st3(p.fin + 3 * (size_t)slot, ps.color);
        else store_path(out, k_out, ps, slot);
      }
    }
    __syncthreads(); // s_list / s_total are rewritten by the next iteration
  }
  flush_counters<COUNT>(cnt, p.counters, 4, lane);
}

FM_DEV V3 shfl3(V3 v, int src) { return v3(__shfl(v.x, src, WAVE), __shfl(v.y, src, WAVE), __shfl(v.z, src, WAVE)); }

// ---- tail: the live paths of a late round run to completion in ONE kernel ------------------------
// After a few rounds only a few thousand paths are alive, and every further round costs a trace launch as long as its
// longest ray plus a logic launch plus two launch gaps (~220 us per round at 1080p, whatever the path count).  This
// kernel takes the survivors of round p.round and alternates T (trace) and S (advance_path) per path until it ends,
// refilling finished lanes from the list: the rounds are no longer synchronised.  A path lives on a PAIR of lanes: the
// even lane owns the state and traces the extension ray, the odd lane traces the path's NEE shadow ray at the same
// time (handed over and back with lane shuffles) - what such a launch costs is the longest chain of dependent rays,
// and the shadow rays are off that chain this way.  It also ends paths that refraction keeps alive beyond NUM_BOUNCES
// rounds (tracer.fs:488) without any host round trip.
// 4 waves/SIMD (<= 128 VGPRs; 133 unbounded = 3): the kernel is a bundle of dependent chains, more of them in flight
// is all that speeds it up - K=20 +1.3 %, 5 waves (spills) no better (profiles/r02/ab_ref8_tail_waves.log)
#ifndef WF_TAIL_WAVES
#define WF_TAIL_WAVES 4
#endif
#ifndef WF_TAIL_PAIRS
#define WF_TAIL_PAIRS 32u // lane pairs of a wave that carry a path when WF_TAIL_PAIRS_AUTO is 0 (measurement hook: profiles/r05/ab_tail_pairs_*.log)
#endif
#ifndef WF_TAIL_PAIRS_AUTO
#define WF_TAIL_PAIRS_AUTO 1 // the kernel spreads a small launch's paths over all its waves (see k_wf_tail)
#endif
#ifndef WF_TAIL_SLICE
#define WF_TAIL_SLICE 32u // traversal steps of a lane per T phase.  No slicing / 64 / 32 / 16: tail 0.126 / 0.115 / 0.100 / 0.092 ms per
// tick on the 1 M-triangle scene at 20-tick batches, 0.755 / 0.762 / 0.760 / 0.803 ms for a single tick of the 70 k scene,
// 0.032 everywhere at 20 ticks of it (profiles/r04/ab_tail_slice*.log).  One lane per path (both rays one after the other,
// shading waves twice as dense) instead of the pair: 0.79 -> 1.09 ms single tick (ab_tail_lane*.log): once the list is
// used up the kernel is as long as its longest path CHAIN, and the pair takes the shadow rays off that chain.
#endif
// GEN (stream scheduler, the last launch of a run): when the list is used up the wave also takes whole UNITS the cursor
// has not handed out (the host only estimated how many iterations the run needs) and runs their samples itself.  A
// pair then owns a PIXEL: it generates and traces the pixel's ticks one after the other and folds every finished
// sample into the pixel's accumulator value in its registers, in tick order (tracer.fs:515-517 is order-dependent) -
// nothing of such a unit goes through the fin ring, and the run's last resolve ends where this launch began
// (WfStreamCtl::hist of the last plan).
// WIDE (node form of the traversal, trace_slice): 0 the 64-byte nodes, 1 the two-level nodes, 2 ADAPTIVE - the 64-byte
// nodes while the wave can still refill its pairs from the list (the CU's vector-memory front end is what its resident waves
// share: fewer requests per step win), the two-level nodes once the list is used up: what remains of the launch is the
// dependent chains of the paths still alive, walked by a few lanes on a mostly idle chip - there a step costs a cache-miss
// latency, and two levels per round trip shorten the chain (profiles/r05/launch_list_*.txt: the tail launch is 0.8 ms of a
// 9.9 ms batch on the 70 k-triangle scene, 2.4 of 13.3 ms on the 1 M-triangle one, most of it this end phase).
template <bool COUNT, bool ANYHIT, bool GEN, int WIDE = 0>
__global__ __launch_bounds__(BLOCK_THREADS, WF_TAIL_WAVES) void k_wf_tail(const WfP p) {
  extern __shared__ int lds_stack[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = wave_index();
  const DScene &S = p.scene;
  int *stack = lds_stack + (size_t)wave * S.stack_n * WAVE + lane;
  const WfSet in = p.set[p.set_out];
  WfCounts *cn = p.counts + p.cnt_out;
  const uint32_t total = cn->n_ext;
  Counters cnt = {0, 0, 0, 0, 0, 0};
  // Lane pairs of a wave that take paths.  A wave's traversal step lasts as long as the slowest of its lanes' fetches (and
  // its shading phase as long as its paths' branches differ), so a launch with FEW paths spreads them over all its
  // waves - as few pairs per wave as that takes - instead of filling some waves to the brim: 20 K paths 0.031 -> 0.026 ms
  // per tick, 5 K paths 0.55 -> 0.36 ms (profiles/r05/ab_tail_pairs_*.log); with many paths (> 32 per wave) nothing changes.
  const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
  uint32_t PAIRS = WF_TAIL_PAIRS;
  if (WF_TAIL_PAIRS_AUTO && !GEN) {
    PAIRS = (total + n_waves - 1u) / n_waves;
    PAIRS = PAIRS < 1u ? 1u : (PAIRS > WAVE / 2 ? WAVE / 2 : PAIRS);
  }
  const bool is_main = (lane & 1) == 0 && (uint32_t)(lane >> 1) < PAIRS;
  Path ps;
  ps.pix = -1;
  ps.hasShadow = false;
  ps.lag = 0u;
  ps.ro = ps.rd = ps.envDir = v3(0.0f, 0.0f, 1.0f);
  uint32_t slot = 0;
  // this lane's ray (even lanes: the path's extension ray, odd lanes: its NEE shadow ray): RAY_NONE, RAY_GOING (its
  // traversal state below is valid; the stack is the lane's LDS column) or RAY_DONE (result in r_t / r_hit)
  enum { RAY_NONE = 0, RAY_GOING = 1, RAY_DONE = 2 };
  int r_state = RAY_NONE, r_cur = REF_SENTINEL, r_sp = 0, r_hit = -1;
  float r_t = MAX_T;
  // pool as in k_wf_trace: chunks of 32 paths; every wave's first chunk is its own, the rest is dealt out by the
  // wave's stripe head
  const uint32_t wave_id = blockIdx.x * WAVES_PER_BLOCK + wave;
  const uint32_t stripe = wave_id % WF_HEADS;
  const uint32_t n_chunks = (total + PAIRS - 1u) / PAIRS;
  uint32_t pool_next = min(wave_id * PAIRS, total), pool_end = min(wave_id * PAIRS + PAIRS, total);
  bool exhausted = n_chunks <= n_waves;
  // GEN: the pair's own pixel (work index; -1 none), the tick its current path is, the pixel's accumulator value so far
  int g_w = -1;
  uint32_t g_j = 0, g_pix = 0;
  bool g_path = false; // the pair's current path is one it generated (its result goes to g_acc, not to fin)
  float4 g_acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  uint32_t gen_next = 0, gen_end = 0; // work indices [gen_next, gen_end) of the wave's current unit not yet given to a pair
  bool gen_done = !GEN;
  while (true) {
    // ---- refill idle pairs with paths of the list ----
    while (true) {
      unsigned long long need = __ballot(is_main && ps.pix < 0 && g_w < 0);
      if (need == 0ull) break;
      uint32_t avail = pool_end - pool_next;
      if (avail == 0u) {
        if (exhausted) break;
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&p.heads[((size_t)p.cnt_out * WF_HEADS + stripe) * WF_HEAD_STRIDE], 1u);
        b = __builtin_amdgcn_readfirstlane(b);
        const unsigned long long c = (unsigned long long)n_waves + (unsigned long long)b * WF_HEADS + stripe;
        if (c >= n_chunks) { exhausted = true; break; }
        pool_next = (uint32_t)c * PAIRS;
        pool_end = min(pool_next + PAIRS, total);
        continue;
      }
      uint32_t rank = lane_rank(need);
      uint32_t want = (uint32_t)__popcll(need);
      uint32_t take = want < avail ? want : avail;
      if (is_main && ps.pix < 0 && g_w < 0 && rank < take) {
        int unused;
        slot = load_path(in, pool_next + rank, ps, nullptr, unused);
        ps.pix = 0;
      }
      pool_next += take;
    }
    if (GEN && exhausted && pool_end == pool_next) {
      // ---- the list is used up: idle pairs without a pixel take one from the wave's unit (units from the cursor) ----
      while (!gen_done) {
        unsigned long long need = __ballot(is_main && ps.pix < 0 && g_w < 0);
        if (need == 0ull) break;
        if (gen_next == gen_end) {
          uint32_t u = 0;
          if (lane == 0) u = atomicAdd(&p.ctl->cursor, 1u);
          u = __builtin_amdgcn_readfirstlane(u);
          if (u >= (p.work_total >> 6)) { gen_done = true; break; }
          if (lane == 0) atomicAdd(&p.ctl->fin_gen_units, 1u);
          gen_next = u * 64u;
          gen_end = gen_next + 64u;
        }
        const uint32_t left = gen_end - gen_next;
        const uint32_t rank = lane_rank(need), want = (uint32_t)__popcll(need);
        const uint32_t take = want < left ? want : left;
        if (is_main && ps.pix < 0 && g_w < 0 && rank < take) {
          uint32_t fx = 0, fy = 0;
          const uint32_t w = gen_next + rank;
          if (work_to_pixel(p, w, fx, fy)) { // (a pixel outside the viewport does not exist)
            g_w = (int)w;
            g_j = 0;
            g_pix = fy * p.W + fx;
            g_acc = p.accum[g_pix];
            g_path = false;
          }
        }
        gen_next += take;
      }
      // ---- pairs that own a pixel and have no path: the pixel's next tick ----
      if (is_main && ps.pix < 0 && g_w >= 0) {
        const uint32_t fx = g_pix % p.W, fy = g_pix / p.W;
        if (p.gen_rays) {
          camera_ray(fx, fy, p.W, p.H, p.cam, p.rb_cam[g_j], ps.ro, ps.rd);
        } else {
          const float4 po = p.ray_pos[g_pix], di = p.ray_dir[g_pix];
          ps.ro = v3(po.x, po.y, po.z);
          ps.rd = v3(di.x, di.y, di.z);
        }
        ps.thr = v3(1.0f, 1.0f, 1.0f);
        ps.color = v3(0.0f, 0.0f, 0.0f);
        ps.envDir = v3(0.0f, 0.0f, 1.0f);
        ps.pend = v3(0.0f, 0.0f, 0.0f);
        ps.wx = ps.wy = 0.0f;
        ps.bounce = 0; ps.iters = 0;
        ps.hasShadow = false; ps.primary = true;
        ps.pix = 0;
        slot = g_j; // (only its tick is used: slot % n_batch)
        g_path = true;
        if (COUNT) cnt.samples++;
      }
    }
    if (__ballot(is_main && ps.pix >= 0) == 0ull) break;
    // ---- T: even lanes trace their path's extension ray, odd lanes the same path's shadow ray - in slices of
    // WF_TAIL_SLICE steps.  A wave used to sit in this phase until its LONGEST ray was done (a few hundred dependent node
    // fetches walked by one lane, ten times the average ray) while 63 lanes waited: VALU lane utilisation 0.09.  Now a
    // ray that is not finished when the slice ends simply keeps its traversal state and goes on in the next T phase; the
    // pairs whose rays ARE finished are shaded and get their next rays (or a new path) in between.
    const int src = lane & ~1;
    const bool m_live = ps.pix >= 0; // (false on odd lanes)
    const bool pair_live = __shfl((int)m_live, src, WAVE) != 0;
    const bool pair_shadow = __shfl((int)(m_live && ps.hasShadow), src, WAVE) != 0;
    const V3 o = shfl3(ps.ro, src);
    const V3 d_sh = shfl3(ps.envDir, src);
    const V3 d = is_main ? ps.rd : d_sh;
    if (r_state == RAY_NONE && (is_main ? pair_live : pair_shadow)) { // a new ray starts at the root
      r_state = RAY_GOING; r_cur = S.root_ref; r_sp = 0; r_t = MAX_T; r_hit = -1;
      if (COUNT) cnt.rays++;
    }
    if (r_state != RAY_GOING) r_cur = REF_SENTINEL;
    uint32_t used = 0;
    trace_slice<COUNT, WIDE>(S, stack, o, d, v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), ANYHIT && !is_main, r_cur, r_sp, r_t, r_hit, p.tail_slice ? p.tail_slice : WF_TAIL_SLICE,
                             used, cnt, exhausted && pool_next == pool_end && gen_done);
    if (r_state == RAY_GOING && r_cur == REF_SENTINEL) r_state = RAY_DONE;
    // a pair is ready when its extension ray is done and its shadow ray is done or was never cast
    const int st_other = __shfl(r_state, lane | 1, WAVE);
    const bool ready = m_live && r_state == RAY_DONE && st_other != RAY_GOING;
    const int hit_other = __shfl(r_hit, lane | 1, WAVE);
    const int hitA = st_other == RAY_DONE ? hit_other : -1; // the shadow ray's result, back on the even lane
    const bool pair_ready = __shfl((int)ready, src, WAVE) != 0;
    // ---- S: consume them ----
    if (pair_ready) r_state = RAY_NONE; // both lanes of the pair: their rays are consumed now
    const float tR = r_t;
    const int hitR = r_hit;
    if (ready) {
      if (advance_path<COUNT>(S, ps, hitA, tR, hitR, p.rb_trace[slot % p.n_batch], p.env_theta, p.num_bounces, cnt)) {
        if (GEN && g_path) {
          g_acc = accumulate_sample(g_acc, ps.color, p.first_tick + g_j);
          g_path = false;
          if (++g_j == p.n_batch) { p.accum[g_pix] = g_acc; g_w = -1; }
        } else {
          st3(p.fin + 3 * (size_t)slot, ps.color);
        }
        ps.pix = -1;
      }
    }
  }
  flush_counters<COUNT>(cnt, p.counters, GEN ? 0 : 1, lane);
}

// ---- resolve: tracer.fs:515-517 for the batch's ticks in order, per pixel --------------
// fin is slot-major (pixel-major): a wave reads 8 ticks of 64 pixels as full 128-byte segments, transposes them
// through LDS and every lane folds its pixel's ticks in order (the running mean is order-dependent).
#define WF_RESOLVE_TICKS 8
#ifndef WF_RESOLVE_BLOCKS_PER_CU
#define WF_RESOLVE_BLOCKS_PER_CU 16u
#endif
__global__ __launch_bounds__(BLOCK_THREADS) void k_wf_resolve(const WfP p) {
  __shared__ float4 s_t[WAVES_PER_BLOCK][WAVE * (WF_RESOLVE_TICKS + 1)];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  // one wave per UNIT (64 consecutive work indices = one 8x8 pixel patch, all n_batch ticks).  Batch scheduler: every
  // unit of the batch; stream: the units between two recorded positions of the cursor (fspt_device.hpp).
  const uint32_t unit_slots = 64u * p.n_batch;
  uint32_t u0 = 0, u1 = p.work_total >> 6;
  if (p.ctl) {
    u0 = p.res_from < 0 ? 0u : p.ctl->hist[p.res_from % WF_HIST];
    u1 = p.res_to == -2 ? (p.work_total >> 6) : (p.res_to < 0 ? 0u : p.ctl->hist[p.res_to % WF_HIST]);
  }
  if (p.zero_rounds && blockIdx.x == 0) {
    // nobody reads this batch's counters or pool heads any more (every trace / logic / tail launch is done): hand the
    // live-path counts to the host and clear both for the next batch
    for (uint32_t r = threadIdx.x; r < p.zero_rounds; r += BLOCK_THREADS) {
      if (p.live_out) p.live_out[r] = p.counts[r].n_ext;
      p.counts[r].n_ext = 0u;
      p.counts[r].n_susp = 0u;
    }
    for (uint32_t i = threadIdx.x; i < p.zero_rounds * WF_HEADS; i += BLOCK_THREADS) {
      p.heads[(size_t)i * WF_HEAD_STRIDE] = 0u;
      p.heads[(size_t)i * WF_HEAD_STRIDE + WF_HEAD_STRIDE / 2] = 0u; // (slots 0 / 1 of a round: the logic / primary launch's ticket counters)
    }
  }
  for (uint32_t ub = u0 + blockIdx.x * WAVES_PER_BLOCK; ub < u1; ub += gridDim.x * WAVES_PER_BLOCK) { // block-uniform trip count
    const uint32_t unit = ub + (uint32_t)wave;
    const bool unit_ok = unit < u1;
    const uint32_t w = wf_work_index(p, unit * unit_slots + (uint32_t)lane * p.n_batch);
    const size_t fin0 = (size_t)(((unsigned long long)unit * unit_slots) % p.ring_slots); // the unit's place in the fin ring
    uint32_t x = 0, y = 0;
    const bool ok = unit_ok && work_to_pixel(p, w, x, y);
    const uint32_t pix = y * p.W + x;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (ok) acc = p.accum[pix];
    for (uint32_t j0 = 0; j0 < p.n_batch; j0 += WF_RESOLVE_TICKS) {
      const uint32_t tt = j0 + ((uint32_t)lane & (WF_RESOLVE_TICKS - 1));
#pragma unroll
      for (int q = 0; q < WF_RESOLVE_TICKS; ++q) {
        const uint32_t pp = (uint32_t)q * (WAVE / WF_RESOLVE_TICKS) + ((uint32_t)lane / WF_RESOLVE_TICKS);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (unit_ok && tt < p.n_batch) { const V3 c = ld3(p.fin + 3 * (fin0 + (size_t)pp * p.n_batch + tt)); v = make_float4(c.x, c.y, c.z, 0.0f); }
        s_t[wave][pp * (WF_RESOLVE_TICKS + 1) + ((uint32_t)lane & (WF_RESOLVE_TICKS - 1))] = v;
      }
      __syncthreads();
      if (ok) {
#pragma unroll
        for (int jj = 0; jj < WF_RESOLVE_TICKS; ++jj) {
          if (j0 + jj < p.n_batch) {
            const float4 c = s_t[wave][lane * (WF_RESOLVE_TICKS + 1) + jj];
            acc = accumulate_sample(acc, v3(c.x, c.y, c.z), p.first_tick + j0 + jj);
          }
        }
      }
      __syncthreads();
    }
    if (ok) p.accum[pix] = acc;
  }
}

// ---- plan (stream scheduler): how many units iteration p.round generates -------------------------
// One thread decides, before primary(i) / logic(i) start: every live path of the previous set may survive logic(i) and
// every new sample may survive its first shading, so  units <= (cap - live(i-1)) / unit size  can never overflow set i.
// Also zeroes the counter and the pool heads of iteration i + 1 (nobody uses them yet: fspt_api.cpp render_stream).
__global__ __launch_bounds__(WAVE) void k_wf_plan(const WfP p) {
  WfStreamCtl *c = p.ctl;
  const uint32_t nxt = (p.cnt_out + 1u) % WF_RING;
  if (threadIdx.x < WF_HEADS) {
    p.heads[((size_t)nxt * WF_HEADS + threadIdx.x) * WF_HEAD_STRIDE] = 0u;
    p.heads[((size_t)nxt * WF_HEADS + threadIdx.x) * WF_HEAD_STRIDE + WF_HEAD_STRIDE / 2] = 0u; // ticket counters of primary(i + 1), logic(i + 1)
  }
  if (threadIdx.x != 0) return;
  p.counts[nxt].n_ext = 0u;
  p.counts[nxt].n_susp = 0u;
  const uint32_t unit_slots = 64u * p.n_batch;
  // paths set i may have to hold besides the new ones: everything alive in set i-1 (logic(i) has yet to run), or -
  // one-stream form, logic(i) is done - the survivors it has already written
  const uint32_t live = p.serial ? p.counts[p.cnt_out].n_ext : (p.round == 0u ? 0u : p.counts[p.cnt_in].n_ext);
  const uint32_t room = p.cap > live ? (p.cap - live) / unit_slots : 0u;
  const uint32_t total_units = p.work_total >> 6;
  const uint32_t cur = c->cursor, left = total_units > cur ? total_units - cur : 0u;
  uint32_t take = p.take_max < room ? p.take_max : room;
  if (take > left) take = left;
  c->plan_start[p.round % WF_RING] = cur;
  c->plan_units[p.round % WF_RING] = take;
  c->cursor = cur + take;
  c->hist[p.round % WF_HIST] = cur + take;
  c->n_iters = p.round + 1u;
  if (take) c->last_gen_it = p.round;
  if (live > c->max_live) c->max_live = live;
  c->sum_live += live;
}

// multi-device read-out (fspt_multi_read_radiance): pack a shard's own pixels into work-index order / scatter them back
template <bool UNPACK>
__global__ __launch_bounds__(BLOCK_THREADS) void k_tile_pack(const TilePackP p) {
  const uint32_t n = p.n_owned_tiles * p.tile * p.tile;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
    uint32_t x, y;
    if (!work_to_pixel(p, w, x, y)) continue;
    if (p.channels == 3u) {
      float *q = reinterpret_cast<float *>(p.packed) + 3 * (size_t)w;
      if (UNPACK) { const V3 c = ld3(q); p.accum[(size_t)y * p.W + x] = make_float4(c.x, c.y, c.z, 1.0f); }
      else { const float4 a = p.accum[(size_t)y * p.W + x]; st3(q, v3(a.x, a.y, a.z)); }
    } else if (UNPACK) p.accum[(size_t)y * p.W + x] = p.packed[w];
    else p.packed[w] = p.accum[(size_t)y * p.W + x];
  }
}

// camera.fs as a stand-alone pass (drawCamera, main.js:741-756)
__global__ __launch_bounds__(BLOCK_THREADS) void k_camera(uint32_t W, uint32_t H, uint32_t vw, uint32_t vh, CameraP cam,
                                                         float randBase, float4 *pos, float4 *dir) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  uint32_t x = i % W, y = i / W;
  if (x >= vw || y >= vh) return; // outside gl.viewport: the ray textures keep their old texels
  V3 o, d;
  camera_ray(x, y, W, H, cam, randBase, o, d);
  pos[i] = make_float4(o.x, o.y, o.z, 1.0f);
  dir[i] = make_float4(d.x, d.y, d.z, 1.0f);
}

// draw.fs (1-93): exposure -> ACES fit -> saturation -> gamma (+ optional 5x5 firefly filter) -> RGBA8
FM_DEV float draw_luma(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }
FM_DEV V3 draw_fetch(const float4 *acc, int W, int H, int x, int y) {
  if (x < 0 || y < 0 || x >= W || y >= H) return v3(0.0f, 0.0f, 0.0f);
  float4 p = acc[(size_t)y * W + x];
  return v3(p.x, p.y, p.z);
}
FM_DEV float rrt_odt(float v) {
  float a = fma_(v, v + 0.0245786f, -0.000090537f);
  float b = fma_(v, fma_(0.983729f, v, 0.4329510f), 0.238081f);
  return a / b;
}
__global__ __launch_bounds__(BLOCK_THREADS) void k_draw(const float4 *acc, uint32_t W, uint32_t H, float exposure,
                                                       float saturation, int denoise, float maxSigma, float scale,
                                                       uint32_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  // ivec2(gl_FragCoord * scale) (draw.fs:59,87): the reference draws with scale 0.25 while the camera moves
  int x = (int)(((float)(i % W) + 0.5f) * scale), y = (int)(((float)(i / W) + 0.5f) * scale);
  V3 c;
  if (denoise) {
    float sum = 0.0f, sq = 0.0f, middleLuma = 0.0f;
    V3 middle = v3(0.0f, 0.0f, 0.0f);
    for (int a = 0; a < 5; ++a)
      for (int b = 0; b < 5; ++b) {
        int ox = a - 2, oy = b - 2;
        V3 col = draw_fetch(acc, (int)W, (int)H, x + ox, y + oy);
        float l = draw_luma(col);
        if (ox == 0 && oy == 0) { middle = col; middleLuma = l; continue; }
        sum += l;
        sq = fma_(l, l, sq);
      }
    float mean = sum / 24.0f;
    float variance = fma_(-mean, mean, sq / 24.0f);
    float sigma = sqrt_(variance);
    if (abs_(middleLuma - mean) > maxSigma * sigma) middle = middle * (mean / middleLuma);
    c = middle * exposure;
  } else {
    c = draw_fetch(acc, (int)W, (int)H, x, y) * exposure;
  }
  V3 a = v3(dot(c, v3(0.59719f, 0.35458f, 0.04823f)), dot(c, v3(0.07600f, 0.90834f, 0.01566f)),
            dot(c, v3(0.02840f, 0.13383f, 0.83777f)));
  a = v3(rrt_odt(a.x), rrt_odt(a.y), rrt_odt(a.z));
  V3 m = v3(dot(a, v3(1.60475f, -0.53108f, -0.07367f)), dot(a, v3(-0.10208f, 1.10813f, -0.00605f)),
            dot(a, v3(-0.00327f, -0.07276f, 1.07602f)));
  m = v3(clamp_(m.x, 0.0f, 1.0f), clamp_(m.y, 0.0f, 1.0f), clamp_(m.z, 0.0f, 1.0f));
  float l = draw_luma(m);
  float os = 1.0f - saturation;
  m = v3(fma_(m.x, saturation, l * os), fma_(m.y, saturation, l * os), fma_(m.z, saturation, l * os));
  float g0 = pow_(m.x, 0.454545f), g1 = pow_(m.y, 0.454545f), g2 = pow_(m.z, 0.454545f);
  uint32_t r8 = (uint32_t)floor_(fma_(clamp_(g0, 0.0f, 1.0f), 255.0f, 0.5f));
  uint32_t g8 = (uint32_t)floor_(fma_(clamp_(g1, 0.0f, 1.0f), 255.0f, 0.5f));
  uint32_t b8 = (uint32_t)floor_(fma_(clamp_(g2, 0.0f, 1.0f), 255.0f, 0.5f));
  out[i] = r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
}

// intersectScene as a stand-alone pass
__global__ __launch_bounds__(BLOCK_THREADS) void k_intersect(const IntersectP p) {
  extern __shared__ int lds_stack[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  int *stack = lds_stack + (size_t)wave * p.scene.stack_n * WAVE + lane;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  V3 o = v3(p.rays[i * 6], p.rays[i * 6 + 1], p.rays[i * 6 + 2]);
  V3 d = v3(p.rays[i * 6 + 3], p.rays[i * 6 + 4], p.rays[i * 6 + 5]);
  Counters cnt = {0, 0, 0, 0, 0, 0};
  int hitA, hitB;
  float tB;
  if (p.wide) trace_rays<true, false, true>(p.scene, stack, o, false, d, d, hitA, tB, hitB, cnt);
  else trace_rays<true, false>(p.scene, stack, o, false, d, d, hitA, tB, hitB, cnt);
  p.t_out[i] = tB;
  p.index_out[i] = slot_to_tri(p.scene, hitB); // the reference's triangle index (tracer.fs:360)
  if (p.steps_out) p.steps_out[i] = cnt.steps;
  if (p.leaves_out) p.leaves_out[i] = cnt.leaves;
}

// bvh_test.fs main (224-232): traversal-step heat map of the camera rays, the reference's `mode=test`
__global__ __launch_bounds__(BLOCK_THREADS) void k_bvh_test(const TraceP p) {
  extern __shared__ int lds_stack[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  int *stack = lds_stack + (size_t)wave * p.scene.stack_n * WAVE + lane;
  const uint32_t work_total = p.n_owned_tiles * p.tile * p.tile;
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x, y;
  if (w >= work_total || !work_to_pixel(p, w, x, y)) return;
  uint32_t pix = y * p.W + x;
  float4 po = p.ray_pos[pix], di = p.ray_dir[pix];
  V3 o = v3(po.x, po.y, po.z), d = v3(di.x, di.y, di.z);
  Counters cnt = {0, 0, 0, 0, 0, 0};
  int hitA, hitB;
  float tB;
  trace_rays<true, false>(p.scene, stack, o, false, d, d, hitA, tB, hitB, cnt);
  float c = (float)cnt.steps * 0.001f;
  float ft = (float)p.tick, den = ft + 1.0f;
  float4 prev = p.accum[pix], o4;
  o4.x = fma_(prev.x, ft, c) / den;
  o4.y = fma_(prev.y, ft, c) / den;
  o4.z = fma_(prev.z, ft, c) / den;
  o4.w = 1.0f;
  p.accum[pix] = o4;
}

__global__ void k_math(int op, const float *a, const float *b, uint32_t n, float *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
  switch (op) {
    case 0: r = sin_(x); break;
    case 1: r = cos_(x); break;
    case 2: r = atan2_(x, y); break;
    case 3: r = asin_(x); break;
    case 4: r = exp2_(x); break;
    case 5: r = x / y; break;
    case 6: r = sqrt_(x); break;
    case 7: { float sd = x; r = rnd(sd); break; }
    case 8: r = fract_(x); break;
    case 9: r = log2_(x); break;
    case 10: r = pow_(x, y); break;
    default: r = 0.0f;
  }
  out[i] = r;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// Stacks deeper than the default 64 KB of dynamic LDS (trees deeper than ~31 levels under a 512-thread block) need the
// per-kernel opt-in; gfx950 has 160 KB per CU.
template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 48u * 1024u) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
static size_t stack_bytes(const DScene &S) { return (size_t)WAVES_PER_BLOCK * S.stack_n * WAVE * sizeof(int); }

hipError_t launch_trace(const TraceP &p, bool gen_rays, bool count, int num_cus, hipStream_t stream) {
  size_t lds = stack_bytes(p.scene);
  // persistent grid: enough resident blocks to fill every CU; the work counter balances the rest
  int blocks_per_cu = 4;
  uint32_t total_work = p.n_owned_tiles * p.tile * p.tile;
  uint32_t max_useful = (total_work + BLOCK_THREADS - 1) / BLOCK_THREADS;
  uint32_t grid = (uint32_t)(num_cus * blocks_per_cu);
  if (grid > max_useful) grid = max_useful;
  if (grid == 0) return hipSuccess;
  dim3 g(grid), b(BLOCK_THREADS);
  hipError_t e;
#define FSPT_LAUNCH_MEGA(G, C)                                                              \
  do {                                                                                        \
    if ((e = allow_lds(k_trace<G, C>, lds)) != hipSuccess) return e;                          \
    hipLaunchKernelGGL((k_trace<G, C>), g, b, lds, stream, p);                                \
  } while (0)
  if (gen_rays) { if (count) FSPT_LAUNCH_MEGA(true, true); else FSPT_LAUNCH_MEGA(true, false); }
  else { if (count) FSPT_LAUNCH_MEGA(false, true); else FSPT_LAUNCH_MEGA(false, false); }
#undef FSPT_LAUNCH_MEGA
  return hipGetLastError();
}

size_t wf_max_stack_entries() {
  // the largest LDS user per stack entry is the primary launch: 8 waves x 256 B per entry next to the staged tables
  const size_t lds_cu = 160u * 1024u, tables = WF_LDS_TABLE_MAX + (size_t)WF_PRIMARY_R_MAX * WF_PRIMARY_THREADS * 32u;
  return (lds_cu - tables) / ((WF_PRIMARY_THREADS / WAVE) * WAVE * sizeof(int));
}

hipError_t launch_wf(int kernel, const WfP &p, int count, int num_cus, hipStream_t stream) {
  // the most paths / samples the launch can have to process (sizes the grid; the kernels read the real counts)
  uint32_t total = p.n_batch * p.work_total;
  if (total == 0) return hipSuccess;
  if (p.ctl) {
    const unsigned long long lim = kernel == WF_K_PRIMARY ? (unsigned long long)p.take_max * 64ull * p.n_batch : (unsigned long long)p.cap;
    if (kernel != WF_K_RESOLVE && !(kernel == WF_K_TAIL && p.finish) && lim < total) total = (uint32_t)lim;
  }
  hipError_t e = hipSuccess;
  if (kernel == WF_K_PLAN) {
    hipLaunchKernelGGL(k_wf_plan, dim3(1), dim3(WAVE), 0, stream, p);
    return hipGetLastError();
  }
  if (kernel == WF_K_CARRY) {
    hipLaunchKernelGGL(k_wf_carry, dim3(64), dim3(BLOCK_THREADS), 0, stream, p);
    return hipGetLastError();
  }
  // two-level nodes (WfP::wide, one bit per kernel class): production kernels only, and only when the scene has them
  const bool wide = count == 0 && p.scene.quads != nullptr && kernel < WF_K_KINDS && ((p.wide >> kernel) & 1u) != 0u;
  if (kernel == WF_K_TRACE) {
    // persistent: the grid only has to fill the machine; the pool heads balance the work
    uint32_t grid = min((total + BLOCK_THREADS - 1) / BLOCK_THREADS, (uint32_t)num_cus * WF_TRACE_GRID_PER_CU);
    size_t lds = stack_bytes(p.scene) + (size_t)4 * WAVES_PER_BLOCK * WAVE * sizeof(int); // + four entries per lane (k_wf_trace: the kept shadow result, the extension ray's direction)
    // LDS left over per block at the occupancy the stacks (and the registers: 7 blocks) allow -> top-of-tree cache
    WfP q = p;
    {
      const size_t LDS_CU = 160u * 1024u;
      size_t blocks = LDS_CU / (lds ? lds : 1);
      if (blocks > WF_TRACE_BLOCKS_PER_CU) blocks = WF_TRACE_BLOCKS_PER_CU;
      if (blocks < 1) blocks = 1;
      size_t spare = (LDS_CU / blocks - lds) & ~(size_t)255; // allocation granularity
      const size_t rec = (size_t)(wide ? QUAD_F4 : NODE_F4) * sizeof(float4);
      uint32_t fit = (uint32_t)(spare / rec);
      q.lds_top = WF_TRACE_LDS_TOP ? min(min(fit, p.scene.n_top), (uint32_t)WF_TRACE_LDS_TOP) : 0u;
      lds += (size_t)q.lds_top * rec;
#if WF_TRACE_GRID_RESIDENT
      // no more blocks than are resident at once (the registers allow WF_TRACE_GRID_RESIDENT per CU, the stacks `blocks`):
      // every wave's first chunk of paths is its own, and a block that only starts when another has run out of work would
      // keep its waves' first chunks back until the end of the launch
      grid = min(grid, (uint32_t)num_cus * min((uint32_t)blocks, (uint32_t)WF_TRACE_GRID_RESIDENT));
#endif
    }
#define FSPT_LAUNCH_TRACE(C, A, Wd)                                                                        \
    do {                                                                                                     \
      if ((e = allow_lds(k_wf_trace<C, A, Wd>, lds)) != hipSuccess) return e;                                \
      hipLaunchKernelGGL((k_wf_trace<C, A, Wd>), dim3(grid), dim3(BLOCK_THREADS), lds, stream, q);           \
    } while (0)
    if (count == 1) FSPT_LAUNCH_TRACE(true, false, false);
    else if (count == 2) FSPT_LAUNCH_TRACE(true, true, false);
    else if (wide) FSPT_LAUNCH_TRACE(false, true, true);
    else FSPT_LAUNCH_TRACE(false, true, false);
#undef FSPT_LAUNCH_TRACE
  } else if (kernel == WF_K_TAIL) {
    // paths a block holds at a time (two lanes per path; one pair per wave when the kernel spreads a small launch's paths)
    const uint32_t per_block = WAVES_PER_BLOCK * (WF_TAIL_PAIRS_AUTO && !(p.ctl && p.finish) ? 1u : (uint32_t)WF_TAIL_PAIRS);
    uint32_t grid = min((total + per_block - 1) / per_block, (uint32_t)num_cus * (WF_TAIL_WAVES > 4 ? WF_TAIL_WAVES : 4));
    size_t lds = stack_bytes(p.scene);
#define FSPT_LAUNCH_TAIL(C, A, Wd)                                                                         \
    do {                                                                                                     \
      if (p.ctl && p.finish) {                                                                               \
        if ((e = allow_lds(k_wf_tail<C, A, true, Wd>, lds)) != hipSuccess) return e;                         \
        hipLaunchKernelGGL((k_wf_tail<C, A, true, Wd>), dim3(grid), dim3(BLOCK_THREADS), lds, stream, p);    \
      } else {                                                                                               \
        if ((e = allow_lds(k_wf_tail<C, A, false, Wd>, lds)) != hipSuccess) return e;                        \
        hipLaunchKernelGGL((k_wf_tail<C, A, false, Wd>), dim3(grid), dim3(BLOCK_THREADS), lds, stream, p);   \
      }                                                                                                      \
    } while (0)
    if (count == 1) FSPT_LAUNCH_TAIL(true, false, 0);
    else if (count == 2) FSPT_LAUNCH_TAIL(true, true, 0);
    else if (wide && p.tail_adaptive) FSPT_LAUNCH_TAIL(false, true, 2);
    else if (wide) FSPT_LAUNCH_TAIL(false, true, 1);
    else FSPT_LAUNCH_TAIL(false, true, 0);
#undef FSPT_LAUNCH_TAIL
  } else if (kernel == WF_K_LOGIC || kernel == WF_K_PRIMARY) {
    // resident blocks per CU at WF_LOGIC_WAVES waves per SIMD (4 SIMDs): 2 blocks of 512 threads at 4 waves; twice that many in flight
    const uint32_t threads = kernel == WF_K_PRIMARY ? (uint32_t)WF_PRIMARY_THREADS : (uint32_t)WF_LOGIC_THREADS;
    const uint32_t blocks_per_cu = WF_PL_OVERSUB * ((uint32_t)WF_LOGIC_WAVES * 4u * WAVE / threads);
    const uint32_t prim_r = p.primary_r >= 2u ? 2u : 1u;
    const uint32_t per_block = kernel == WF_K_PRIMARY ? threads * prim_r : threads;
    uint32_t grid = min((total + per_block - 1) / per_block, (uint32_t)num_cus * blocks_per_cu);
    if (kernel == WF_K_LOGIC) grid += p.carry_blocks; // (trailing blocks: k_wf_logic)
    const uint32_t tab_bytes = wf_table_bytes(p.scene.n_tex_sets, p.scene.n_bins, p.n_batch);
    const bool tab = WF_LOGIC_LDSTAB && tab_bytes <= WF_LDS_TABLE_MAX;
    if (kernel == WF_K_PRIMARY) {
      const size_t dyn = (size_t)(WF_PRIMARY_THREADS / WAVE) * p.scene.stack_n * WAVE * sizeof(int) + (tab ? tab_bytes : 0u) +
                         (prim_r > 1u ? (size_t)prim_r * WF_PRIMARY_THREADS * 32u : 0u); // + ray and hit of every sample of a block iteration
#define FSPT_LAUNCH_PRIMARY(C, T, RR, Wd)                                                                      \
      do {                                                                                                       \
        if ((e = allow_lds(k_wf_primary<C, T, RR, Wd>, dyn)) != hipSuccess) return e;                            \
        hipLaunchKernelGGL((k_wf_primary<C, T, RR, Wd>), dim3(grid), dim3(WF_PRIMARY_THREADS), dyn, stream, p);  \
      } while (0)
#define FSPT_LAUNCH_PRIMARY_R(C, T, Wd) do { if (prim_r > 1u) FSPT_LAUNCH_PRIMARY(C, T, 2, Wd); else FSPT_LAUNCH_PRIMARY(C, T, 1, Wd); } while (0)
      if (count) { if (tab) FSPT_LAUNCH_PRIMARY_R(true, true, false); else FSPT_LAUNCH_PRIMARY_R(true, false, false); }
      else if (wide) { if (tab) FSPT_LAUNCH_PRIMARY_R(false, true, true); else FSPT_LAUNCH_PRIMARY_R(false, false, true); }
      else { if (tab) FSPT_LAUNCH_PRIMARY_R(false, true, false); else FSPT_LAUNCH_PRIMARY_R(false, false, false); }
#undef FSPT_LAUNCH_PRIMARY_R
#undef FSPT_LAUNCH_PRIMARY
    } else {
      const size_t dyn = tab ? tab_bytes : 0u;
#define FSPT_LAUNCH_LOGIC(C, T) hipLaunchKernelGGL((k_wf_logic<C, T>), dim3(grid), dim3(WF_LOGIC_THREADS), dyn, stream, p)
      if (count) { if (tab) FSPT_LAUNCH_LOGIC(true, true); else FSPT_LAUNCH_LOGIC(true, false); }
      else { if (tab) FSPT_LAUNCH_LOGIC(false, true); else FSPT_LAUNCH_LOGIC(false, false); }
#undef FSPT_LAUNCH_LOGIC
    }
  } else {
    uint32_t grid = min((p.work_total + BLOCK_THREADS - 1) / BLOCK_THREADS, (uint32_t)num_cus * WF_RESOLVE_BLOCKS_PER_CU);
    hipLaunchKernelGGL(k_wf_resolve, dim3(grid), dim3(BLOCK_THREADS), 0, stream, p);
  }
  return hipGetLastError();
}

hipError_t launch_tile_pack(const TilePackP &p, bool unpack, hipStream_t stream) {
  const uint32_t n = p.n_owned_tiles * p.tile * p.tile;
  if (n == 0) return hipSuccess;
  const uint32_t grid = min((n + BLOCK_THREADS - 1) / BLOCK_THREADS, 4096u);
  if (unpack) hipLaunchKernelGGL(k_tile_pack<true>, dim3(grid), dim3(BLOCK_THREADS), 0, stream, p);
  else hipLaunchKernelGGL(k_tile_pack<false>, dim3(grid), dim3(BLOCK_THREADS), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_camera(uint32_t W, uint32_t H, uint32_t vw, uint32_t vh, const CameraP &cam, float rand_base, float4 *pos, float4 *dir,
                         hipStream_t stream) {
  uint32_t n = W * H;
  hipLaunchKernelGGL(k_camera, dim3((n + BLOCK_THREADS - 1) / BLOCK_THREADS), dim3(BLOCK_THREADS), 0, stream, W, H, vw, vh, cam,
                     rand_base, pos, dir);
  return hipGetLastError();
}

hipError_t launch_draw(const float4 *acc, uint32_t W, uint32_t H, float exposure, float saturation, int denoise,
                       float max_sigma, float scale, uint32_t *out, hipStream_t stream) {
  uint32_t n = W * H;
  hipLaunchKernelGGL(k_draw, dim3((n + BLOCK_THREADS - 1) / BLOCK_THREADS), dim3(BLOCK_THREADS), 0, stream, acc, W, H,
                     exposure, saturation, denoise, max_sigma, scale, out);
  return hipGetLastError();
}

hipError_t launch_bvh_test(const TraceP &p, hipStream_t stream) {
  uint32_t n = p.n_owned_tiles * p.tile * p.tile;
  if (n == 0) return hipSuccess;
  hipError_t e = allow_lds(k_bvh_test, stack_bytes(p.scene));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bvh_test, dim3((n + BLOCK_THREADS - 1) / BLOCK_THREADS), dim3(BLOCK_THREADS), stack_bytes(p.scene),
                     stream, p);
  return hipGetLastError();
}

hipError_t launch_intersect(const IntersectP &p, hipStream_t stream) {
  if (p.n == 0) return hipSuccess;
  hipError_t e = allow_lds(k_intersect, stack_bytes(p.scene));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_intersect, dim3((p.n + BLOCK_THREADS - 1) / BLOCK_THREADS), dim3(BLOCK_THREADS),
                     stack_bytes(p.scene), stream, p);
  return hipGetLastError();
}

hipError_t launch_math(int op, const float *a, const float *b, uint32_t n, float *out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, n, out);
  return hipGetLastError();
}

} // namespace fspt
