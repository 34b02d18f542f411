// fspt_api.cpp — host side of the libfspt C ABI (include/fspt.h).
//
// Mirrors the WebGL2 resource/draw-call layer of the reference's main.js:
// scene upload (initBVH 408-437, initAtlas 548-560), render targets
// (initBuffers 598-617), drawCamera (741-756), drawTracer (758-807), clear
// (826-836) and the tick loop (838-857).  No CPU fallback: every device entry
// point fails with FSPT_E_NO_DEVICE when there is no HIP device.
#include "fspt_internal.hpp"

#ifndef FSPT_NODE_TREELET
#define FSPT_NODE_TREELET 0 // nodes per treelet below the breadth-first top of the tree; 0 = pre-order (profiles/r02: A/B on the 1 M-triangle scene)
#endif

static thread_local char g_err[512] = "";

void fspt_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// RGBA8 image (row-major, w x h) -> 8 x 4-texel tiles (fspt_device.hpp: TEX_TILE_*), padded to whole tiles.
// Returns the number of texels of the tiled image; with src == nullptr only that.
// Bytes of interleaved four-layer texture images (TEXSET_QUAD) a scene may allocate; sets beyond it fetch their image
// layers from single-layer images (fspt_set_texture_interleave_budget).
static std::atomic<uint64_t> g_texset_budget{8ull << 30};

static size_t tile_image(const uint8_t *src, uint32_t w, uint32_t h, std::vector<uint32_t> &out) {
  const uint32_t tx = (w + fspt::TEX_TILE_W - 1) / fspt::TEX_TILE_W, ty = (h + fspt::TEX_TILE_H - 1) / fspt::TEX_TILE_H;
  const size_t n = (size_t)tx * ty * fspt::TEX_TILE_W * fspt::TEX_TILE_H;
  if (!src) return n;
  out.assign(n, 0u);
  for (uint32_t j = 0; j < h; ++j)
    for (uint32_t i = 0; i < w; ++i) {
      uint32_t v;
      std::memcpy(&v, src + ((size_t)j * w + i) * 4, 4);
      out[((size_t)(j / fspt::TEX_TILE_H) * tx + i / fspt::TEX_TILE_W) * (fspt::TEX_TILE_W * fspt::TEX_TILE_H) +
          (j % fspt::TEX_TILE_H) * fspt::TEX_TILE_W + (i % fspt::TEX_TILE_W)] = v;
    }
  return n;
}


int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fspt_set_error("no HIP device available (%s); libfspt has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return FSPT_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    fspt_set_error("device %d out of range (have %d)", device, n);
    return FSPT_E_INVALID;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    fspt_set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return FSPT_E_NO_DEVICE;
  }
  return FSPT_OK;
}

extern "C" {

const char *fspt_last_error(void) { return g_err; }
int fspt_abi_version(void) { return FSPT_ABI_VERSION; }

int fspt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fspt_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!free_bytes || !total_bytes) { fspt_set_error("fspt_device_memory: NULL argument"); return FSPT_E_INVALID; }
  if (fspt_device_count() <= 0) { fspt_set_error("fspt_device_memory: no HIP device"); return FSPT_E_NO_DEVICE; }
  size_t f = 0, t = 0;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemGetInfo(&f, &t));
  *free_bytes = f; *total_bytes = t;
  return FSPT_OK;
}

float fspt_rand_base_next(uint64_t *state) {
  uint64_t x = *state;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *state = x;
  uint64_t r = x * 2685821657736338717ULL;
  return ((float)(r >> 40) * (1.0f / 16777216.0f)) * 10000.0f;
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------
int fspt_set_texture_interleave_budget(uint64_t bytes) {
  g_texset_budget = bytes;
  return FSPT_OK;
}

int fspt_scene_create(const fspt_scene_desc *desc, int device, fspt_scene **out) {
  if (!desc || !out) { fspt_set_error("fspt_scene_create: NULL argument"); return FSPT_E_INVALID; }
  *out = nullptr;
  if (!desc->bvh || !desc->tri || !desc->mat || !desc->norm || !desc->uv || desc->n_nodes == 0 || desc->n_tris == 0) {
    fspt_set_error("fspt_scene_create: bvh/tri/mat/norm/uv must be non-empty");
    return FSPT_E_INVALID;
  }
  if (!desc->atlas || desc->atlas_res == 0 || desc->atlas_layers == 0) {
    fspt_set_error("fspt_scene_create: atlas must have at least one layer");
    return FSPT_E_INVALID;
  }
  if (!desc->bins || desc->n_bins == 0) {
    fspt_set_error("fspt_scene_create: radianceBins must hold at least one bin (main.js:292)");
    return FSPT_E_INVALID;
  }
  if (desc->env && (desc->env_w == 0 || desc->env_h == 0)) {
    fspt_set_error("fspt_scene_create: env given with zero size");
    return FSPT_E_INVALID;
  }
  if (desc->leaf_size == 0 || desc->leaf_size > 64) {
    fspt_set_error("fspt_scene_create: leaf_size %u out of range [1,64]", desc->leaf_size);
    return FSPT_E_INVALID;
  }
  const uint32_t N = desc->n_nodes, T = desc->n_tris;
  auto word = [&](uint32_t node, int w) -> int32_t {
    int32_t v;
    std::memcpy(&v, desc->bvh + (size_t)node * 9 + w, 4);
    return v;
  };
  // ---- validate + renumber interior nodes ------------------------------------------
  // The first TOP_BFS interior nodes in breadth-first order get the lowest numbers (every ray walks the top of
  // the tree: the traversal kernel keeps a prefix of them in LDS); the rest keep their pre-order.
  std::vector<int32_t> ref(N);
  std::vector<uint32_t> leaf_first; // first triangle of every leaf, in node order
  uint32_t n_interior = 0;
  for (uint32_t i = 0; i < N; ++i) {
    int32_t l = word(i, 0), r = word(i, 1), ts = word(i, 2);
    if (ts > -1) {
      if ((uint32_t)ts > T) { fspt_set_error("node %u: triStart %d > n_tris %u", i, ts, T); return FSPT_E_INVALID; }
      ref[i] = ~(int32_t)leaf_first.size(); // leaf record index
      leaf_first.push_back((uint32_t)ts);
    } else {
      // serializeTree is pre-order (bvh.js:33-50): children come after their parent.
      if (l <= (int32_t)i || r <= (int32_t)i || (uint32_t)l >= N || (uint32_t)r >= N) {
        fspt_set_error("node %u: child indices (%d,%d) violate pre-order / range [%u,%u)", i, l, r, i + 1, N);
        return FSPT_E_INVALID;
      }
      ref[i] = INT32_MAX; // interior, numbered below
      n_interior++;
    }
  }
  {
    const uint32_t TOP_BFS = 256;
    uint32_t next = 0;
    std::vector<uint32_t> queue;
    if (N && word(0, 2) <= -1) queue.push_back(0);
    for (size_t q = 0; q < queue.size() && next < TOP_BFS; ++q) {
      uint32_t i = queue[q];
      ref[i] = (int32_t)next++;
      uint32_t l = (uint32_t)word(i, 0), r = (uint32_t)word(i, 1);
      if (word(l, 2) <= -1) queue.push_back(l);
      if (word(r, 2) <= -1) queue.push_back(r);
    }
#if FSPT_NODE_TREELET > 1
    // Below the breadth-first top: TREELETS.  A treelet = a subtree root and its descendants in breadth-first order, up to
    // FSPT_NODE_TREELET nodes, stored contiguously; the treelets hanging off it follow, depth-first.  A ray that enters
    // a treelet finds the next few levels of its descent - and the sibling it will pop later - in the same or the next
    // 128-byte lines, instead of one line per level (pre-order keeps only the LEFT child next to its parent).  Only the
    // numbering changes: same nodes, same boxes, same traversal order, bit-identical results.
    {
      std::vector<uint32_t> roots; // subtree roots waiting to be laid out (a stack: depth-first over treelets)
      for (size_t q = queue.size(); q-- > 0;)
        if (ref[queue[q]] == INT32_MAX) roots.push_back(queue[q]); // discovered by the top's BFS but beyond its budget
      std::vector<uint32_t> local;
      while (!roots.empty()) {
        const uint32_t root = roots.back();
        roots.pop_back();
        local.assign(1, root);
        for (size_t q = 0; q < local.size(); ++q) {
          const uint32_t i = local[q];
          ref[i] = (int32_t)next++;
          const uint32_t ch[2] = {(uint32_t)word(i, 0), (uint32_t)word(i, 1)};
          for (uint32_t c : ch)
            if (word(c, 2) <= -1 && local.size() < (size_t)FSPT_NODE_TREELET) local.push_back(c);
        }
        // children of the treelet's nodes that did not fit: roots of the next treelets (right before left on the
        // stack, so the left subtree is laid out first, like pre-order)
        for (size_t q = local.size(); q-- > 0;) {
          const uint32_t i = local[q];
          const uint32_t ch[2] = {(uint32_t)word(i, 1), (uint32_t)word(i, 0)};
          for (uint32_t c : ch)
            if (word(c, 2) <= -1 && ref[c] == INT32_MAX) roots.push_back(c);
        }
      }
    }
#endif
    for (uint32_t i = 0; i < N; ++i)
      if (ref[i] == INT32_MAX) ref[i] = (int32_t)next++; // (pre-order for whatever is left: nothing, with treelets)
  }
  std::vector<float> nodes((size_t)(n_interior ? n_interior : 1) * 16, 0.0f);
  // depth of every node (root 0); a child's depth = parent's + 1
  std::vector<uint32_t> depth(N, 0);
  uint32_t max_depth = 0;
  for (uint32_t i = 0; i < N; ++i) {
    int32_t ts = word(i, 2);
    if (ts > -1) continue;
    int32_t l = word(i, 0), r = word(i, 1);
    depth[l] = depth[i] + 1;
    depth[r] = depth[i] + 1;
    if (depth[i] + 1 > max_depth) max_depth = depth[i] + 1;
    float *n = &nodes[(size_t)ref[i] * 16];
    const float *lb = desc->bvh + (size_t)l * 9 + 3, *rb = desc->bvh + (size_t)r * 9 + 3;
    n[0] = lb[0]; n[1] = lb[1]; n[2] = lb[3]; n[3] = lb[4];   // lmin.xy lmax.xy
    n[4] = rb[0]; n[5] = rb[1]; n[6] = rb[3]; n[7] = rb[4];   // rmin.xy rmax.xy
    n[8] = lb[2]; n[9] = lb[5]; n[10] = rb[2]; n[11] = rb[5]; // lmin.z lmax.z rmin.z rmax.z
    int32_t lr[4] = {ref[l], ref[r], 0, 0};
    std::memcpy(n + 12, lr, 16);
  }
  // ---- two-level nodes (fspt_device.hpp "quad"): the node records of both children side by side, one cache line ----
  // Usable only when every interior node's box IS the union of its children's boxes, bit for bit (true for bvh.js trees:
  // a node's box is built from its own triangles); checked here on the caller's arrays, no quads otherwise.
  bool quad_ok = n_interior > 0;
  std::vector<float> quads;
  if (quad_ok) {
    quads.assign((size_t)n_interior * 32, 0.0f);
    auto bits = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; };
    for (uint32_t i = 0; i < N && quad_ok; ++i) {
      if (word(i, 2) > -1) continue;
      const int32_t ch[2] = {word(i, 0), word(i, 1)};
      float *q = &quads[(size_t)ref[i] * 32];
      int32_t refs[8] = {fspt::REF_SENTINEL, fspt::REF_SENTINEL, ref[ch[0]], ref[ch[1]], fspt::REF_SENTINEL, fspt::REF_SENTINEL, 0, 0};
      for (int k = 0; k < 2 && quad_ok; ++k) {
        const int32_t c = ch[k];
        float *part = q + 16 * k;
        const float *cb = desc->bvh + (size_t)c * 9 + 3; // the child's own box: min.xyz max.xyz
        if (word(c, 2) > -1) { // a leaf: its own box, twice
          part[0] = part[4] = cb[0]; part[1] = part[5] = cb[1]; part[2] = part[6] = cb[3]; part[3] = part[7] = cb[4];
          part[8] = part[10] = cb[2]; part[9] = part[11] = cb[5];
        } else {
          const float *cn = &nodes[(size_t)ref[c] * 16];
          std::memcpy(part, cn, 48);
          std::memcpy(&refs[4 * k], cn + 12, 8);
        }
        // box(c) == union of the two boxes of its part, exactly?  (the comparison the device's v_min / v_max make; a pair
        // of candidates that compare equal must be the same bits: -0 / +0)
        const float lo[3][2] = {{part[0], part[4]}, {part[1], part[5]}, {part[8], part[10]}};
        const float hi[3][2] = {{part[2], part[6]}, {part[3], part[7]}, {part[9], part[11]}};
        for (int a = 0; a < 3 && quad_ok; ++a) {
          const float mn = lo[a][0] < lo[a][1] ? lo[a][0] : lo[a][1], mx = hi[a][0] > hi[a][1] ? hi[a][0] : hi[a][1];
          if (std::isnan(lo[a][0]) || std::isnan(lo[a][1]) || std::isnan(hi[a][0]) || std::isnan(hi[a][1])) quad_ok = false;
          if (lo[a][0] == lo[a][1] && bits(lo[a][0]) != bits(lo[a][1])) quad_ok = false;
          if (hi[a][0] == hi[a][1] && bits(hi[a][0]) != bits(hi[a][1])) quad_ok = false;
          if (bits(mn) != bits(cb[a]) || bits(mx) != bits(cb[3 + a])) quad_ok = false;
        }
      }
      std::memcpy(q + 12, &refs[0], 16);
      std::memcpy(q + 28, &refs[4], 16);
    }
    if (!quad_ok) quads.clear();
  }
  if (max_depth + 1 > 64 || max_depth + 1 > fspt::wf_max_stack_entries()) {
    // the reference's stack is int[64] (tracer.fs:368); here one entry per level, in LDS (all 64 fit: 128 KB of the CU's
    // 160 KB under the 512-thread primary launch)
    fspt_set_error("BVH depth %u exceeds the traversal stack (64)", max_depth);
    return FSPT_E_INVALID;
  }
  // ---- pre-edged triangles, padded by leaf_size "-1" triangles (main.js:150-152) ----
  const uint32_t TP = T + desc->leaf_size;
  std::vector<float> tris((size_t)TP * 9, 0.0f);
  for (uint32_t i = 0; i < TP; ++i) {
    float v[9];
    if (i < T) std::memcpy(v, desc->tri + (size_t)i * 9, 36);
    else for (int k = 0; k < 9; ++k) v[k] = -1.0f;
    float *o = &tris[(size_t)i * 9];
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    o[3] = v[3] - v[0]; o[4] = v[4] - v[1]; o[5] = v[5] - v[2]; // e1 = v2 - v1 (tracer.fs:301)
    o[6] = v[6] - v[0]; o[7] = v[7] - v[1]; o[8] = v[8] - v[2]; // e2 = v3 - v1 (tracer.fs:302)
  }
  // ---- leaf records: the leaf_size triangles processLeaf reads from each leaf's first one, component-major ----
  const uint32_t LS = desc->leaf_size;
  const size_t n_leaves = leaf_first.size();
  std::vector<float> leaves((n_leaves ? n_leaves : 1) * (size_t)LS * 9, 0.0f);
  std::vector<uint32_t> slot_tri((n_leaves ? n_leaves : 1) * (size_t)LS, 0u);
  for (size_t L = 0; L < n_leaves; ++L) {
    float *rec = &leaves[L * LS * 9];
    for (uint32_t k = 0; k < LS; ++k) {
      const uint32_t ti = leaf_first[L] + k; // <= T - 1 + leaf_size: inside the padded array
      for (int c = 0; c < 9; ++c) rec[(size_t)c * LS + k] = tris[(size_t)ti * 9 + c];
      slot_tri[L * LS + k] = ti;
    }
  }
  // ---- material texture sets: the four atlas layers a triangle samples at one uv (tracer.fs:453-456) ----
  // layer = clamp(floor(id + 0.5), 0, layers - 1) as texture(sampler2DArray) selects it; same binary32 arithmetic here
  const uint32_t n_layers = desc->atlas_layers;
  auto layer_of = [&](float id) -> uint32_t {
    const float x = std::floor(id + 0.5f);
    if (!(x >= 0.0f)) return 0u; // negative, NaN (the device's float -> int conversion gives 0 for NaN)
    if (x >= (float)(n_layers - 1u)) return n_layers - 1u;
    return (uint32_t)x;
  };
  std::map<std::array<uint32_t, 4>, uint32_t> set_ids;
  std::vector<std::array<uint32_t, 4>> set_keys;
  std::vector<uint32_t> tri_set(T);
  for (uint32_t i = 0; i < T; ++i) {
    const float *m = desc->mat + (size_t)i * 12;
    const std::array<uint32_t, 4> key = {layer_of(m[0]), layer_of(m[1]), layer_of(m[3]), layer_of(m[2])}; // diffuse, emissive, mr, normal
    auto it = set_ids.find(key);
    if (it == set_ids.end()) {
      it = set_ids.emplace(key, (uint32_t)set_keys.size()).first;
      set_keys.push_back(key);
    }
    tri_set[i] = it->second;
  }
  // ---- 192-byte hit records, one per leaf SLOT (what the traversal reports): slot (L, k) holds triangle leaf_first[L] + k ----
  bool has_dielectric = false;
  const size_t n_slots = (n_leaves ? n_leaves : 1) * (size_t)LS;
  std::vector<float> shade(n_slots * 48, 0.0f);
  for (size_t sl = 0; sl < n_leaves * LS; ++sl) {
    const uint32_t i = slot_tri[sl];
    if (i >= T) continue; // "-1" padding: never hit (det = 0)
    float *o = &shade[sl * 48];
    std::memcpy(o, &tris[(size_t)i * 9], 36);
    std::memcpy(o + 9, desc->norm + (size_t)i * 27, 27 * 4);
    std::memcpy(o + 36, desc->uv + (size_t)i * 6, 6 * 4);
    const float *m = desc->mat + (size_t)i * 12;
    std::memcpy(&o[42], &tri_set[i], 4);                     // material texture set (diffuse, emissive, mr, normal layers)
    o[46] = m[9]; o[47] = m[10];                             // ior, dielectric
  }
  for (uint32_t i = 0; i < T; ++i)
    if (desc->mat[(size_t)i * 12 + 10] >= 0.0f) has_dielectric = true;

  int rc = check_device(device);
  if (rc) return rc;
  fspt_scene *s = new fspt_scene();
  s->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount;
  auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = upload(&s->nodes, nodes.data(), nodes.size() * 4);
  if (e == hipSuccess && quad_ok) e = upload(&s->quads, quads.data(), quads.size() * 4);
  if (e == hipSuccess) e = upload(&s->tris, leaves.data(), leaves.size() * 4);
  if (e == hipSuccess) e = upload(&s->slot_tri, slot_tri.data(), slot_tri.size() * 4);
  if (e == hipSuccess) e = upload(&s->shade, shade.data(), shade.size() * 4);
  // Atlas.  A layer whose texels are all equal - every flat colour: TexturePacker fills whole layers with them
  // (texture_packer.js:36-42), and a colours-only atlas is 1 x 1 - is never stored: its texel sits in the sets that use
  // it.  A set with two or more image layers gets ONE interleaved image (16-byte texels: diffuse, emissive, mr, normal;
  // 4 x 2-texel tiles = 128 bytes), as long as the interleaving budget lasts; the image layers of the other sets are
  // stored once each as single-layer images in 8 x 4-texel tiles.
  uint32_t n_sets = (uint32_t)set_keys.size();
  if (e == hipSuccess) {
    const uint32_t res = desc->atlas_res;
    const size_t per_layer = (size_t)res * res;
    std::vector<uint8_t> is_const(n_layers, 1);
    std::vector<uint32_t> first(n_layers, 0u);
    for (uint32_t l = 0; l < n_layers; ++l) {
      const uint8_t *src = desc->atlas + (size_t)l * per_layer * 4;
      std::memcpy(&first[l], src, 4);
      for (size_t k = 1; k < per_layer && is_const[l]; ++k) is_const[l] = std::memcmp(src + k * 4, &first[l], 4) == 0;
    }
    const uint32_t qtx = (res + 3u) / 4u, qty = (res + 1u) / 2u;
    const size_t quad_tiles = (size_t)qtx * qty;             // 128-byte tiles per interleaved image
    std::vector<uint32_t> tab((size_t)n_sets * 12, 0u), kind(n_sets, fspt::TEXSET_CONST);
    std::vector<int64_t> layer_base(n_layers, -1);            // single-layer image of a layer, in tiles (-1: not stored)
    std::vector<uint32_t> single;                              // the single-layer images, tiled
    std::vector<uint32_t> tiled;
    uint64_t quad_bytes = 0;
    uint32_t n_quad = 0;
    for (uint32_t si = 0; si < n_sets; ++si) {
      const auto &key = set_keys[si];
      uint32_t n_img = 0, distinct[4];
      for (int k = 0; k < 4; ++k) {
        if (is_const[key[k]]) continue;
        bool seen = false;
        for (uint32_t q = 0; q < n_img; ++q) seen = seen || distinct[q] == key[k];
        if (!seen) distinct[n_img++] = key[k];
      }
      if (n_img >= 2 && quad_bytes + quad_tiles * 128u <= g_texset_budget && (n_quad + 1ull) * quad_tiles < 0xFFFFFFFFull) {
        kind[si] = fspt::TEXSET_QUAD;
        quad_bytes += quad_tiles * 128u;
        n_quad++;
      } else if (n_img >= 1) {
        kind[si] = fspt::TEXSET_SEPARATE;
      }
    }
    // single-layer images: the image layers of SEPARATE sets
    for (uint32_t si = 0; si < n_sets; ++si) {
      if (kind[si] != fspt::TEXSET_SEPARATE) continue;
      for (int k = 0; k < 4; ++k) {
        const uint32_t l = set_keys[si][k];
        if (is_const[l] || layer_base[l] >= 0) continue;
        layer_base[l] = (int64_t)(single.size() / (fspt::TEX_TILE_W * fspt::TEX_TILE_H));
        tile_image(desc->atlas + (size_t)l * per_layer * 4, res, res, tiled);
        single.insert(single.end(), tiled.begin(), tiled.end());
      }
    }
    if (single.size() / (fspt::TEX_TILE_W * fspt::TEX_TILE_H) >= 0xFFFFFFFFull) {
      fspt_set_error("atlas too large: %zu texels of image layers", single.size());
      fspt_scene_destroy(s);
      return FSPT_E_INVALID;
    }
    if (e == hipSuccess) e = upload(&s->atlas, single.data(), single.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&s->atlas4, quad_bytes ? quad_bytes : 16);
    std::vector<uint32_t> quad;
    uint32_t qi = 0;
    for (uint32_t si = 0; si < n_sets && e == hipSuccess; ++si) {
      const auto &key = set_keys[si];
      uint32_t *q = &tab[(size_t)si * 12];
      q[0] = kind[si];
      for (int k = 0; k < 4; ++k) {
        q[4 + k] = first[key[k]];
        q[8 + k] = (kind[si] == fspt::TEXSET_SEPARATE && !is_const[key[k]]) ? (uint32_t)layer_base[key[k]] : fspt::LAYER_CONST;
      }
      if (kind[si] != fspt::TEXSET_QUAD) continue;
      q[1] = (uint32_t)(qi * quad_tiles);
      quad.assign(quad_tiles * 32, 0u);
      for (int k = 0; k < 4; ++k) {
        const uint32_t l = key[k];
        const uint8_t *src = desc->atlas + (size_t)l * per_layer * 4;
        for (uint32_t jy = 0; jy < res; ++jy)
          for (uint32_t ix = 0; ix < res; ++ix) {
            uint32_t v = first[l];
            if (!is_const[l]) std::memcpy(&v, src + ((size_t)jy * res + ix) * 4, 4);
            quad[(((size_t)(jy >> 1) * qtx + (ix >> 2)) * 8 + ((jy & 1u) << 2) + (ix & 3u)) * 4 + k] = v;
          }
      }
      e = hipMemcpy((char *)s->atlas4 + (size_t)qi * quad_tiles * 128u, quad.data(), quad.size() * 4, hipMemcpyHostToDevice);
      qi++;
    }
    if (e == hipSuccess) e = upload(&s->tex_sets, tab.data(), tab.size() * 4);
  }
  if (e == hipSuccess && desc->env) {
    std::vector<uint32_t> tiled;
#if FSPT_ENV_APRON
    // overlapping 8 x 4-texel tiles: tile (a, b) = texels [7a, 7a + 8) x [3b, 3b + 4), REPEAT in s, CLAMP in t (main.js:174-178)
    const uint32_t w = desc->env_w, h = desc->env_h, tx = (w + 6u) / 7u, ty = (h + 2u) / 3u;
    tiled.assign((size_t)tx * ty * 32u, 0u);
    for (uint32_t b = 0; b < ty; ++b)
      for (uint32_t a = 0; a < tx; ++a)
        for (uint32_t lb = 0; lb < 4; ++lb)
          for (uint32_t la = 0; la < 8; ++la) {
            const uint32_t i = (7u * a + la) % w;
            uint32_t j = 3u * b + lb;
            if (j > h - 1u) j = h - 1u;
            std::memcpy(&tiled[((size_t)b * tx + a) * 32u + lb * 8u + la], desc->env + ((size_t)j * w + i) * 4, 4);
          }
#else
    tile_image(desc->env, desc->env_w, desc->env_h, tiled);
#endif
    e = upload(&s->env, tiled.data(), tiled.size() * 4);
  }
  if (e == hipSuccess) e = upload(&s->bins, desc->bins, (size_t)desc->n_bins * 16);
  if (e != hipSuccess) {
    fspt_set_error("scene upload failed: %s", hipGetErrorString(e));
    fspt_scene_destroy(s);
    return FSPT_E_HIP;
  }
  s->d.nodes = (const float4 *)s->nodes;
  s->d.quads = (const float4 *)s->quads; // NULL when the boxes are not unions (see above)
  s->d.leaves = (const float *)s->tris;
  s->d.slot_tri = (const uint32_t *)s->slot_tri;
  s->d.hitrec = (const float4 *)s->shade;
  s->d.atlas = (const uint32_t *)s->atlas;
  s->d.atlas4 = (const uint4 *)s->atlas4;
  s->d.tex_sets = (const uint4 *)s->tex_sets;
  s->d.n_tex_sets = n_sets;
  s->d.env = (const uint32_t *)s->env;
  s->d.bins = (const uint4 *)s->bins;
  s->d.atlas_res = desc->atlas_res;
  s->d.atlas_layers = desc->atlas_layers;
  s->d.env_w = desc->env ? desc->env_w : 0;
  s->d.env_h = desc->env ? desc->env_h : 0;
  s->d.n_bins = desc->n_bins;
  s->d.leaf_size = desc->leaf_size;
  s->d.root_ref = ref[0];
  // Entries of a lane's traversal stack: one more than the walk can use (it pushes a far child only at an interior node,
  // the deepest of which sits at depth max_depth - 1).  Dropping the spare entry gives the 1 M-triangle scene - depth 22 -
  // its sixth resident trace block per CU at the price of the LDS copy of the top of the tree (8 nodes instead of 31):
  // measured +-0 (profiles/r06/ab_final_constants_c3.log, f_spare0), so the spare stays.
  s->d.stack_n = max_depth + 1;
  s->d.n_top = n_interior < 256u ? n_interior : 256u; // interior nodes numbered breadth-first
  s->depth = max_depth;
  s->n_nodes = N;
  s->n_tris = T;
  s->n_interior = n_interior;
  s->has_dielectric = has_dielectric;
  *out = s;
  return FSPT_OK;
}

int fspt_scene_destroy(fspt_scene *s) {
  if (!s) return FSPT_OK;
  hipSetDevice(s->device);
  hipFree(s->nodes); hipFree(s->quads); hipFree(s->tris); hipFree(s->slot_tri); hipFree(s->shade); hipFree(s->atlas); hipFree(s->atlas4); hipFree(s->tex_sets); hipFree(s->env); hipFree(s->bins);
  delete s;
  return FSPT_OK;
}

int fspt_scene_depth(const fspt_scene *s, uint32_t *depth) {
  if (!s || !depth) { fspt_set_error("fspt_scene_depth: NULL argument"); return FSPT_E_INVALID; }
  *depth = s->depth;
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// target
// ---------------------------------------------------------------------------
int fspt_target_create(fspt_scene *scene, uint32_t W, uint32_t H, fspt_target **out) {
  if (!scene || !out || W == 0 || H == 0) { fspt_set_error("fspt_target_create: bad argument"); return FSPT_E_INVALID; }
  if ((uint64_t)W * H > (1ull << 30)) { fspt_set_error("fspt_target_create: %ux%u too large", W, H); return FSPT_E_INVALID; }
  int rc = check_device(scene->device);
  if (rc) return rc;
  fspt_target *t = new fspt_target();
  t->scene = scene;
  t->W = W; t->H = H;
  t->vw = W; t->vh = H;
  size_t px = (size_t)W * H;
  hipError_t e = hipMalloc((void **)&t->accum_own, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->ray_pos, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->ray_dir, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->work_counters, WORK_RING * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&t->counters, 8 * 8); // 6 work counters (fspt_counters) + [6] = traversal steps the trace kernel served from LDS
  if (e == hipSuccess) e = hipStreamCreate(&t->stream);
  if (e == hipSuccess) e = hipEventCreate(&t->ev0);
  if (e == hipSuccess) e = hipEventCreate(&t->ev1);
  if (e == hipSuccess) e = hipEventCreate(&t->ev_start);
  if (e == hipSuccess) e = hipEventCreate(&t->prim_ev[0]);
  if (e == hipSuccess) e = hipEventCreate(&t->prim_ev[1]);
  {
    fspt_target::WfLane &ln = t->wf;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.stream_b, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.resolved, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_run, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_b_last, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ctl_ready, hipEventDisableTiming);
    for (int k = 0; k < fspt::WF_RING; ++k) {
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_logic[k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_b[k], hipEventDisableTiming);
    }
  }
  if (e == hipSuccess) e = hipMemsetAsync(t->accum_own, 0, px * 16, t->stream);
  if (e == hipSuccess) e = hipMemsetAsync(t->counters, 0, 64, t->stream);
  if (e != hipSuccess) {
    fspt_set_error("fspt_target_create: %s", hipGetErrorString(e));
    fspt_target_destroy(t);
    return FSPT_E_HIP;
  }
  t->accum = t->accum_own;
  *out = t;
  return FSPT_OK;
}

int fspt_target_destroy(fspt_target *t) {
  if (!t) return FSPT_OK;
  hipSetDevice(t->scene->device);
  // Recorded ticks are dropped: nothing can observe the library's own accumulator any more, and a caller-owned one
  // (fspt_target_bind_accumulator) may already have been freed by its owner - destroy never writes to it.  A caller
  // that wants the recorded ticks in its buffer calls fspt_sync (or re-binds, which flushes) first.
  t->pending.clear();
  if (t->stream) hipStreamSynchronize(t->stream);
  hipFree(t->accum_own); hipFree(t->ray_pos); hipFree(t->ray_dir); hipFree(t->work_counters); hipFree(t->counters);
  {
    fspt_target::WfLane &ln = t->wf;
    for (void *m : ln.mem) hipFree(m);
    hipFree(ln.counts);
    hipFree(ln.heads);
    if (ln.counts_host) hipHostFree(ln.counts_host);
    if (ln.live_host) hipHostFree(ln.live_host);
    if (ln.counts_ready) hipEventDestroy(ln.counts_ready);
    if (ln.resolved) hipEventDestroy(ln.resolved);
    if (ln.stream_b) hipStreamSynchronize(ln.stream_b);
    hipFree(ln.ctl);
    for (int *b : ln.susp) hipFree(b);
    if (ln.ctl_host) hipHostFree(ln.ctl_host);
    for (hipEvent_t ev : {ln.ev_run, ln.ev_b_last, ln.ctl_ready}) if (ev) hipEventDestroy(ev);
    for (int k = 0; k < fspt::WF_RING; ++k) { if (ln.ev_logic[k]) hipEventDestroy(ln.ev_logic[k]); if (ln.ev_b[k]) hipEventDestroy(ln.ev_b[k]); }
    if (ln.stream_b) hipStreamDestroy(ln.stream_b);
    if (ln.stream) hipStreamDestroy(ln.stream);
  }
  if (t->ev_start) hipEventDestroy(t->ev_start);
  for (hipEvent_t ev : t->prim_ev) if (ev) hipEventDestroy(ev);
  for (hipEvent_t e : t->ev_pool) hipEventDestroy(e);
  if (t->ev0) hipEventDestroy(t->ev0);
  if (t->ev1) hipEventDestroy(t->ev1);
  if (t->stream) hipStreamDestroy(t->stream);
  delete t;
  return FSPT_OK;
}

int fspt_target_set_shard(fspt_target *t, uint32_t shard, uint32_t n_shards, uint32_t tile) {
  if (!t) { fspt_set_error("fspt_target_set_shard: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (n_shards == 0 || shard >= n_shards) { fspt_set_error("shard %u of %u invalid", shard, n_shards); return FSPT_E_INVALID; }
  if (tile == 0 || tile % 8 != 0 || tile > 256) { fspt_set_error("tile %u must be a multiple of 8 in [8,256]", tile); return FSPT_E_INVALID; }
  if (t->shard != shard || t->n_shards != n_shards || t->tile != tile) {
    prim_reset(t);
    if (t->stream_fallback) { HIP_TRY(hipSetDevice(t->scene->device)); wf_release(t->wf); t->stream_fallback = false; }
  }
  t->shard = shard; t->n_shards = n_shards; t->tile = tile;
  return FSPT_OK;
}

int fspt_target_bind_accumulator(fspt_target *t, void *device_ptr) {
  if (!t) { fspt_set_error("fspt_target_bind_accumulator: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->accum = device_ptr ? (float4 *)device_ptr : t->accum_own;
  return FSPT_OK;
}

int fspt_target_size(fspt_target *t, uint32_t *W, uint32_t *H) {
  if (!t || !W || !H) { fspt_set_error("fspt_target_size: NULL argument"); return FSPT_E_INVALID; }
  *W = t->W; *H = t->H;
  return FSPT_OK;
}

int fspt_target_accumulator(fspt_target *t, void **device_ptr) {
  if (!t || !device_ptr) { fspt_set_error("fspt_target_accumulator: NULL argument"); return FSPT_E_INVALID; }
  *device_ptr = t->accum;
  return FSPT_OK;
}

int fspt_camera(fspt_target *t, const float P[3], const float I[3], float fov_scale, const float lens[2],
                float rand_base) {
  if (!t || !P || !I || !lens) { fspt_set_error("fspt_camera: NULL argument"); return FSPT_E_INVALID; }
  // drawCamera is recorded, not launched: the ticks that use these rays generate them inside the path kernel (the ray
  // textures are only written when somebody looks at them: fspt_read_rays, fspt_trace_test)
  std::memset(&t->last_cam, 0, sizeof(t->last_cam));
  std::memcpy(t->last_cam.P, P, 12); std::memcpy(t->last_cam.I, I, 12);
  t->last_cam.fov_scale = fov_scale; t->last_cam.lens[0] = lens[0]; t->last_cam.lens[1] = lens[1];
  t->last_rb_cam = rand_base;
  t->cam_recorded = true;
  t->rays_injected = false;
  t->rays_valid = true;
  return FSPT_OK;
}

int fspt_set_rays(fspt_target *t, const float *pos, const float *dir) {
  if (!t || !pos || !dir) { fspt_set_error("fspt_set_rays: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  t->cam_recorded = false;
  t->rays_injected = true;
  size_t bytes = (size_t)t->W * t->H * 16;
  HIP_TRY(hipMemcpyAsync(t->ray_pos, pos, bytes, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->ray_dir, dir, bytes, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  t->rays_valid = true;
  return FSPT_OK;
}

int fspt_read_rays(fspt_target *t, float *pos, float *dir) {
  if (!t || !pos || !dir) { fspt_set_error("fspt_read_rays: NULL argument"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_read_rays: no rays generated yet"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  { int rcm = materialise_rays(t); if (rcm) return rcm; }
  size_t bytes = (size_t)t->W * t->H * 16;
  HIP_TRY(hipMemcpyAsync(pos, t->ray_pos, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipMemcpyAsync(dir, t->ray_dir, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

} // extern "C"

// Every path ends after MAX_PATH_ITERS loop iterations (the cap on tracer.fs:488's `i--`), and `i` never exceeds the
// iteration count: a larger NUM_BOUNCES cannot change any sample.  Clamping keeps the per-round tables
// (WfCounts[WF_ROUNDS_MAX + 2], the 8-bit bounce field of the path flags) in range for any caller value.
uint32_t clamp_bounces(uint32_t nb) { return nb > (uint32_t)FSPT_MAX_BOUNCES ? (uint32_t)FSPT_MAX_BOUNCES : nb; }

// n_ticks ticks with ray generation in the path kernels and explicit per-tick randBase values, on either pipeline
static int render_ticks(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                        const float *rbc, const float *rbt) {
  t->ev_used = 0; t->ev_overflow = false;
  if (t->pipeline == 1) {
    HIP_TRY(hipEventRecord(t->ev0, t->stream));
    int rc;
    if (t->sched == 1 || t->stream_fallback) {
      rc = render_stream(t, cam, first_tick, n_ticks, rbc, rbt, false);
    } else {
      rc = render_wavefront(t, cam, first_tick, n_ticks, rbc, rbt, false);
      if (rc == FSPT_E_NOMEM) {
        // fewer than FSPT_MIN_BATCH ticks of path state fit (nothing has been launched yet): the bounded pool instead
        t->stream_fallback = true;
        rc = render_stream(t, cam, first_tick, n_ticks, rbc, rbt, false);
      }
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(t->ev1, t->stream));
    t->timed = true; t->last_launches = n_ticks;
    return FSPT_OK;
  }
  fspt::TraceP p{};
  fill_trace_params(t, p);
  std::memcpy(p.cam.P, cam->P, 12); std::memcpy(p.cam.I, cam->I, 12);
  p.cam.fov_scale = cam->fov_scale; p.cam.lens[0] = cam->lens[0]; p.cam.lens[1] = cam->lens[1];
  p.env_theta = cam->env_theta; p.num_bounces = cam->num_bounces;
  bool first = true;
  uint32_t done = 0;
  while (done < n_ticks) {
    uint32_t batch = n_ticks - done < WORK_RING ? n_ticks - done : WORK_RING;
    HIP_TRY(hipMemsetAsync(t->work_counters, 0, (size_t)batch * 4, t->stream));
    if (first) { HIP_TRY(hipEventRecord(t->ev0, t->stream)); first = false; }
    for (uint32_t k = 0; k < batch; ++k) {
      p.rand_base_cam = rbc[done + k];
      p.rand_base = rbt[done + k];
      p.tick = first_tick + done + k;
      p.work_counter = t->work_counters + k;
      HIP_TRY(fspt::launch_trace(p, true, t->count != 0, t->scene->num_cus, t->stream));
    }
    done += batch;
  }
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = n_ticks;
  return FSPT_OK;
}

// Execute the recorded two-call ticks: runs of consecutive ticks with the same camera / envTheta / NUM_BOUNCES go
// through the batched path (ray generation in the kernel from the recorded randBase values - the same arithmetic as
// k_camera followed by a trace of the ray buffers, tests/test_parity_gpu.py).  Called by everything that observes or
// changes state the ticks depend on.
static bool same_view(const fspt_camera_params &a, const fspt_camera_params &b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }
int flush_pending(fspt_target *t) {
  if (t->pending.empty()) return FSPT_OK;
  std::vector<fspt_target::Deferred> q;
  q.swap(t->pending); // (a failing batch drops the rest: the error is reported once)
  HIP_TRY(hipSetDevice(t->scene->device));
  std::vector<float> rbc, rbt;
  size_t i = 0;
  while (i < q.size()) {
    size_t j = i + 1;
    while (j < q.size() && same_view(q[j].cam, q[i].cam) && q[j].tick == q[j - 1].tick + 1) ++j;
    rbc.clear(); rbt.clear();
    for (size_t k = i; k < j; ++k) { rbc.push_back(q[k].rb_cam); rbt.push_back(q[k].rb_trace); }
    int rc = render_ticks(t, &q[i].cam, q[i].tick, (uint32_t)(j - i), rbc.data(), rbt.data());
    if (rc) return rc;
    i = j;
  }
  return FSPT_OK;
}

// the ray buffers as the most recent fspt_camera call left them (drawCamera's two render targets)
int materialise_rays(fspt_target *t) {
  if (!t->cam_recorded) return FSPT_OK;
  fspt::CameraP c;
  std::memcpy(c.P, t->last_cam.P, 12); std::memcpy(c.I, t->last_cam.I, 12);
  c.fov_scale = t->last_cam.fov_scale; c.lens[0] = t->last_cam.lens[0]; c.lens[1] = t->last_cam.lens[1];
  HIP_TRY(fspt::launch_camera(t->W, t->H, t->vw, t->vh, c, t->last_rb_cam, t->ray_pos, t->ray_dir, t->stream));
  t->cam_recorded = false;
  return FSPT_OK;
}

extern "C" {

int fspt_trace(fspt_target *t, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces) {
  if (!t) { fspt_set_error("fspt_trace: NULL target"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_trace: call fspt_camera or fspt_set_rays first"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  num_bounces = clamp_bounces(num_bounces);
  if (!t->rays_injected) {
    // rays come from fspt_camera: record the tick; it runs with its neighbours in one batch at the next flush point
    fspt_target::Deferred d;
    d.cam = t->last_cam; d.cam.env_theta = env_theta; d.cam.num_bounces = num_bounces;
    d.rb_cam = t->last_rb_cam; d.tick = tick; d.rb_trace = rand_base;
    t->pending.push_back(d);
    if (!t->defer || t->pending.size() >= (size_t)t->batch_ticks) return flush_pending(t);
    return FSPT_OK;
  }
  FLUSH_OR_RETURN(t);
  if (t->pipeline == 1) {
    fspt_camera_params cp{};
    cp.env_theta = env_theta; cp.num_bounces = num_bounces;
    t->ev_used = 0; t->ev_overflow = false;
    HIP_TRY(hipEventRecord(t->ev0, t->stream));
    int rc;
    if (t->sched == 1 || t->stream_fallback) {
      rc = render_stream(t, &cp, tick, 1, nullptr, &rand_base, true);
    } else {
      rc = render_wavefront(t, &cp, tick, 1, nullptr, &rand_base, true);
      if (rc == FSPT_E_NOMEM) { // as in render_ticks: the bounded pool when the batch scheduler's path state does not fit
        t->stream_fallback = true;
        rc = render_stream(t, &cp, tick, 1, nullptr, &rand_base, true);
      }
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(t->ev1, t->stream));
    t->timed = true; t->last_launches = 1;
    return FSPT_OK;
  }
  t->ev_used = 0;
  fspt::TraceP p{};
  fill_trace_params(t, p);
  p.tick = tick; p.rand_base = rand_base; p.rand_base_cam = 0.0f; p.env_theta = env_theta; p.num_bounces = num_bounces;
  HIP_TRY(hipMemsetAsync(t->work_counters, 0, 4, t->stream));
  p.work_counter = t->work_counters;
  HIP_TRY(hipEventRecord(t->ev0, t->stream));
  HIP_TRY(fspt::launch_trace(p, false, t->count != 0, t->scene->num_cus, t->stream));
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = 1;
  return FSPT_OK;
}

int fspt_trace_test(fspt_target *t, uint32_t tick) {
  if (!t) { fspt_set_error("fspt_trace_test: NULL target"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_trace_test: call fspt_camera or fspt_set_rays first"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  int rcm = materialise_rays(t);
  if (rcm) return rcm;
  fspt::TraceP p{};
  fill_trace_params(t, p);
  p.tick = tick;
  t->ev_used = 0;
  HIP_TRY(hipEventRecord(t->ev0, t->stream));
  HIP_TRY(fspt::launch_bvh_test(p, t->stream));
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = 1;
  return FSPT_OK;
}

int fspt_render(fspt_target *t, const fspt_camera_params *cam_in, uint32_t first_tick, uint32_t n_ticks, uint64_t seed) {
  if (!t || !cam_in) { fspt_set_error("fspt_render: NULL argument"); return FSPT_E_INVALID; }
  if (n_ticks == 0) return FSPT_OK;
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  fspt_camera_params cam_c = *cam_in;
  cam_c.num_bounces = clamp_bounces(cam_c.num_bounces);
  std::vector<float> rbc(n_ticks), rbt(n_ticks);
  uint64_t st0 = seed;
  for (uint32_t k = 0; k < n_ticks; ++k) { rbc[k] = fspt_rand_base_next(&st0); rbt[k] = fspt_rand_base_next(&st0); }
  return render_ticks(t, &cam_c, first_tick, n_ticks, rbc.data(), rbt.data());
}

int fspt_target_set_deferred(fspt_target *t, int enable) {
  if (!t) { fspt_set_error("fspt_target_set_deferred: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->defer = enable != 0;
  return FSPT_OK;
}

int fspt_clear(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_clear: NULL target"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipMemsetAsync(t->accum, 0, (size_t)t->W * t->H * 16, t->stream));
  return FSPT_OK;
}

int fspt_sync(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_sync: NULL target"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

int fspt_read_radiance(fspt_target *t, float *out) {
  if (!t || !out) { fspt_set_error("fspt_read_radiance: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipMemcpyAsync(out, t->accum, (size_t)t->W * t->H * 16, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

int fspt_draw(fspt_target *t, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8) {
  return fspt_draw_scaled(t, exposure, saturation, denoise, max_sigma, 1.0f, out_rgba8);
}

int fspt_draw_scaled(fspt_target *t, float exposure, float saturation, int denoise, float max_sigma, float scale,
                     uint8_t *out_rgba8) {
  if (!t || !out_rgba8) { fspt_set_error("fspt_draw: NULL argument"); return FSPT_E_INVALID; }
  if (!(scale > 0.0f && scale <= 1.0f)) { fspt_set_error("fspt_draw: scale must be in (0, 1]"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  size_t n = (size_t)t->W * t->H;
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, n * 4));
  hipError_t e = fspt::launch_draw(t->accum, t->W, t->H, exposure, saturation, denoise, max_sigma, scale, d, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_rgba8, d, n * 4, hipMemcpyDeviceToHost, t->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if (e != hipSuccess) { fspt_set_error("fspt_draw: %s", hipGetErrorString(e)); return FSPT_E_HIP; }
  return FSPT_OK;
}

int fspt_last_kernel_ms(fspt_target *t, float *ms, uint32_t *launches) {
  if (!t || !ms) { fspt_set_error("fspt_last_kernel_ms: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (!t->timed) { fspt_set_error("fspt_last_kernel_ms: nothing traced yet"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipEventSynchronize(t->ev1));
  HIP_TRY(hipEventElapsedTime(ms, t->ev0, t->ev1));
  if (launches) *launches = t->last_launches;
  return FSPT_OK;
}

int fspt_target_set_viewport(fspt_target *t, uint32_t w, uint32_t h) {
  if (!t) { fspt_set_error("fspt_target_set_viewport: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (w > t->W || h > t->H) { fspt_set_error("fspt_target_set_viewport: %ux%u exceeds the target %ux%u", w, h, t->W, t->H); return FSPT_E_INVALID; }
  const uint32_t nw = w ? w : t->W, nh = h ? h : t->H;
  if (nw != t->vw || nh != t->vh) prim_reset(t);
  t->vw = nw;
  t->vh = nh;
  return FSPT_OK;
}

int fspt_target_set_pipeline(fspt_target *t, int pipeline, uint32_t batch_ticks) {
  if (!t) { fspt_set_error("fspt_target_set_pipeline: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (pipeline < 0 || pipeline > 2) { fspt_set_error("pipeline must be 0 (megakernel), 1 (wavefront, batches) or 2 (wavefront, stream)"); return FSPT_E_INVALID; }
  if (batch_ticks > (uint32_t)fspt::WF_MAX_BATCH) {
    fspt_set_error("batch_ticks must be <= %d", fspt::WF_MAX_BATCH);
    return FSPT_E_INVALID;
  }
  const int sched = pipeline == 2 ? 1 : 0;
  if (pipeline != 0 && (sched != t->sched || t->stream_fallback)) {
    // the two schedulers size the path state differently: give it back (the next render allocates what it needs)
    HIP_TRY(hipSetDevice(t->scene->device));
    wf_release(t->wf);
  }
  t->pipeline = pipeline == 0 ? 0 : 1;
  if (pipeline != 0) t->sched = sched;
  t->stream_fallback = false;
  if (batch_ticks) t->batch_ticks = batch_ticks;
  return FSPT_OK;
}

int fspt_target_set_pool(fspt_target *t, uint32_t paths, int drain_iterations, uint32_t max_iterations, int overlap) {
  if (!t) { fspt_set_error("fspt_target_set_pool: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (drain_iterations < -1 || drain_iterations > FSPT_MAX_BOUNCES + 1) { fspt_set_error("fspt_target_set_pool: drain_iterations must be -1 (default) or 0..%d", FSPT_MAX_BOUNCES + 1); return FSPT_E_INVALID; }
  t->pool_paths = paths;
  t->stream_drain = drain_iterations;
  t->stream_iter_cap = max_iterations;
  t->stream_overlap = overlap < 0 ? -1 : (overlap ? 1 : 0);
  return FSPT_OK;
}

int fspt_target_set_trace_budget(fspt_target *t, uint32_t steps) {
  if (!t) { fspt_set_error("fspt_target_set_trace_budget: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->susp_budget = steps;
  return FSPT_OK;
}

int fspt_target_set_primary_form(fspt_target *t, int form) {
  if (!t) { fspt_set_error("fspt_target_set_primary_form: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (form < 0 || form > 2) { fspt_set_error("fspt_target_set_primary_form: form must be 0 (measure and choose), 1 or 2"); return FSPT_E_INVALID; }
  t->primary_form = form;
  return FSPT_OK;
}

int fspt_target_get_primary_form(fspt_target *t, uint32_t batch_ticks, int *form, double ms_per_sample[2]) {
  if (!t || !form) { fspt_set_error("fspt_target_get_primary_form: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  prim_collect(t, true);
  double m1 = -1.0, m2 = -1.0;
  const auto it = t->prim_ms.find(batch_ticks);
  if (it != t->prim_ms.end()) { m1 = it->second.best[1]; m2 = it->second.best[2]; }
  *form = (t->primary_form == 1 || t->primary_form == 2) ? t->primary_form : (int)prim_choose(t, batch_ticks);
  if (ms_per_sample) { ms_per_sample[0] = m1; ms_per_sample[1] = m2; }
  return FSPT_OK;
}

int fspt_target_set_node_form(fspt_target *t, int primary, int trace, int tail, int64_t trace_below) {
  if (!t) { fspt_set_error("fspt_target_set_node_form: NULL target"); return FSPT_E_INVALID; }
  const int v[3] = {primary, trace, tail};
  for (int k = 0; k < 3; ++k)
    if (v[k] < -1 || v[k] > (k == 2 ? 2 : 1)) { fspt_set_error("fspt_target_set_node_form: form %d (want -1, 0, 1; the tail also 2 = adaptive)", v[k]); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  for (int k = 0; k < 3; ++k) t->node_form[k] = v[k];
  if (trace_below >= 0) t->wide_trace_below = trace_below > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)trace_below;
  prim_reset(t); // the primary-form measurements were taken with the other node form
  return FSPT_OK;
}

int fspt_target_set_stage_timing(fspt_target *t, int enable) {
  if (!t) { fspt_set_error("fspt_target_set_stage_timing: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->stage_events = enable != 0;
  return FSPT_OK;
}

int fspt_target_set_tail(fspt_target *t, int round) {
  if (!t) { fspt_set_error("fspt_target_set_tail: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (round < -1 || round > FSPT_MAX_BOUNCES + 1) { fspt_set_error("fspt_target_set_tail: round must be -1 (adaptive), 0 (never) or 1..%d", FSPT_MAX_BOUNCES + 1); return FSPT_E_INVALID; }
  t->tail_round = round;
  return FSPT_OK;
}

int fspt_target_live_paths(fspt_target *t, double *frac, uint32_t n_rounds) {
  if (!t || !frac) { fspt_set_error("fspt_target_live_paths: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  if (t->wf.counts_pending) {
    HIP_TRY(hipEventSynchronize(t->wf.counts_ready));
    wf_collect_counts(t, t->wf);
  }
  for (uint32_t r = 0; r < n_rounds; ++r) frac[r] = (t->live_known && r < 80) ? (double)t->live_frac[r] : 0.0;
  return t->live_known ? FSPT_OK : FSPT_E_STATE;
}

int fspt_target_set_memory_limit(fspt_target *t, uint64_t bytes) {
  if (!t) { fspt_set_error("fspt_target_set_memory_limit: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (t->stream_fallback) { HIP_TRY(hipSetDevice(t->scene->device)); wf_release(t->wf); t->stream_fallback = false; } // what fits is decided afresh
  t->mem_limit = bytes;
  return FSPT_OK;
}

int fspt_target_path_state_bytes(fspt_target *t, uint64_t *bytes, uint32_t *batch_ticks) {
  if (!t || !bytes) { fspt_set_error("fspt_target_path_state_bytes: NULL argument"); return FSPT_E_INVALID; }
  uint64_t b = 0;
  b = t->wf.bytes + t->wf.susp_bytes;
  *bytes = b;
  if (batch_ticks) *batch_ticks = t->batch_ticks;
  return FSPT_OK;
}

int fspt_target_prepare(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_target_prepare: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  if (t->pipeline != 1) return FSPT_OK;
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  const uint64_t work_total = (uint64_t)tp.n_owned_tiles * tp.tile * tp.tile;
  if (work_total == 0) return FSPT_OK;
  if (t->sched == 0 && !t->stream_fallback) {
    uint32_t batch;
    const int rc = wf_plan_and_ensure(t, work_total, 0, batch);
    if (rc != FSPT_E_NOMEM) return rc;
    t->stream_fallback = true; // (see render_ticks)
  }
  {
    // the pool of the configured steady state: runs of batch_ticks ticks (at most WF_MAX_BATCH per run)
    const uint32_t units_total = (uint32_t)(work_total >> 6);
    const uint32_t nbt = t->batch_ticks < (uint32_t)fspt::WF_MAX_BATCH ? (t->batch_ticks ? t->batch_ticks : 1u) : (uint32_t)fspt::WF_MAX_BATCH;
    StPlan pl;
    int rc = st_plan(t, units_total, nbt, clamp_bounces(t->last_cam.num_bounces ? t->last_cam.num_bounces : 8u), pl);
    if (rc == FSPT_OK) rc = st_ensure(t, t->wf, pl.cap, pl.ring_slots, t->mem_limit ? t->mem_limit : ~0ull);
    return rc;
  }
}

int fspt_last_stage_ms(fspt_target *t, float ms[5], uint32_t launches[5]) {
  if (!t || !ms || !launches) { fspt_set_error("fspt_last_stage_ms: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipStreamSynchronize(t->stream));
  for (int k = 0; k < fspt::WF_K_KINDS; ++k) { ms[k] = 0.0f; launches[k] = 0; }
  for (uint32_t i = 0; i < t->ev_used; ++i) {
    float e = 0.0f;
    HIP_TRY(hipEventElapsedTime(&e, t->ev_pool[2 * i], t->ev_pool[2 * i + 1]));
    int k = t->ev_kind[i];
    if (k >= 0 && k < fspt::WF_K_KINDS) { ms[k] += e; launches[k]++; }
  }
  if (t->ev_overflow) { fspt_set_error("stage timing: more than %u launches, timing truncated", EV_PAIRS); return FSPT_E_STATE; }
  return FSPT_OK;
}

int fspt_enable_counters(fspt_target *t, int enable) {
  if (!t) { fspt_set_error("fspt_enable_counters: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->count = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return FSPT_OK;
}

int fspt_counters_reset(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_counters_reset: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipMemsetAsync(t->counters, 0, 64, t->stream));
  return FSPT_OK;
}

int fspt_get_counters(fspt_target *t, fspt_counters *out) {
  if (!t || !out) { fspt_set_error("fspt_get_counters: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  unsigned long long v[6];
  HIP_TRY(hipMemcpyAsync(v, t->counters, 48, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  out->samples = v[0]; out->rays = v[1]; out->steps = v[2]; out->leaves = v[3]; out->shades = v[4];
  out->env_lookups = v[5];
  return FSPT_OK;
}

int fspt_get_trace_lds_steps(fspt_target *t, uint64_t *steps) {
  if (!t || !steps) { fspt_set_error("fspt_get_trace_lds_steps: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  unsigned long long v = 0;
  HIP_TRY(hipMemcpyAsync(&v, t->counters + 6, 8, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  *steps = v;
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// stand-alone intersect + math probes
// ---------------------------------------------------------------------------
int fspt_intersect(fspt_scene *s, const float *rays, uint32_t n, float *t_out, int32_t *index_out, uint32_t *steps_out,
                   uint32_t *leaves_out) {
  return fspt_intersect_form(s, 0, rays, n, t_out, index_out, steps_out, leaves_out);
}

int fspt_scene_two_level_nodes(const fspt_scene *s, int *present, uint64_t *bytes) {
  if (!s) { fspt_set_error("fspt_scene_two_level_nodes: NULL argument"); return FSPT_E_INVALID; }
  if (present) *present = s->quads != nullptr;
  if (bytes) *bytes = s->quads ? (uint64_t)s->n_interior * fspt::QUAD_F4 * 16u : 0u;
  return FSPT_OK;
}

int fspt_intersect_form(fspt_scene *s, int two_level, const float *rays, uint32_t n, float *t_out, int32_t *index_out, uint32_t *steps_out,
                        uint32_t *leaves_out) {
  if (!s || (n && (!rays || !t_out || !index_out))) { fspt_set_error("fspt_intersect: NULL argument"); return FSPT_E_INVALID; }
  if (two_level && !s->quads) { fspt_set_error("fspt_intersect_form: the scene has no two-level nodes (its boxes are not the unions of their children's)"); return FSPT_E_INVALID; }
  if (n == 0) return FSPT_OK;
  HIP_TRY(hipSetDevice(s->device));
  float *d_rays = nullptr, *d_t = nullptr;
  int *d_i = nullptr;
  uint32_t *d_s = nullptr, *d_l = nullptr;
  int rc = FSPT_OK;
  hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 24);
  if (e == hipSuccess) e = hipMalloc((void **)&d_t, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_i, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_s, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_l, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fspt::IntersectP p{};
    p.scene = s->d; p.rays = d_rays; p.n = n; p.t_out = d_t; p.index_out = d_i; p.steps_out = d_s; p.leaves_out = d_l;
    p.wide = two_level ? 1u : 0u;
    e = fspt::launch_intersect(p, nullptr);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(t_out, d_t, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(index_out, d_i, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && steps_out) e = hipMemcpy(steps_out, d_s, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && leaves_out) e = hipMemcpy(leaves_out, d_l, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { fspt_set_error("fspt_intersect: %s", hipGetErrorString(e)); rc = FSPT_E_HIP; }
  hipFree(d_rays); hipFree(d_t); hipFree(d_i); hipFree(d_s); hipFree(d_l);
  return rc;
}

int fspt_math_eval(int device, int op, const float *a, const float *b, uint32_t n, float *out) {
  if (!a || !out) { fspt_set_error("fspt_math_eval: NULL argument"); return FSPT_E_INVALID; }
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return FSPT_OK;
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc((void **)&da, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)n * 4);
  if (e == hipSuccess && b) e = hipMalloc((void **)&db, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && b) e = hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = fspt::launch_math(op, da, db, n, dout, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  if (e != hipSuccess) { fspt_set_error("fspt_math_eval: %s", hipGetErrorString(e)); return FSPT_E_HIP; }
  return FSPT_OK;
}

} // extern "C"
