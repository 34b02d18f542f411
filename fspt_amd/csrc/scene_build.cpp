// scene_build.cpp — host-side scene pipeline of libfspt (CPU, float64).
//
// Native equivalent of the reference's JS scene pipeline, making the same
// decisions in the same IEEE binary64 arithmetic so that the packed arrays are
// byte-identical to what main.js uploads:
//   obj_loader.js:6-215   parseMesh  (v/vt/vn/f, fan triangulation, transforms,
//                                     flat/smooth/mesh normals, tangents)
//   bvh.js:5-216          BVH / Node.setSplit (full-sweep SAH on 3 pre-sorted
//                                     index lists), serializeTree (pre-order)
//   main.js:360-392       packing loops (bvh/tri/mat/norm/uv buffers)
//   main.js:272-282       maskBVHBuffer (int bits in float slots)
//   env_sampler.js:1-74   ProcessEnvRadiance (importance bins)
// Build with -ffp-contract=off: JS never fuses a*b+c.
#include "../../include/fspt.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>

void fspt_set_error(const char *fmt, ...);

namespace {

struct D3 { double x, y, z; };
inline D3 d3(double x, double y, double z) { return D3{x, y, z}; }
// vector.js:18-24,34-36,42-44
inline D3 add(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline D3 sub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline D3 scale(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vector.js:6-13
inline D3 normalize(D3 v) {
  double m = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  return scale(v, 1 / m);
}
// vector.js:112-117 (note the sign form of y)
inline D3 cross(D3 a, D3 b) {
  return d3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
// Math.min/Math.max (NaN-propagating)
inline double jsmin(double a, double b) {
  if (std::isnan(a) || std::isnan(b)) return std::numeric_limits<double>::quiet_NaN();
  return a < b ? a : b;
}
inline double jsmax(double a, double b) {
  if (std::isnan(a) || std::isnan(b)) return std::numeric_limits<double>::quiet_NaN();
  return a > b ? a : b;
}
// vector.js:86-101 rotateArbitrary + matVecMultiply
inline D3 rotate_arbitrary(D3 v, D3 axis, double angle) {
  double x = axis.x, y = axis.y, z = axis.z;
  double s = std::sin(angle), c = std::cos(angle), oc = 1.0 - c;
  double m[9] = {oc * x * x + c,     oc * x * y - z * s, oc * z * x + y * s,
                 oc * x * y + z * s, oc * y * y + c,     oc * y * z - x * s,
                 oc * z * x - y * s, oc * y * z + x * s, oc * z * z + c};
  return d3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

struct Box {
  D3 mn{INFINITY, INFINITY, INFINITY}, mx{-INFINITY, -INFINITY, -INFINITY};
  void add_vertex(D3 v) {
    mn = d3(jsmin(v.x, mn.x), jsmin(v.y, mn.y), jsmin(v.z, mn.z));
    mx = d3(jsmax(v.x, mx.x), jsmax(v.y, mx.y), jsmax(v.z, mx.z));
  }
  void add_box(const Box &b) {
    mn = d3(jsmin(mn.x, b.mn.x), jsmin(mn.y, b.mn.y), jsmin(mn.z, b.mn.z));
    mx = d3(jsmax(mx.x, b.mx.x), jsmax(mx.y, b.mx.y), jsmax(mx.z, b.mx.z));
  }
  // bvh.js:136-142
  double surface_area() const {
    double xl = mx.x - mn.x, yl = mx.y - mn.y, zl = mx.z - mn.z;
    return (xl * yl + xl * zl + yl * zl) * 2;
  }
  // bvh.js:129-134
  D3 centroid() const { return scale(add(mn, mx), 0.5); }
};

struct Tri {
  D3 v[3];
  int vi[3];
  double uv[3][2];
  bool has_uv0 = false;
  D3 n[3];
  std::vector<D3> tangents, bitangents;
  Box box;
  double mat[12];
};

double js_parse_float(const std::string &s) {
  // parseFloat: longest valid prefix, NaN when none
  const char *p = s.c_str();
  while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
  char *end = nullptr;
  // reject hex / inf forms strtod accepts but parseFloat does not
  const char *q = p;
  if (*q == '+' || *q == '-') ++q;
  if (!((*q >= '0' && *q <= '9') || *q == '.')) {
    if (std::strncmp(q, "Infinity", 8) == 0) return (*p == '-') ? -INFINITY : INFINITY;
    return std::numeric_limits<double>::quiet_NaN();
  }
  if (q[0] == '0' && (q[1] == 'x' || q[1] == 'X')) return 0.0;
  double v = std::strtod(p, &end);
  if (end == p) return std::numeric_limits<double>::quiet_NaN();
  return v;
}

// String.prototype.trim + split(/[ ]+/)
std::vector<std::string> split_spaces(const std::string &line) {
  size_t b = 0, e = line.size();
  auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; };
  while (b < e && is_ws(line[b])) ++b;
  while (e > b && is_ws(line[e - 1])) --e;
  std::vector<std::string> out;
  size_t i = b;
  std::string cur;
  while (i < e) {
    if (line[i] == ' ') {
      out.push_back(cur);
      cur.clear();
      while (i < e && line[i] == ' ') ++i;
    } else {
      cur.push_back(line[i++]);
    }
  }
  out.push_back(cur);
  return out;
}

}  // namespace

struct PendingGroup {
  std::string name;
  std::vector<Tri> tris;
  int32_t mtllib = -1;  // ordinal of the latest `mtllib` line seen when the group was created (-1: none)
};

struct fspt_builder {
  std::vector<Tri> geometry;
  // groups of the OBJ parsed last, waiting for their materials (fspt_builder_commit_obj)
  std::vector<PendingGroup> pending;
  std::vector<std::string> mtllibs;
  // scene bounds (main.js:310,317-318), for scene.normalize
  Box bounds;
  // packed output
  std::vector<float> bvh, tri, mat, norm, uv;
  uint32_t n_nodes = 0, n_tris = 0, depth = 0;
  bool built = false;
  // the built tree in float64 (for fspt_builder_autofocus): per node box, children, leaf triangle range in tri_order
  struct HostNode { Box box; int left, right; uint32_t lo, hi; bool leaf; };
  std::vector<HostNode> host_nodes;
  std::vector<uint32_t> tri_order;
};

namespace {

// ---- obj_loader.js parseMesh ------------------------------------------------
// An own property key that JS orders first, numerically (array index: canonical uint32 < 2^32 - 1)
bool js_array_index(const std::string &k, uint64_t &val) {
  if (k.empty() || k.size() > 10) return false;
  if (k.size() > 1 && k[0] == '0') return false;
  uint64_t v = 0;
  for (char c : k) { if (c < '0' || c > '9') return false; v = v * 10 + (uint64_t)(c - '0'); }
  if (v > 4294967294ull) return false;
  val = v;
  return true;
}

int parse_obj(fspt_builder *B, const char *text, size_t len, const fspt_prop_desc *prop,
              const fspt_world_transform *world, uint32_t n_world, const char *const *skips, uint32_t n_skips) {
  std::vector<D3> vertices, mesh_normals;
  std::vector<std::pair<bool, std::pair<double, double>>> uvs;  // (defined, (u,v))
  std::vector<std::vector<D3>> vert_normals;
  std::vector<std::string> group_order;
  std::map<std::string, std::vector<Tri>> groups;
  std::map<std::string, int32_t> group_mtllib;
  std::string current_group = "FSPT_DEFAULT_GROUP";
  int32_t cur_mtllib = -1;
  B->pending.clear();
  B->mtllibs.clear();
  auto skipped = [&](const std::string &g) {
    for (uint32_t i = 0; i < n_skips; ++i) if (g == skips[i]) return true;
    return false;
  };
  Box prop_bounds;  // parsed.bounds (obj_loader.js:17,137-142)

  auto apply_rotations = [&](D3 v) {
    for (uint32_t r = 0; r < prop->n_rotate; ++r) {
      const double *q = prop->rotate + 4 * r;
      v = rotate_arbitrary(v, d3(q[0], q[1], q[2]), q[3]);
    }
    return v;
  };
  // obj_loader.js:24-38
  auto apply_transforms = [&](D3 v, bool rotation_only) {
    D3 r = apply_rotations(v);
    D3 s = scale(r, rotation_only ? 1.0 : prop->scale);
    D3 t = rotation_only ? d3(0, 0, 0) : d3(prop->translate[0], prop->translate[1], prop->translate[2]);
    D3 m = add(s, t);
    for (uint32_t w = 0; w < n_world; ++w) {  // scene.worldTransforms: `if (rotate) ... else if (translate && !rotationOnly)`
      const fspt_world_transform &wt = world[w];
      if (wt.has_rotate) {
        for (uint32_t r2 = 0; r2 < wt.n_rotate; ++r2) {
          const double *q = wt.rotate + 4 * r2;
          m = rotate_arbitrary(m, d3(q[0], q[1], q[2]), q[3]);
        }
      } else if (wt.has_translate && !rotation_only) {
        m = add(m, d3(wt.translate[0], wt.translate[1], wt.translate[2]));
      }
    }
    return m;
  };

  struct Idx { double v, t, n; };
  auto parse_triangle = [&](Idx *ia, Idx *ib, Idx *ic) -> int {
    Idx *ind[3] = {ia, ib, ic};
    for (int i = 0; i < 3; ++i) {  // obj_loader.js:108-121 (in place, shared between fan triangles)
      if (ind[i]->v < 1) ind[i]->v = (double)vertices.size() + ind[i]->v + 1;
      if (ind[i]->n < 1) ind[i]->n = (double)mesh_normals.size() + ind[i]->n + 1;
    }
    Tri tri;
    for (int i = 0; i < 3; ++i) {
      double vi = ind[i]->v - 1;
      if (!(vi >= 0 && vi < (double)vertices.size()) || vi != std::floor(vi)) {
        fspt_set_error("OBJ face references vertex %g of %zu", vi + 1, vertices.size());
        return FSPT_E_PARSE;
      }
      tri.vi[i] = (int)vi;
      tri.v[i] = apply_transforms(vertices[(size_t)vi], false);
      double ti = ind[i]->t - 1;
      bool ok = (ti >= 0 && ti < (double)uvs.size() && ti == std::floor(ti) && uvs[(size_t)ti].first);
      if (i == 0) tri.has_uv0 = ok;
      tri.uv[i][0] = ok ? uvs[(size_t)ti].second.first : std::numeric_limits<double>::quiet_NaN();
      tri.uv[i][1] = ok ? uvs[(size_t)ti].second.second : std::numeric_limits<double>::quiet_NaN();
    }
    for (int i = 0; i < 3; ++i) tri.box.add_vertex(tri.v[i]);
    for (int i = 0; i < 3; ++i) {  // Vec3.max(bounds.max, vert) / Vec3.min(bounds.min, vert)
      prop_bounds.mx = d3(jsmax(prop_bounds.mx.x, tri.v[i].x), jsmax(prop_bounds.mx.y, tri.v[i].y), jsmax(prop_bounds.mx.z, tri.v[i].z));
      prop_bounds.mn = d3(jsmin(prop_bounds.mn.x, tri.v[i].x), jsmin(prop_bounds.mn.y, tri.v[i].y), jsmin(prop_bounds.mn.z, tri.v[i].z));
    }
    if (prop->normals_mode == 2) {  // "mesh", obj_loader.js:144-149
      for (int i = 0; i < 3; ++i) {
        double ni = ind[i]->n - 1;
        if (!(ni >= 0 && ni < (double)mesh_normals.size())) {
          fspt_set_error("OBJ face references normal %g of %zu", ni + 1, mesh_normals.size());
          return FSPT_E_PARSE;
        }
        tri.n[i] = normalize(apply_transforms(mesh_normals[(size_t)ni], true));
      }
    } else {  // obj_loader.js:150-159
      D3 e1 = sub(tri.v[1], tri.v[0]), e2 = sub(tri.v[2], tri.v[0]);
      D3 nrm = normalize(cross(e1, e2));
      tri.n[0] = tri.n[1] = tri.n[2] = nrm;
      for (int j = 0; j < 3; ++j) {
        size_t vi = (size_t)tri.vi[j];
        if (vert_normals.size() <= vi) vert_normals.resize(vi + 1);
        vert_normals[vi].push_back(nrm);
      }
    }
    groups[current_group].push_back(std::move(tri));
    return 0;
  };

  size_t pos = 0;
  while (pos <= len) {
    size_t nl = pos;
    while (nl < len && text[nl] != '\n') ++nl;
    std::string line(text + pos, nl - pos);
    pos = nl + 1;
    std::vector<std::string> a = split_spaces(line);
    if (a.empty()) continue;
    const std::string &key = a[0];
    if (key == "v") {
      double c[3];
      for (int i = 0; i < 3; ++i)
        c[i] = (a.size() > (size_t)(1 + i)) ? js_parse_float(a[1 + i]) : std::numeric_limits<double>::quiet_NaN();
      vertices.push_back(d3(c[0], c[1], c[2]));
    } else if (key == "f" && !skipped(current_group)) {
      // obj_loader.js:170-172: the group (and its material, from the mtllib seen so far) is created by its first face
      if (!groups.count(current_group)) {
        groups[current_group];
        group_order.push_back(current_group);
        group_mtllib[current_group] = cur_mtllib;
      }
      std::vector<Idx> fi;
      for (size_t k = 1; k < a.size(); ++k) {
        const std::string &s = a[k];
        Idx id{std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN(),
               std::numeric_limits<double>::quiet_NaN()};
        size_t p0 = 0;
        int part = 0;
        while (part < 3) {
          size_t p1 = s.find('/', p0);
          std::string tok = s.substr(p0, p1 == std::string::npos ? std::string::npos : p1 - p0);
          double v = js_parse_float(tok);
          if (part == 0) id.v = v; else if (part == 1) id.t = v; else id.n = v;
          ++part;
          if (p1 == std::string::npos) break;
          p0 = p1 + 1;
        }
        // `indices[i][j] < 1` is false for NaN: a missing vn index stays NaN
        fi.push_back(id);
      }
      // parseFace, obj_loader.js:55-61: fan; the index triples are shared between the fan's
      // triangles, so the in-place negative-index remap (NaN < 1 is false) happens once
      for (size_t i = 0; i + 2 < fi.size(); ++i) {
        int rc = parse_triangle(&fi[0], &fi[i + 1], &fi[i + 2]);
        if (rc) return rc;
      }
    } else if (key == "vt") {
      double u = (a.size() > 1) ? js_parse_float(a[1]) : std::numeric_limits<double>::quiet_NaN();
      double v = (a.size() > 2) ? js_parse_float(a[2]) : std::numeric_limits<double>::quiet_NaN();
      // `parseFloat(coord) || 0`
      if (std::isnan(u)) u = 0;
      if (std::isnan(v)) v = 0;
      bool defined = a.size() > 2;  // splice(0,2) of a 1-element list leaves v undefined -> NaN later
      uvs.push_back({true, {u, defined ? v : std::numeric_limits<double>::quiet_NaN()}});
    } else if (key == "vn") {
      double c[3];
      for (int i = 0; i < 3; ++i)
        c[i] = (a.size() > (size_t)(1 + i)) ? js_parse_float(a[1 + i]) : std::numeric_limits<double>::quiet_NaN();
      mesh_normals.push_back(d3(c[0], c[1], c[2]));
    } else if (key == "usemtl") {
      std::string name;
      for (size_t k = 1; k < a.size(); ++k) { if (k > 1) name += ' '; name += a[k]; }
      current_group = name;
    } else if (key == "mtllib") {
      // the library's text is read and resolved by the host (mtl_loader.js, getMaterial main.js:206-270);
      // the builder reports which library was current when each group was created
      std::string name;
      for (size_t k = 1; k < a.size(); ++k) { if (k > 1) name += ' '; name += a[k]; }
      B->mtllibs.push_back(name);
      cur_mtllib = (int32_t)B->mtllibs.size() - 1;
    }
  }

  // Object.entries(groups) order: array-index keys ascending, then the rest in insertion order
  {
    std::vector<std::pair<uint64_t, std::string>> numeric;
    std::vector<std::string> rest;
    for (auto &g : group_order) {
      uint64_t v;
      if (js_array_index(g, v)) numeric.push_back({v, g}); else rest.push_back(g);
    }
    std::sort(numeric.begin(), numeric.end());
    group_order.clear();
    for (auto &pr : numeric) group_order.push_back(pr.second);
    for (auto &g : rest) group_order.push_back(g);
  }

  // smooth normals, obj_loader.js:194-203 (average is NOT re-normalised)
  if (prop->normals_mode == 1) {
    for (auto &gname : group_order)
      for (auto &t : groups[gname])
        for (int j = 0; j < 3; ++j) {
          const std::vector<D3> &arr = vert_normals[(size_t)t.vi[j]];
          D3 total = d3(0, 0, 0);
          for (const D3 &q : arr) total = add(total, q);
          t.n[j] = scale(total, 1.0 / (double)arr.size());
        }
  }

  // calcTangents, obj_loader.js:63-103
  const double EPS = 2.220446049250313e-16;  // Number.EPSILON
  for (auto &gname : group_order)
    for (auto &t : groups[gname]) {
      if (!t.has_uv0) {
        for (int i = 0; i < 3; ++i) {
          D3 dir = normalize(t.v[i]);
          t.uv[i][0] = std::atan2(dir.z, dir.x) / (M_PI * 2);
          t.uv[i][1] = std::asin(-dir.y) / M_PI + 0.5;
        }
      }
      for (int i = 0; i < 3; ++i) {
        t.uv[i][0] += EPS * (i + 1);
        t.uv[i][1] += EPS * (i + 1);
      }
      D3 dp0 = sub(t.v[1], t.v[0]), dp1 = sub(t.v[2], t.v[0]);
      double du0[2] = {t.uv[1][0] - t.uv[0][0], t.uv[1][1] - t.uv[0][1]};
      double du1[2] = {t.uv[2][0] - t.uv[0][0], t.uv[2][1] - t.uv[0][1]};
      double r = 1.0 / ((du0[0] * du1[1]) - (du0[1] * du1[0]));
      D3 pre_tangent = normalize(scale(sub(scale(dp0, du1[1]), scale(dp1, du0[1])), r));
      auto assign = [](std::vector<D3> &arr, size_t i, D3 v) {
        if (arr.size() <= i) arr.resize(i + 1, d3(NAN, NAN, NAN));
        arr[i] = v;
      };
      for (int i = 0; i < 3; ++i) {
        D3 nrm = t.n[i];
        D3 pre_bt = normalize(cross(nrm, pre_tangent));
        D3 tangent = normalize(cross(pre_bt, nrm));
        D3 bitangent = normalize(cross(nrm, tangent));
        if (std::isnan(dot(tangent, bitangent))) {  // obj_loader.js:95-99 (then still pushes)
          D3 tt = cross(t.n[i], d3(0, 1, 0));
          assign(t.tangents, (size_t)i, tt);
          assign(t.bitangents, (size_t)i, cross(tt, t.n[i]));
        }
        t.tangents.push_back(tangent);
        t.bitangents.push_back(bitangent);
      }
    }

  for (auto &gname : group_order) {
    PendingGroup pg;
    pg.name = gname;
    pg.tris = std::move(groups[gname]);
    pg.mtllib = group_mtllib[gname];
    B->pending.push_back(std::move(pg));
  }
  // main.js:317-318: bounds.addVertex(parsed.bounds.max); bounds.addVertex(parsed.bounds.min)
  B->bounds.add_vertex(prop_bounds.mx);
  B->bounds.add_vertex(prop_bounds.mn);
  return 0;
}

// main.js:326-334 + 376-382: every triangle of a group gets the group's material record
void commit_groups(fspt_builder *B, const fspt_group_material *mats) {
  for (size_t g = 0; g < B->pending.size(); ++g) {
    const fspt_group_material &gm = mats[g];
    double m[12] = {gm.diffuse_layer, gm.emissive_layer, gm.normal_layer, gm.mr_layer, 0, 0,
                    gm.emittance[0], gm.emittance[1], gm.emittance[2], gm.ior, gm.dielectric, 0};
    for (auto &t : B->pending[g].tris) {
      std::memcpy(t.mat, m, sizeof(m));
      B->geometry.push_back(std::move(t));
    }
  }
  B->pending.clear();
}

// ---- bvh.js --------------------------------------------------------------
struct BuildNode {
  Box box;
  int left = -1, right = -1;  // indices into nodes (pre-order)
  uint32_t lo = 0, hi = 0;    // range in the axis-0 index list (leaf triangles = idx[0][lo..hi))
  bool leaf = false;
};

struct BvhBuilder {
  const std::vector<Tri> &tris;
  uint32_t max_tris;
  std::vector<uint32_t> idx[3];
  std::vector<uint32_t> tmp;
  std::vector<uint8_t> mark;
  std::vector<double> sf, sb;
  std::vector<BuildNode> nodes;
  uint32_t depth = 0;
  bool failed = false;

  BvhBuilder(const std::vector<Tri> &t, uint32_t m) : tris(t), max_tris(m) {}

  void build() {
    size_t n = tris.size();
    std::vector<D3> cent(n);
    for (size_t i = 0; i < n; ++i) cent[i] = tris[i].box.centroid();
    for (int a = 0; a < 3; ++a) {
      idx[a].resize(n);
      for (size_t i = 0; i < n; ++i) idx[a][i] = (uint32_t)i;
      // Array.prototype.sort is a stable TimSort in V8 >= 7.0 (Node 12); bvh.js:78-90
      std::stable_sort(idx[a].begin(), idx[a].end(), [&](uint32_t i1, uint32_t i2) {
        double c1 = a == 0 ? cent[i1].x : (a == 1 ? cent[i1].y : cent[i1].z);
        double c2 = a == 0 ? cent[i2].x : (a == 1 ? cent[i2].y : cent[i2].z);
        return c1 < c2;
      });
    }
    tmp.resize(n);
    mark.assign(n, 0);
    sf.resize(n);
    sb.resize(n);
    build_tree(0, (uint32_t)n, 0);
  }

  // bvh.js:19-31 buildTree + Node ctor + setSplit (168-197); returns node index
  int build_tree(uint32_t lo, uint32_t hi, uint32_t d) {
    if (failed) return -1;
    depth = std::max(depth, d);
    int me = (int)nodes.size();
    nodes.emplace_back();
    uint32_t n = hi - lo;
    Box box;
    for (uint32_t i = lo; i < hi; ++i) {  // addNode, bvh.js:122-128
      const Tri &t = tris[idx[0][i]];
      box.add_vertex(t.v[0]); box.add_vertex(t.v[1]); box.add_vertex(t.v[2]);
    }
    nodes[me].box = box;
    nodes[me].lo = lo; nodes[me].hi = hi;
    // setSplit
    double best = INFINITY;
    double parent_sa = box.surface_area();
    int split_axis = -1; uint32_t split_index = 0;
    for (int axis = 0; axis < 3; ++axis) {
      const uint32_t *ic = idx[axis].data() + lo;
      Box bf, bb;
      for (uint32_t i = 0; i < n; ++i) {
        bf.add_box(tris[ic[i]].box);
        bb.add_box(tris[ic[n - 1 - i]].box);
        sf[i] = bf.surface_area();
        sb[i] = bb.surface_area();
      }
      for (uint32_t i = 0; i < n; ++i) {
        double sAf = sf[i], sAb = sb[n - 1 - i];
        double cost = 1 + (sAf / parent_sa) * 1 * (double)(i + 1) + (sAb / parent_sa) * 1 * (double)(n - 1 - i);
        if (cost < best) { best = cost; split_index = i + 1; split_axis = axis; }
      }
    }
    if (n <= max_tris) {  // bvh.js:22 (uses indices[splitAxis || 0].length == n)
      nodes[me].leaf = true;
      return me;
    }
    if (split_axis < 0 || split_index == 0 || split_index >= n) {
      // the JS recurses forever here (NaN costs / degenerate boxes)
      failed = true;
      return me;
    }
    // _constructCachedIndexList, bvh.js:52-76: stable partition of the other two lists
    for (uint32_t i = 0; i < split_index; ++i) mark[idx[split_axis][lo + i]] = 1;
    for (int axis = 0; axis < 3; ++axis) {
      if (axis == split_axis) continue;
      uint32_t *ic = idx[axis].data() + lo;
      uint32_t li = 0, ri = 0;
      for (uint32_t j = 0; j < n; ++j) {
        uint32_t v = ic[j];
        if (mark[v]) ic[li++] = v; else tmp[ri++] = v;
      }
      std::memcpy(ic + li, tmp.data(), ri * sizeof(uint32_t));
    }
    for (uint32_t i = 0; i < split_index; ++i) mark[idx[split_axis][lo + i]] = 0;
    int l = build_tree(lo, lo + split_index, d + 1);
    int r = build_tree(lo + split_index, hi, d + 1);
    nodes[me].left = l; nodes[me].right = r;
    return me;
  }
};

}  // namespace

extern "C" {

int fspt_builder_create(fspt_builder **out) {
  if (!out) { fspt_set_error("fspt_builder_create: out is NULL"); return FSPT_E_INVALID; }
  *out = new fspt_builder();
  return FSPT_OK;
}

int fspt_builder_destroy(fspt_builder *b) {
  delete b;
  return FSPT_OK;
}

int fspt_builder_parse_obj(fspt_builder *b, const char *obj_text, size_t len, const fspt_prop_desc *prop,
                           const fspt_world_transform *world, uint32_t n_world, const char *const *skips,
                           uint32_t n_skips, uint32_t *n_groups) {
  if (!b || !obj_text || !prop) { fspt_set_error("fspt_builder_parse_obj: NULL argument"); return FSPT_E_INVALID; }
  if (prop->n_rotate && !prop->rotate) { fspt_set_error("fspt_builder_parse_obj: rotate is NULL"); return FSPT_E_INVALID; }
  if ((n_world && !world) || (n_skips && !skips)) { fspt_set_error("fspt_builder_parse_obj: NULL world / skips"); return FSPT_E_INVALID; }
  for (uint32_t w = 0; w < n_world; ++w)
    if (world[w].has_rotate && world[w].n_rotate && !world[w].rotate) { fspt_set_error("fspt_builder_parse_obj: world rotate is NULL"); return FSPT_E_INVALID; }
  if (!b->pending.empty()) { fspt_set_error("fspt_builder_parse_obj: the previous OBJ was not committed"); return FSPT_E_STATE; }
  b->built = false;
  int rc = parse_obj(b, obj_text, len, prop, world, n_world, skips, n_skips);
  if (rc) { b->pending.clear(); return rc; }
  if (n_groups) *n_groups = (uint32_t)b->pending.size();
  return FSPT_OK;
}

int fspt_builder_group_info(const fspt_builder *b, uint32_t group, const char **name, uint32_t *n_tris, int32_t *mtllib) {
  if (!b || group >= b->pending.size()) { fspt_set_error("fspt_builder_group_info: no such pending group"); return FSPT_E_INVALID; }
  if (name) *name = b->pending[group].name.c_str();
  if (n_tris) *n_tris = (uint32_t)b->pending[group].tris.size();
  if (mtllib) *mtllib = b->pending[group].mtllib;
  return FSPT_OK;
}

int fspt_builder_mtllib_name(const fspt_builder *b, uint32_t index, const char **name) {
  if (!b || !name || index >= b->mtllibs.size()) { fspt_set_error("fspt_builder_mtllib_name: no such mtllib"); return FSPT_E_INVALID; }
  *name = b->mtllibs[index].c_str();
  return FSPT_OK;
}

int fspt_builder_commit_obj(fspt_builder *b, const fspt_group_material *mats, uint32_t n_groups) {
  if (!b) { fspt_set_error("fspt_builder_commit_obj: NULL builder"); return FSPT_E_INVALID; }
  if (n_groups != b->pending.size()) { fspt_set_error("fspt_builder_commit_obj: %u materials for %zu pending groups", n_groups, b->pending.size()); return FSPT_E_INVALID; }
  if (n_groups && !mats) { fspt_set_error("fspt_builder_commit_obj: mats is NULL"); return FSPT_E_INVALID; }
  commit_groups(b, mats);
  return FSPT_OK;
}

int fspt_builder_add_obj(fspt_builder *b, const char *obj_text, size_t len, const fspt_prop_desc *prop) {
  uint32_t ng = 0;
  int rc = fspt_builder_parse_obj(b, obj_text, len, prop, nullptr, 0, nullptr, 0, &ng);
  if (rc) return rc;
  fspt_group_material gm;
  gm.diffuse_layer = prop->diffuse_layer; gm.emissive_layer = prop->emissive_layer;
  gm.normal_layer = prop->normal_layer; gm.mr_layer = prop->mr_layer;
  gm.emittance[0] = prop->emittance[0]; gm.emittance[1] = prop->emittance[1]; gm.emittance[2] = prop->emittance[2];
  gm.ior = prop->ior; gm.dielectric = prop->dielectric;
  std::vector<fspt_group_material> mats(ng, gm);
  return fspt_builder_commit_obj(b, mats.data(), ng);
}

// main.js:337-348: `scene.normalize`: centre the scene and scale its longest side to 2 * size.  Only the vertices
// move; the per-triangle boxes the SAH sweeps and the centroid sort use stay as built (bvh.js:204), as in the reference.
int fspt_builder_normalize(fspt_builder *b, double size) {
  if (!b) { fspt_set_error("fspt_builder_normalize: NULL builder"); return FSPT_E_INVALID; }
  if (!b->pending.empty()) { fspt_set_error("fspt_builder_normalize: an OBJ is waiting for fspt_builder_commit_obj"); return FSPT_E_STATE; }
  D3 diff = sub(b->bounds.mx, b->bounds.mn);
  double longest = jsmax(jsmax(diff.x, diff.y), diff.z);
  D3 centroid = b->bounds.centroid();
  double sc = 2 * size / longest;
  for (auto &t : b->geometry)
    for (int j = 0; j < 3; ++j) t.v[j] = scale(sub(t.v[j], centroid), sc);
  b->built = false;
  return FSPT_OK;
}

int fspt_builder_build(fspt_builder *b, uint32_t leaf_size) {
  if (!b) { fspt_set_error("fspt_builder_build: NULL builder"); return FSPT_E_INVALID; }
  if (leaf_size == 0) { fspt_set_error("fspt_builder_build: leaf_size must be >= 1"); return FSPT_E_INVALID; }
  if (!b->pending.empty()) { fspt_set_error("fspt_builder_build: an OBJ is waiting for fspt_builder_commit_obj"); return FSPT_E_STATE; }
  if (b->geometry.empty()) { fspt_set_error("fspt_builder_build: no triangles"); return FSPT_E_INVALID; }
  BvhBuilder bb(b->geometry, leaf_size);
  bb.build();
  if (bb.failed) {
    fspt_set_error("fspt_builder_build: SAH found no valid split (degenerate geometry); bvh.js would not terminate");
    return FSPT_E_INVALID;
  }
  // packing loops, main.js:360-392.  Node order is already pre-order (serializeTree, bvh.js:33-50).
  size_t nn = bb.nodes.size(), nt = b->geometry.size();
  b->bvh.assign(nn * 9, 0.0f);
  b->tri.clear(); b->mat.clear(); b->norm.clear(); b->uv.clear();
  b->tri.reserve(nt * 9); b->mat.reserve(nt * 12); b->norm.reserve(nt * 27); b->uv.reserve(nt * 6);
  for (size_t i = 0; i < nn; ++i) {
    const BuildNode &nd = bb.nodes[i];
    int32_t w[3];
    // leaf children are JS `undefined` -> Int32Array 0 (maskBVHBuffer, main.js:272-282)
    w[0] = nd.leaf ? 0 : nd.left;
    w[1] = nd.leaf ? 0 : nd.right;
    w[2] = nd.leaf ? (int32_t)(b->tri.size() / 9) : -1;
    std::memcpy(&b->bvh[i * 9], w, 12);
    b->bvh[i * 9 + 3] = (float)nd.box.mn.x; b->bvh[i * 9 + 4] = (float)nd.box.mn.y; b->bvh[i * 9 + 5] = (float)nd.box.mn.z;
    b->bvh[i * 9 + 6] = (float)nd.box.mx.x; b->bvh[i * 9 + 7] = (float)nd.box.mx.y; b->bvh[i * 9 + 8] = (float)nd.box.mx.z;
    if (nd.leaf) {
      for (uint32_t k = nd.lo; k < nd.hi; ++k) {  // getTriangles = indices[0] order
        const Tri &t = b->geometry[bb.idx[0][k]];
        for (int v = 0; v < 3; ++v) {
          b->tri.push_back((float)t.v[v].x); b->tri.push_back((float)t.v[v].y); b->tri.push_back((float)t.v[v].z);
        }
        for (int q = 0; q < 12; ++q) b->mat.push_back((float)t.mat[q]);
        for (int v = 0; v < 3; ++v) {
          D3 tg = t.tangents.size() > (size_t)v ? t.tangents[v] : d3(NAN, NAN, NAN);
          D3 bt = t.bitangents.size() > (size_t)v ? t.bitangents[v] : d3(NAN, NAN, NAN);
          b->norm.push_back((float)t.n[v].x); b->norm.push_back((float)t.n[v].y); b->norm.push_back((float)t.n[v].z);
          b->norm.push_back((float)tg.x); b->norm.push_back((float)tg.y); b->norm.push_back((float)tg.z);
          b->norm.push_back((float)bt.x); b->norm.push_back((float)bt.y); b->norm.push_back((float)bt.z);
        }
        for (int v = 0; v < 3; ++v) { b->uv.push_back((float)t.uv[v][0]); b->uv.push_back((float)t.uv[v][1]); }
      }
    }
  }
  b->n_nodes = (uint32_t)nn;
  b->n_tris = (uint32_t)nt;
  b->depth = bb.depth;
  b->host_nodes.resize(nn);
  for (size_t i = 0; i < nn; ++i) {
    const BuildNode &nd = bb.nodes[i];
    b->host_nodes[i] = fspt_builder::HostNode{nd.box, nd.left, nd.right, nd.lo, nd.hi, nd.leaf};
  }
  b->tri_order = bb.idx[0];
  b->built = true;
  return FSPT_OK;
}

// shootAutoFocusRay (main.js:447-546): the distance along (eye, dir) to the first triangle, found on the host
// BVH in float64; main.js then sets lensFeatures[0] = 1 - 1 / dist.  1e6 (maxT) when nothing is hit.
int fspt_builder_autofocus(const fspt_builder *b, const double eye_[3], const double dir_[3], double *dist) {
  if (!b || !b->built) { fspt_set_error("fspt_builder_autofocus: builder not built"); return FSPT_E_STATE; }
  if (!eye_ || !dir_ || !dist) { fspt_set_error("fspt_builder_autofocus: NULL argument"); return FSPT_E_INVALID; }
  const double maxT = 1e6;
  const D3 eye = d3(eye_[0], eye_[1], eye_[2]), dir = d3(dir_[0], dir_[1], dir_[2]);
  auto ray_tri = [&](const Tri &tri) -> double {  // main.js:448-473
    const double epsilon = 0.000000000001;
    D3 e1 = sub(tri.v[1], tri.v[0]), e2 = sub(tri.v[2], tri.v[0]);
    D3 p = cross(dir, e2);
    double det = dot(e1, p);
    if (det > -epsilon && det < epsilon) return maxT;
    double invDet = 1.0 / det;
    D3 t = sub(eye, tri.v[0]);
    double u = dot(t, p) * invDet;
    if (u < 0 || u > 1) return maxT;
    D3 q = cross(t, e1);
    double v = dot(dir, q) * invDet;
    if (v < 0 || u + v > 1) return maxT;
    double tt = dot(e2, q) * invDet;
    if (tt > epsilon) return tt;
    return maxT;
  };
  auto ray_box = [&](const Box &bx) -> double {  // main.js:487-504
    D3 inv = d3(1 / dir.x, 1 / dir.y, 1 / dir.z);  // Vec3.inverse
    double tx1 = (bx.mn.x - eye.x) * inv.x, tx2 = (bx.mx.x - eye.x) * inv.x;
    double ty1 = (bx.mn.y - eye.y) * inv.y, ty2 = (bx.mx.y - eye.y) * inv.y;
    double tz1 = (bx.mn.z - eye.z) * inv.z, tz2 = (bx.mx.z - eye.z) * inv.z;
    double tmin = jsmin(tx1, tx2), tmax = jsmax(tx1, tx2);
    tmin = jsmax(tmin, jsmin(ty1, ty2)); tmax = jsmin(tmax, jsmax(ty1, ty2));
    tmin = jsmax(tmin, jsmin(tz1, tz2)); tmax = jsmin(tmax, jsmax(tz1, tz2));
    return (tmax >= tmin && tmax >= 0) ? tmin : maxT;
  };
  // findTriangles (main.js:531-542), explicit stack instead of recursion; `closest` is threaded through
  struct Frame { int node; int stage; int ord[2]; double t[2]; };
  std::vector<Frame> st;
  double closest = maxT;
  // recursion returns a value that is min-ed into the caller's `closest`; since every callee starts from the
  // caller's current `closest` and only lowers it, one running minimum is equivalent
  st.push_back(Frame{0, 0, {-1, -1}, {0, 0}});
  while (!st.empty()) {
    Frame &f = st.back();
    const fspt_builder::HostNode &nd = b->host_nodes[(size_t)f.node];
    if (nd.leaf) {
      double res = maxT;  // processLeaf: the leaf's closest hit, then Math.min(res, closest)
      for (uint32_t k = nd.lo; k < nd.hi; ++k) {
        double tmp = ray_tri(b->geometry[b->tri_order[k]]);
        if (tmp < res) res = tmp;
      }
      closest = jsmin(res, closest);
      st.pop_back();
      continue;
    }
    if (f.stage == 0) {  // closestNode (main.js:506-529)
      double tl = ray_box(b->host_nodes[(size_t)nd.left].box), tr = ray_box(b->host_nodes[(size_t)nd.right].box);
      int left = tl < maxT ? nd.left : -1, right = tr < maxT ? nd.right : -1;
      if (tl < tr) { f.ord[0] = left; f.t[0] = tl; f.ord[1] = right; f.t[1] = tr; }
      else { f.ord[0] = right; f.t[0] = tr; f.ord[1] = left; f.t[1] = tl; }
    }
    if (f.stage >= 2) { st.pop_back(); continue; }
    int i = f.stage++;
    if (f.ord[i] >= 0 && f.t[i] < closest) {
      int child = f.ord[i];
      st.push_back(Frame{child, 0, {-1, -1}, {0, 0}});  // invalidates f; not used after this point
    }
  }
  *dist = closest;
  return FSPT_OK;
}

int fspt_builder_counts(const fspt_builder *b, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *depth) {
  if (!b || !b->built) { fspt_set_error("fspt_builder_counts: builder not built"); return FSPT_E_STATE; }
  if (n_nodes) *n_nodes = b->n_nodes;
  if (n_tris) *n_tris = b->n_tris;
  if (depth) *depth = b->depth;
  return FSPT_OK;
}

int fspt_builder_get(const fspt_builder *b, float *bvh, float *tri, float *mat, float *norm, float *uv) {
  if (!b || !b->built) { fspt_set_error("fspt_builder_get: builder not built"); return FSPT_E_STATE; }
  if (bvh) std::memcpy(bvh, b->bvh.data(), b->bvh.size() * 4);
  if (tri) std::memcpy(tri, b->tri.data(), b->tri.size() * 4);
  if (mat) std::memcpy(mat, b->mat.data(), b->mat.size() * 4);
  if (norm) std::memcpy(norm, b->norm.data(), b->norm.size() * 4);
  if (uv) std::memcpy(uv, b->uv.data(), b->uv.size() * 4);
  return FSPT_OK;
}

// ---- env_sampler.js ProcessEnvRadiance ------------------------------------
namespace {
struct EnvCtx {
  const uint8_t *data; uint32_t w, h;
  double min_radiance;
  std::vector<double> boxes;
  double radiance_at(double x, double y) const {  // getRadiance + pixelAt, env_sampler.js:6-21
    double base = (y * ((double)w * 4)) + (x * 4);
    double c[4];
    for (int i = 0; i < 4; ++i) {
      double ix = base + i;
      // typed-array read at a non-integer / out-of-range index is `undefined` -> NaN arithmetic
      if (!(ix >= 0 && ix < (double)w * h * 4) || ix != std::floor(ix)) c[i] = NAN;
      else c[i] = (double)data[(size_t)ix];
    }
    double power = std::pow(2.0, c[3] - 128);
    double n0 = power * c[0] / 255.0, n1 = power * c[1] / 255.0, n2 = power * c[2] / 255.0;
    return 0.2126 * n0 + 0.7152 * n1 + 0.0722 * n2;
  }
  void bisplit(double radiance, double x0, double y0, double x1, double y1, int depth) {
    // env_sampler.js:25-47
    if (radiance <= min_radiance || (y1 - y0) * (x1 - x0) < 2 || depth > 64) {
      boxes.push_back(x0); boxes.push_back(y0); boxes.push_back(x1); boxes.push_back(y1);
      return;
    }
    double sub = 0;
    bool vert = (x1 - x0) > (y1 - y0);
    double xs = x1, ys = (y1 - y0) / 2 + y0;
    if (vert) { xs = (x1 - x0) / 2 + x0; ys = y1; }
    for (double x = x0; x < xs; x++)
      for (double y = y0; y < ys; y++) sub += radiance_at(x, y);
    bisplit(sub, x0, y0, xs, ys, depth + 1);
    if (vert) bisplit(radiance - sub, xs, y0, x1, y1, depth + 1);
    else bisplit(radiance - sub, x0, ys, x1, y1, depth + 1);
  }
};
}  // namespace

int fspt_env_bins(const uint8_t *rgbe, uint32_t w, uint32_t h, uint32_t *bins, uint32_t cap, uint32_t *n_bins) {
  if (!rgbe || !w || !h || !n_bins) { fspt_set_error("fspt_env_bins: NULL/empty argument"); return FSPT_E_INVALID; }
  EnvCtx ctx{rgbe, w, h, 0.0, {}};
  double total = 0, brightest = 0;
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      double rad = ctx.radiance_at((double)x, (double)y);
      brightest = jsmax(rad, brightest);
      total += rad;
    }
  ctx.min_radiance = jsmax(total / 64, brightest / 2);
  ctx.bisplit(total, 0, 0, (double)w, (double)h, 0);
  uint32_t nb = (uint32_t)(ctx.boxes.size() / 4);
  *n_bins = nb;
  if (bins) {
    uint32_t m = nb < cap ? nb : cap;
    for (uint32_t i = 0; i < m * 4; ++i) {
      double v = ctx.boxes[i];
      // new Uint16Array(boxes): ToUint16 (truncate, modulo 2^16; NaN -> 0)
      uint32_t u = 0;
      if (std::isfinite(v)) { double tr = std::trunc(v); double mo = std::fmod(tr, 65536.0); if (mo < 0) mo += 65536.0; u = (uint32_t)mo; }
      bins[i] = u;
    }
  }
  return FSPT_OK;
}

}  // extern "C"
