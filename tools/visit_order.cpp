// visit_order.cpp -- CPU experiment (scratch tool): how many node visits per ray the packed 8-wide tree costs under
// (A) the kernels' order: children front to back by octant, whole groups on the stack, no distance kept;
// (B) the same order with every stacked child dropped when its entry distance lies behind the hit found meanwhile;
// (C) children visited nearest first AND dropped by distance -- the least a stack traversal of this tree can do.
// Rays are the rays of short random walks from the camera through the synthetic sphere scene.
//   g++ -O2 -std=c++17 -ffp-contract=off tools/visit_order.cpp rust-pathtracer_amd/csrc/{bvh_build,scene,scene_instances}.cpp -o /tmp/visit_order -lpthread
//   /tmp/visit_order SPHERES SUBDIV [RAYS]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../rust-pathtracer_amd/csrc/bvh_build.h"
#include "../rust-pathtracer_amd/csrc/pt_bvh.h"
#include "../rust-pathtracer_amd/csrc/scene.h"

using namespace pt;

struct Counts {
  double visits = 0, items = 0, culled = 0;
};

struct ChildHit {
  float tn;
  uint32_t slot;
};

// the slab test of intersect_children, keeping every hit child's entry distance
static int children_with_distance(const NodeWords& n, const RayBox& rb, float tmin, float tmax, ChildHit out[8], uint32_t* leaf_bits) {
  const uint32_t ew = n.q[0].w;
  const float sx = u2f((ew & 0xffu) << 23) * rb.idir.x, sy = u2f(((ew >> 8) & 0xffu) << 23) * rb.idir.y,
              sz = u2f(((ew >> 16) & 0xffu) << 23) * rb.idir.z;
  const float bx = (u2f(n.q[0].x) - rb.o.x) * rb.idir.x, by = (u2f(n.q[0].y) - rb.o.y) * rb.idir.y,
              bz = (u2f(n.q[0].z) - rb.o.z) * rb.idir.z;
  const bool px = (rb.octinv & 4u) != 0u, py = (rb.octinv & 2u) != 0u, pz = (rb.octinv & 1u) != 0u;
  const uint32_t lox[2] = {n.q[2].x, n.q[2].y}, loy[2] = {n.q[2].z, n.q[2].w}, loz[2] = {n.q[3].x, n.q[3].y};
  const uint32_t hix[2] = {n.q[3].z, n.q[3].w}, hiy[2] = {n.q[4].x, n.q[4].y}, hiz[2] = {n.q[4].z, n.q[4].w};
  const uint32_t meta[2] = {n.q[1].z, n.q[1].w};
  int cnt = 0;
  *leaf_bits = 0;
  for (int half = 0; half < 2; ++half)
    for (int j = 0; j < 4; ++j) {
      const uint32_t nxw = px ? lox[half] : hix[half], fxw = px ? hix[half] : lox[half];
      const uint32_t nyw = py ? loy[half] : hiy[half], fyw = py ? hiy[half] : loy[half];
      const uint32_t nzw = pz ? loz[half] : hiz[half], fzw = pz ? hiz[half] : loz[half];
      const float tnx = byte_f(nxw, j) * sx + bx, tfx = byte_f(fxw, j) * sx + bx;
      const float tny = byte_f(nyw, j) * sy + by, tfy = byte_f(fyw, j) * sy + by;
      const float tnz = byte_f(nzw, j) * sz + bz, tfz = byte_f(fzw, j) * sz + bz;
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin)), tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax));
      if (!(tn <= tf)) continue;
      const uint32_t m = (meta[half] >> (8 * j)) & 0xffu;
      const bool inner = ((m & (m << 1)) & 0x10u) != 0u;
      if (inner) out[cnt++] = ChildHit{tn, (uint32_t)(half * 4 + j)};
      else *leaf_bits |= ((m >> 5) & 7u) << (m & 0x1fu);
    }
  return cnt;
}

static void test_items(const SceneView& sc, uint32_t base, uint32_t bits, f3 o, f3 d, float rmag, Hit& hit, Counts& c) {
  while (bits) {
    const int ti = __builtin_ctz(bits);
    bits &= bits - 1u;
    c.items++;
    const q4* tp = sc.tris + (uint32_t)kTriWords * (base + (uint32_t)ti);
    int res = fast_triangle(tp[0], tp[1], tp[2], tp[3], o, d, rmag, 0.0f, hit);
    if (res == 2) res = watertight_item(sc, tp[1], tp[2], tp[3], o, d, 0.0f, hit) ? 1 : 0;
  }
}

// (A) as trace_ray walks, counted
static Hit walk_octant(const SceneView& sc, f3 o, f3 d, Counts& c) {
  Hit hit{-1, 1e16f, 0.f, 0.f};
  RayBox rb;
  rb.o = o;
  rb.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  rb.octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
  const float rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.vmag;
  StackEntry cur{0u, 0x80000000u}, stack[64];
  int sp = 0;
  for (;;) {
    if (cur.bits & 0xff000000u) {
      const uint32_t hits_imask = cur.bits;
      const int bit = 31 - __builtin_clz(hits_imask);
      cur.bits &= ~(1u << bit);
      if (cur.bits & 0xff000000u) stack[sp++] = cur;
      const uint32_t slot = (uint32_t)(bit - 24) ^ rb.octinv;
      const uint32_t node_index = cur.base + (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
      const NodeWords n = load_node(sc.nodes, node_index);
      c.visits++;
      const uint32_t hm = intersect_children(n, rb, 0.0f, hit.t);
      cur.base = n.q[1].x;
      cur.bits = (hm & 0xff000000u) | (n.q[0].w >> 24);
      test_items(sc, n.q[1].y, hm & 0x00ffffffu, o, d, rmag, hit, c);
    }
    if ((cur.bits & 0xff000000u) == 0u) {
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
  return hit;
}

// (B) sorted = false: octant order, children stacked one by one with their entry distance; (C) sorted = true
static Hit walk_children(const SceneView& sc, f3 o, f3 d, bool sorted, Counts& c) {
  Hit hit{-1, 1e16f, 0.f, 0.f};
  RayBox rb;
  rb.o = o;
  rb.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  rb.octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
  const float rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.vmag;
  struct Entry { uint32_t node; float tn; };
  std::vector<Entry> stack;
  stack.push_back(Entry{0u, 0.0f});
  while (!stack.empty()) {
    const Entry e = stack.back();
    stack.pop_back();
    if (e.tn > hit.t) { c.culled++; continue; }
    const NodeWords n = load_node(sc.nodes, e.node);
    c.visits++;
    ChildHit ch[8];
    uint32_t leaf_bits;
    const int cnt = children_with_distance(n, rb, 0.0f, hit.t, ch, &leaf_bits);
    test_items(sc, n.q[1].y, leaf_bits, o, d, rmag, hit, c);
    const uint32_t imask = n.q[0].w >> 24;
    // far to near onto the stack: by distance, or by the octant order the kernels use (slot ^ octinv descending first)
    if (sorted) std::sort(ch, ch + cnt, [](const ChildHit& a, const ChildHit& b) { return a.tn > b.tn; });
    else std::sort(ch, ch + cnt, [&](const ChildHit& a, const ChildHit& b) { return (a.slot ^ rb.octinv) < (b.slot ^ rb.octinv); });
    for (int k = 0; k < cnt; ++k)
      stack.push_back(Entry{n.q[1].x + (uint32_t)__builtin_popcount(imask & ~(0xffffffffu << ch[k].slot)), ch[k].tn});
  }
  return hit;
}

// (B1) the kernels' walk -- whole groups on the stack, octant order -- with one number per group: the least entry distance
// of the group's children when it was made (never updated); a group is dropped when even that lies behind the hit
static Hit walk_group_min(const SceneView& sc, f3 o, f3 d, Counts& c) {
  Hit hit{-1, 1e16f, 0.f, 0.f};
  RayBox rb;
  rb.o = o;
  rb.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  rb.octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
  const float rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.vmag;
  struct Group { StackEntry e; float gmin; };
  Group cur{StackEntry{0u, 0x80000000u}, 0.0f}, stack[64];
  int sp = 0;
  for (;;) {
    if (cur.e.bits & 0xff000000u) {
      const uint32_t hits_imask = cur.e.bits;
      const int bit = 31 - __builtin_clz(hits_imask);
      cur.e.bits &= ~(1u << bit);
      if (cur.e.bits & 0xff000000u) stack[sp++] = cur;
      const uint32_t slot = (uint32_t)(bit - 24) ^ rb.octinv;
      const uint32_t node_index = cur.e.base + (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
      const NodeWords n = load_node(sc.nodes, node_index);
      c.visits++;
      ChildHit ch[8];
      uint32_t leaf_bits;
      const int cnt = children_with_distance(n, rb, 0.0f, hit.t, ch, &leaf_bits);
      uint32_t top = 0;
      float gmin = 1e30f;
      for (int k = 0; k < cnt; ++k) {
        top |= 1u << (24 + (ch[k].slot ^ rb.octinv));
        gmin = fminf(gmin, ch[k].tn);
      }
      // bfloat-like truncation towards zero: what 16 spare bits of the stack entry could hold
      gmin = u2f(f2u(gmin) & 0xffff0000u);
      cur.e.base = n.q[1].x;
      cur.e.bits = top | (n.q[0].w >> 24);
      cur.gmin = gmin;
      test_items(sc, n.q[1].y, leaf_bits, o, d, rmag, hit, c);
    }
    while ((cur.e.bits & 0xff000000u) == 0u) {
      if (sp == 0) return hit;
      cur = stack[--sp];
      if (cur.gmin > hit.t) {  // every child of the group lies behind the hit
        c.culled += __builtin_popcount(cur.e.bits >> 24);
        cur.e.bits = 0u;
      }
    }
  }
}

int main(int argc, char** argv) {
  const int spheres = argc > 1 ? atoi(argv[1]) : 500, subdiv = argc > 2 ? atoi(argv[2]) : 2;
  const int n_rays = argc > 3 ? atoi(argv[3]) : 200000;
  HostScene hs = make_synthetic(42u, spheres, subdiv);
  BvhBuildResult bvh = build_bvh8(hs.vertices.data(), hs.indices.data(), hs.num_tris());
  std::vector<ShadeRec> recs((size_t)hs.num_tris());
  for (int i = 0; i < hs.num_tris(); ++i) {
    ShadeRec& r = recs[(size_t)i];
    const uint32_t co = bvh.corner_order[(size_t)i];
    for (int k = 0; k < 3; ++k) {
      r.v0[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + (co & 3u)] + k];
      r.v1[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + ((co >> 2) & 3u)] + k];
      r.v2[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + ((co >> 4) & 3u)] + k];
    }
  }
  SceneView sc{};
  sc.nodes = bvh.nodes.data();
  sc.tris = bvh.tris.data();
  sc.shade = recs.data();
  sc.num_tris = (uint32_t)hs.num_tris();
  sc.num_items = (uint32_t)bvh.num_items;
  sc.vmag = bvh.vmag;
  printf("scene: %d triangles, %zu nodes, depth %d\n", hs.num_tris(), bvh.nodes.size(), bvh.depth);

  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  Counts a[2], b[2], c[2], g[2];  // [0] camera rays, [1] rays of later bounces
  long n[2] = {0, 0}, differ = 0;
  while (n[0] + n[1] < n_rays) {
    // a camera ray through a random pixel of the default view (eye (0,3,1) looking down -y), then a random walk
    f3 o = mk3(0.f, 3.f, 1.f);
    f3 d = normalize3(mk3((U(rng) - 0.5f) * 0.8f, -1.0f, (U(rng) - 0.5f) * 0.8f));
    for (int depth = 0; depth < 6; ++depth) {
      const int k = depth == 0 ? 0 : 1;
      const Hit h = walk_octant(sc, o, d, a[k]);
      const Hit hb = walk_children(sc, o, d, false, b[k]);
      const Hit hc = walk_children(sc, o, d, true, c[k]);
      const Hit hg = walk_group_min(sc, o, d, g[k]);
      if (h.prim != hb.prim || h.prim != hc.prim || h.prim != hg.prim) differ++;
      n[k]++;
      if (h.prim < 0) break;
      const ShadeRec& r = recs[(size_t)h.prim];
      const f3 e1 = mk3(r.v1[0] - r.v0[0], r.v1[1] - r.v0[1], r.v1[2] - r.v0[2]), e2 = mk3(r.v2[0] - r.v0[0], r.v2[1] - r.v0[1], r.v2[2] - r.v0[2]);
      f3 ng = normalize3(cross(e1, e2));
      if (dot(ng, d) > 0.f) ng = -ng;
      const f3 P = o + d * h.t;
      f3 w;
      do { w = mk3(2 * U(rng) - 1, 2 * U(rng) - 1, 2 * U(rng) - 1); } while (dot(w, w) > 1.f || dot(w, w) < 1e-4f);
      w = normalize3(w);
      if (dot(w, ng) < 0.f) w = -w;
      o = P + ng * 1e-3f;
      d = w;
    }
  }
  const char* kind[2] = {"camera rays", "bounce rays"};
  for (int k = 0; k < 2; ++k) {
    printf("%s (%ld): visits per ray  A octant order, groups %.2f | B octant order + distance cull %.2f | C nearest first + cull %.2f\n", kind[k],
           n[k], a[k].visits / n[k], b[k].visits / n[k], c[k].visits / n[k]);
    printf("   B1 groups with their least entry distance: visits %.2f, children dropped %.2f, item tests %.2f\n", g[k].visits / n[k],
           g[k].culled / n[k], g[k].items / n[k]);
    printf("   item tests per ray A %.2f  B %.2f  C %.2f; stack entries dropped by distance per ray B %.2f C %.2f\n", a[k].items / n[k],
           b[k].items / n[k], c[k].items / n[k], b[k].culled / n[k], c[k].culled / n[k]);
  }
  printf("rays whose closest primitive differs between the walks: %ld\n", differ);
  return 0;
}
