// bvh_build.cpp -- binned-SAH binary build, greedy collapse to 8-wide nodes, octant-ordered
// slot assignment and 8-bit box quantisation.
#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>

namespace pt {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  void grow(const float* p) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
  }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Node2 {
  Box box;
  int left = -1, right = -1;  // internal
  int first = 0, count = 0;   // leaf when count > 0
};

constexpr int kBins = 16;
constexpr int kMaxLeaf = 3;

struct Builder {
  const float* verts;
  const uint32_t* idx;
  std::vector<Box> tri_box;
  std::vector<float> centroid;  // 3 per triangle
  std::vector<uint32_t> order;  // triangle permutation
  std::vector<Node2> nodes2;
  // Optimal 2-ary -> 8-ary collapse (Ylitie, Karras, Laine 2017, section 3.1): cost[n][i] is
  // the SAH cost of representing the subtree of n by a forest of at most i+1 slot entries.
  struct Collapse {
    float cost[8];
    uint8_t split[8];  // for i >= 1: slots given to the left child, 0 = same as i-1
    bool internal;     // i == 0: n becomes an internal 8-wide node (else a leaf entry)
  };
  std::vector<Collapse> dp;

  void compute_collapse(int n) {
    const float c_node = 1.0f, c_prim = 0.3f;
    Node2& nd = nodes2[n];
    Collapse& c = dp[(size_t)n];
    const float area = nd.box.half_area();
    if (nd.count > 0) {
      for (int i = 0; i < 8; ++i) { c.cost[i] = area * c_prim * (float)nd.count; c.split[i] = 0; }
      c.internal = false;
      return;
    }
    compute_collapse(nd.left);
    compute_collapse(nd.right);
    const Collapse& l = dp[(size_t)nd.left];
    const Collapse& r = dp[(size_t)nd.right];
    float dist[8];  // dist[i]: best cost of spreading the two children over i+1 slots (i >= 1)
    uint8_t dsplit[8];
    for (int i = 1; i < 8; ++i) {
      dist[i] = FLT_MAX;
      dsplit[i] = 1;
      for (int k = 1; k <= i; ++k) {  // k slots left, i+1-k right
        const float v = l.cost[k - 1] + r.cost[i - k];
        if (v < dist[i]) { dist[i] = v; dsplit[i] = (uint8_t)k; }
      }
    }
    c.cost[0] = area * c_node + dist[7];
    c.internal = true;
    c.split[0] = 0;
    for (int i = 1; i < 8; ++i) {
      if (dist[i] < c.cost[i - 1]) { c.cost[i] = dist[i]; c.split[i] = dsplit[i]; }
      else { c.cost[i] = c.cost[i - 1]; c.split[i] = 0; }
    }
  }
  // roots of the forest that represents subtree n in at most i+1 slots
  void forest(int n, int i, int* out, int& count) const {
    const Node2& nd = nodes2[n];
    if (nd.count > 0) { out[count++] = n; return; }
    while (i > 0 && dp[(size_t)n].split[i] == 0) --i;
    if (i == 0) { out[count++] = n; return; }
    const int k = dp[(size_t)n].split[i];
    forest(nd.left, k - 1, out, count);
    forest(nd.right, i - k, out, count);
  }

  int build2(int first, int count) {
    Node2 n;
    n.box.reset();
    Box cb;
    cb.reset();
    for (int i = first; i < first + count; ++i) {
      n.box.grow(tri_box[order[i]]);
      cb.grow(&centroid[3 * (size_t)order[i]]);
    }
    const int id = (int)nodes2.size();
    nodes2.push_back(n);
    if (count == 1) {
      nodes2[id].first = first;
      nodes2[id].count = count;
      return id;
    }
    // best binned SAH split over the three axes
    float best_cost = FLT_MAX;
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
      const float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
      if (!(ext > 0.f)) continue;
      Box bb[kBins];
      int bc[kBins];
      for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
      const float scale = (float)kBins / ext;
      for (int i = first; i < first + count; ++i) {
        const uint32_t t = order[i];
        int b = (int)((centroid[3 * (size_t)t + axis] - lo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        bb[b].grow(tri_box[t]);
        bc[b]++;
      }
      float right_area[kBins];
      int right_cnt[kBins];
      Box acc;
      acc.reset();
      int c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        acc.grow(bb[b]);
        c += bc[b];
        right_area[b] = acc.half_area();
        right_cnt[b] = c;
      }
      acc.reset();
      c = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        acc.grow(bb[b]);
        c += bc[b];
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
      }
    }
    const float parent_area = nodes2[id].box.half_area();
    if (count <= kMaxLeaf) {
      // leaf cost: count triangle tests; split cost: one more node level plus the children
      const float leaf_cost = (float)count * parent_area;
      const float split_cost = best_axis < 0 ? FLT_MAX : 1.0f * parent_area + best_cost;
      if (leaf_cost <= split_cost) {
        nodes2[id].first = first;
        nodes2[id].count = count;
        return id;
      }
    }
    int mid;
    if (best_axis < 0) {
      mid = first + count / 2;  // all centroids coincide: split by position in the list
    } else {
      const float lo = cb.lo[best_axis], scale = (float)kBins / (cb.hi[best_axis] - cb.lo[best_axis]);
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
        int b = (int)((centroid[3 * (size_t)t + best_axis] - lo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= best_bin;
      });
      mid = (int)(it - order.begin());
      if (mid == first || mid == first + count) mid = first + count / 2;
    }
    const int l = build2(first, mid - first);
    const int r = build2(mid, first + count - mid);
    nodes2[id].left = l;
    nodes2[id].right = r;
    return id;
  }
};

struct Emitter {
  const Builder& b;
  BvhBuildResult& out;
  uint32_t tri_cursor = 0;

  // traversal record of a triangle (layout: TriRecord comment in pt_bvh.h), computed in double
  void write_tri(uint32_t slot_index, uint32_t prim) {
    const uint32_t* t = b.idx + 3 * (size_t)prim;
    double v[3][3];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) v[k][c] = (double)b.verts[3 * (size_t)t[k] + c];
    const double e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
    const double e2[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    q4* q = &out.tris[4 * (size_t)slot_index];
    q[0] = q4{(float)v[0][0], (float)v[0][1], (float)v[0][2], 0.f};
    q[1] = q4{0.f, 0.f, 0.f, u2f(prim)};
    q[2] = q4{0.f, 0.f, 0.f, 0.f};
    q[3] = q4{0.f, 0.f, 0.f, 0.f};
    if (!(nn > 0.0) || !std::isfinite(nn)) return;  // degenerate: a zero normal makes t = nan, never hit
    const double inv = 1.0 / nn, invlen = 1.0 / std::sqrt(nn);
    // r1 = (e2 x n) / n.n, r2 = (n x e1) / n.n
    const double r1[3] = {(e2[1] * n[2] - e2[2] * n[1]) * inv, (e2[2] * n[0] - e2[0] * n[2]) * inv, (e2[0] * n[1] - e2[1] * n[0]) * inv};
    const double r2[3] = {(n[1] * e1[2] - n[2] * e1[1]) * inv, (n[2] * e1[0] - n[0] * e1[2]) * inv, (n[0] * e1[1] - n[1] * e1[0]) * inv};
    const double norm1 = std::fabs(r1[0]) + std::fabs(r1[1]) + std::fabs(r1[2]);
    const double norm2 = std::fabs(r2[0]) + std::fabs(r2[1]) + std::fabs(r2[2]);
    // 16 ulps of the largest term of the dot products per unit of coordinate magnitude, both rows
    const double eps = std::ldexp(norm1 + norm2, -20);
    q[0].w = (float)eps;
    q[1].x = (float)r1[0]; q[1].y = (float)r1[1]; q[1].z = (float)r1[2];
    q[2].x = (float)r2[0]; q[2].y = (float)r2[1]; q[2].z = (float)r2[2];
    q[3].x = (float)(n[0] * invlen); q[3].y = (float)(n[1] * invlen); q[3].z = (float)(n[2] * invlen);
  }

  // A ray that leaves a surface cannot hit a triangle lying in that surface's plane; traversal drops
  // such candidates up front.  Planes are compared through the unit normal and offset of each triangle.
  void mark_coplanar(uint32_t node_index, uint32_t first, uint32_t count) {
    double pl[24][4];
    uint32_t prim[24];
    bool ok[24];
    for (uint32_t i = 0; i < count; ++i) {
      prim[i] = f2u(out.tris[4 * (size_t)(first + i) + 1].w);
      const uint32_t* t = b.idx + 3 * (size_t)prim[i];
      double v[3][3];
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) v[k][c] = (double)b.verts[3 * (size_t)t[k] + c];
      const double e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
      const double e2[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
      double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      ok[i] = len > 0.0 && std::isfinite(len);
      if (!ok[i]) continue;
      for (int c = 0; c < 3; ++c) pl[i][c] = n[c] / len;
      pl[i][3] = pl[i][0] * v[0][0] + pl[i][1] * v[0][1] + pl[i][2] * v[0][2];
    }
    const double tol = 1e-9 * std::max(1.0, (double)out.vmag);
    for (uint32_t i = 0; i < count; ++i) {
      uint32_t mask = 0;
      if (ok[i]) {
        for (uint32_t j = 0; j < count; ++j) {
          if (!ok[j]) continue;
          const double dp = pl[i][0] * pl[j][0] + pl[i][1] * pl[j][1] + pl[i][2] * pl[j][2];
          const bool same = (dp > 1.0 - 1e-12 && std::fabs(pl[i][3] - pl[j][3]) <= tol) ||
                            (dp < -1.0 + 1e-12 && std::fabs(pl[i][3] + pl[j][3]) <= tol);
          if (same) mask |= 1u << j;
        }
      }
      out.home_node[prim[i]] = node_index;
      out.home_mask[prim[i]] = mask;
    }
  }

  // returns the number of internal-node levels below and including this node
  int emit(uint32_t out_index, int n2) {
    // children of this node: the cost-optimal forest of up to 8 entries below n2
    int child[8];
    int nchild = 0;
    const Node2& root = b.nodes2[n2];
    if (root.count > 0) {
      child[nchild++] = n2;
    } else {
      int k = 1;
      {  // the split that compute_collapse() chose for "n2 as an internal node"
        const auto& l = b.dp[(size_t)root.left];
        const auto& r = b.dp[(size_t)root.right];
        float best = FLT_MAX;
        for (int kk = 1; kk <= 7; ++kk) {
          const float v = l.cost[kk - 1] + r.cost[7 - kk];
          if (v < best) { best = v; k = kk; }
        }
      }
      b.forest(root.left, k - 1, child, nchild);
      b.forest(root.right, 7 - k, child, nchild);
    }
    // node box and grid
    Box nb;
    nb.reset();
    for (int i = 0; i < nchild; ++i) nb.grow(b.nodes2[child[i]].box);
    const float cx = 0.5f * (nb.lo[0] + nb.hi[0]), cy = 0.5f * (nb.lo[1] + nb.hi[1]), cz = 0.5f * (nb.lo[2] + nb.hi[2]);
    // slot assignment: a child placed in slot s should lie towards the corner
    // ((s&4)?+x:-x, (s&2)?+y:-y, (s&1)?+z:-z) of the node, so that visiting slots in the
    // order (slot ^ octinv) descending walks the children front to back for any ray octant.
    int slot_of[8];
    bool slot_used[8] = {false, false, false, false, false, false, false, false};
    bool child_done[8] = {false, false, false, false, false, false, false, false};
    float cost[8][8];
    for (int i = 0; i < nchild; ++i) {
      const Box& cb = b.nodes2[child[i]].box;
      const float ox = 0.5f * (cb.lo[0] + cb.hi[0]) - cx, oy = 0.5f * (cb.lo[1] + cb.hi[1]) - cy, oz = 0.5f * (cb.lo[2] + cb.hi[2]) - cz;
      for (int s = 0; s < 8; ++s)
        cost[i][s] = ((s & 4) ? ox : -ox) + ((s & 2) ? oy : -oy) + ((s & 1) ? oz : -oz);
    }
    for (int round = 0; round < nchild; ++round) {
      int bi = -1, bs = -1;
      float bc = -FLT_MAX;
      for (int i = 0; i < nchild; ++i) {
        if (child_done[i]) continue;
        for (int s = 0; s < 8; ++s) {
          if (slot_used[s]) continue;
          if (cost[i][s] > bc) { bc = cost[i][s]; bi = i; bs = s; }
        }
      }
      slot_of[bi] = bs;
      child_done[bi] = true;
      slot_used[bs] = true;
    }
    int child_in_slot[8];
    for (int s = 0; s < 8; ++s) child_in_slot[s] = -1;
    for (int i = 0; i < nchild; ++i) child_in_slot[slot_of[i]] = child[i];

    BvhNode node;
    memset(&node, 0, sizeof(node));
    node.px = nb.lo[0]; node.py = nb.lo[1]; node.pz = nb.lo[2];
    int e[3];
    for (int k = 0; k < 3; ++k) {
      const double ext = (double)nb.hi[k] - (double)nb.lo[k];
      int ek = -126;
      if (ext > 0.0) {
        ek = (int)std::ceil(std::log2(ext / 255.0));
        while (std::ldexp(255.0, ek) < ext) ++ek;
        ek = std::max(ek, -126);
      }
      e[k] = ek;
    }
    node.ex = (uint8_t)(e[0] + 127); node.ey = (uint8_t)(e[1] + 127); node.ez = (uint8_t)(e[2] + 127);

    int n_internal = 0;
    for (int s = 0; s < 8; ++s)
      if (child_in_slot[s] >= 0 && b.nodes2[child_in_slot[s]].count == 0) n_internal++;
    const uint32_t child_base = (uint32_t)out.nodes.size();
    out.nodes.resize(out.nodes.size() + (size_t)n_internal);
    node.child_base = child_base;
    node.tri_base = tri_cursor;
    uint32_t tri_off = 0;
    uint8_t* qlo[3] = {node.qlox, node.qloy, node.qloz};
    uint8_t* qhi[3] = {node.qhix, node.qhiy, node.qhiz};
    for (int s = 0; s < 8; ++s) {
      const int c = child_in_slot[s];
      if (c < 0) {
        node.meta[s] = 0;
        for (int k = 0; k < 3; ++k) { qlo[k][s] = 255; qhi[k][s] = 0; }
        continue;
      }
      const Node2& cn = b.nodes2[c];
      for (int k = 0; k < 3; ++k) {
        const double sc = std::ldexp(1.0, e[k]);
        double lo = std::floor(((double)cn.box.lo[k] - (double)nb.lo[k]) / sc);
        double hi = std::ceil(((double)cn.box.hi[k] - (double)nb.lo[k]) / sc);
        qlo[k][s] = (uint8_t)std::min(std::max(lo, 0.0), 255.0);
        qhi[k][s] = (uint8_t)std::min(std::max(hi, 0.0), 255.0);
      }
      if (cn.count == 0) {
        node.imask |= (uint8_t)(1u << s);
        node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
      } else {
        const uint32_t unary = cn.count == 1 ? 1u : (cn.count == 2 ? 3u : 7u);
        node.meta[s] = (uint8_t)((unary << 5) | tri_off);
        for (int i = 0; i < cn.count; ++i) write_tri(tri_cursor + tri_off + (uint32_t)i, b.order[cn.first + i]);
        tri_off += (uint32_t)cn.count;
      }
    }
    mark_coplanar(out_index, tri_cursor, tri_off);
    tri_cursor += tri_off;
    out.nodes[out_index] = node;

    int depth_below = 0;
    uint32_t rank = 0;
    for (int s = 0; s < 8; ++s) {
      const int c = child_in_slot[s];
      if (c < 0 || b.nodes2[c].count > 0) continue;
      depth_below = std::max(depth_below, emit(child_base + rank, c));
      rank++;
    }
    return 1 + depth_below;
  }
};

}  // namespace

BvhBuildResult build_bvh8(const float* vertices, const uint32_t* indices, int num_tris) {
  BvhBuildResult out;
  out.num_tris = num_tris;
  if (num_tris <= 0) {
    // a single empty node: every ray misses
    BvhNode n;
    memset(&n, 0, sizeof(n));
    n.ex = n.ey = n.ez = 1;
    for (int s = 0; s < 8; ++s) { n.qlox[s] = n.qloy[s] = n.qloz[s] = 255; }
    out.nodes.push_back(n);
    out.depth = 1;
    return out;
  }
  Builder b;
  b.verts = vertices;
  b.idx = indices;
  b.tri_box.resize((size_t)num_tris);
  b.centroid.resize(3 * (size_t)num_tris);
  b.order.resize((size_t)num_tris);
  std::iota(b.order.begin(), b.order.end(), 0u);
  for (int t = 0; t < num_tris; ++t) {
    Box bx;
    bx.reset();
    for (int k = 0; k < 3; ++k) bx.grow(vertices + 3 * (size_t)indices[3 * (size_t)t + k]);
    for (int k = 0; k < 3; ++k) out.vmag = std::max(out.vmag, std::max(std::fabs(bx.lo[k]), std::fabs(bx.hi[k])));
    for (int k = 0; k < 3; ++k) {
      b.centroid[3 * (size_t)t + k] = 0.5f * (bx.lo[k] + bx.hi[k]);
      // pad so that rays grazing a flat or grid-aligned box are not lost to rounding
      const float pad = 1e-6f * std::max(std::max(std::fabs(bx.lo[k]), std::fabs(bx.hi[k])), 1e-3f);
      bx.lo[k] -= pad;
      bx.hi[k] += pad;
    }
    b.tri_box[(size_t)t] = bx;
  }
  b.nodes2.reserve(2 * (size_t)num_tris);
  const int root2 = b.build2(0, num_tris);
  b.dp.resize(b.nodes2.size());
  b.compute_collapse(root2);
  out.tris.resize(4 * (size_t)num_tris);
  out.home_node.assign((size_t)num_tris, 0u);
  out.home_mask.assign((size_t)num_tris, 0u);
  out.nodes.reserve((size_t)num_tris / 2 + 16);
  out.nodes.resize(1);
  Emitter em{b, out};
  out.depth = em.emit(0, root2);
  return out;
}

}  // namespace pt
