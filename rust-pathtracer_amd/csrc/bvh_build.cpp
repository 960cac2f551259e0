// bvh_build.cpp -- binned-SAH binary build, greedy collapse to 8-wide nodes, octant-ordered
// slot assignment and 8-bit box quantisation.
#include "bvh_build.h"
#include "pt_bvh.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

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
  int prims = 0;              // triangles in the subtree
};

constexpr int kBins = 16;
// Items per leaf of the binary tree.  One: every item gets its own box in an 8-wide node, which culls better
// than it costs (measured against 2 and 3 on the built-in scene, +2 %, and on 12.8 M triangles, +5 %).
constexpr int kMaxLeaf = 1;
constexpr uint32_t kNoPartner = 0xffffffffu;

// What a leaf slot holds: one triangle, or two triangles that together form a parallelogram (a quad split
// along a diagonal), which traversal tests with one plane intersection.
struct Item {
  uint32_t a, b;   // input triangles; b == kNoPartner: single
  // corner orders (2 bits per corner: which input corner comes first, second, third) in which the two halves
  // of a parallelogram are recorded: a starts at its corner off the shared diagonal, b at its own such corner
  // followed by the images of a's second and third corner under the half-turn that maps a onto b
  uint8_t order_a, order_b;
  float delta;     // how far b's own corner is from the exact parallelogram corner (max norm)
};
constexpr uint8_t kInputOrder = 0 | (1 << 2) | (2 << 4);

// Triangles A and B share an edge and B's third corner is A's third corner mirrored through the middle of
// that edge (within rounding of the coordinates).
bool try_pair(const float* verts, const uint32_t* idx, uint32_t A, uint32_t B, float vmag, Item& item) {
  const uint32_t *ia = idx + 3 * (size_t)A, *ib = idx + 3 * (size_t)B;
  const float* va[3];
  const float* vb[3];
  for (int m = 0; m < 3; ++m) { va[m] = verts + 3 * (size_t)ia[m]; vb[m] = verts + 3 * (size_t)ib[m]; }
  auto same = [](const float* p, const float* q) { return p[0] == q[0] && p[1] == q[1] && p[2] == q[2]; };
  int in_b[3] = {-1, -1, -1};  // for A's corner m: the equal corner of B
  bool b_used[3] = {false, false, false};
  int shared = 0;
  for (int m = 0; m < 3; ++m)
    for (int w = 0; w < 3; ++w)
      if (!b_used[w] && in_b[m] < 0 && same(va[m], vb[w])) { in_b[m] = w; b_used[w] = true; shared++; }
  if (shared != 2) return false;
  int k = 0, kb = 0;
  for (int m = 0; m < 3; ++m) if (in_b[m] < 0) k = m;
  for (int w = 0; w < 3; ++w) if (!b_used[w]) kb = w;
  const int i = (k + 1) % 3, j = (k + 2) % 3;
  if (same(va[i], va[j])) return false;
  double delta = 0.0, n[3], e1[3], e2[3];
  for (int c = 0; c < 3; ++c) {
    const double ideal = (double)va[i][c] + (double)va[j][c] - (double)va[k][c];
    delta = std::max(delta, std::fabs((double)vb[kb][c] - ideal));
    e1[c] = (double)va[1][c] - (double)va[0][c];
    e2[c] = (double)va[2][c] - (double)va[0][c];
  }
  n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
  const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  if (!(nn > 0.0) || !std::isfinite(nn)) return false;
  if (!(delta <= std::ldexp((double)vmag, -21))) return false;
  // the half-turn about the middle of the shared edge sends A's corner i to j, j to i and k to B's own corner
  item.a = A;
  item.b = B;
  item.order_a = (uint8_t)(k | (i << 2) | (j << 4));
  item.order_b = (uint8_t)(kb | (in_b[j] << 2) | (in_b[i] << 4));
  item.delta = (float)delta;
  return true;
}

struct Builder {
  const float* verts;
  const uint32_t* idx;
  // Build-time copy of a triangle: ranges are partitioned by moving these records, so every pass
  // over a range streams through memory instead of chasing an index list.
  struct Prim {
    Box box;
    float c[3];
    uint32_t id;
  };
  std::vector<Prim> prims;
  std::vector<uint32_t> order;  // triangle permutation (prims[i].id once the binary tree is built)
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
    if (std::min(nodes2[nd.left].prims, nodes2[nd.right].prims) >= (1 << 15) && live_threads.fetch_add(1) < max_threads) {
      std::thread worker([&] { compute_collapse(nd.left); live_threads.fetch_sub(1); });
      compute_collapse(nd.right);
      worker.join();
    } else {
      if (std::min(nodes2[nd.left].prims, nodes2[nd.right].prims) >= (1 << 15)) live_threads.fetch_sub(1);
      compute_collapse(nd.left);
      compute_collapse(nd.right);
    }
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

  // nodes2 is sized up front and handed out through a counter, so subtrees can be built by
  // different threads; the tree's shape does not depend on which thread built what.
  std::atomic<int> next_node{0};
  int max_threads = 1;
  std::atomic<int> live_threads{1};

  int build_root(int count) {
    Box box, cb;
    box.reset();
    cb.reset();
    for (int i = 0; i < count; ++i) {
      box.grow(prims[(size_t)i].box);
      cb.grow(prims[(size_t)i].c);
    }
    return build2(0, count, box, cb);
  }

  // Ranges of a few triangles (most nodes of the tree): exact sweep SAH over the centroid order on each
  // axis instead of bins -- cheaper than clearing and sweeping 48 bins, and never worse.
  static constexpr int kSweepMax = 8;
  int build_small(int id, int first, int count, const Box& box) {
    Prim* pr = prims.data() + first;
    float best_cost = FLT_MAX;
    int best_axis = -1, best_split = -1;
    uint8_t sorted[3][kSweepMax];  // positions in the range
    Box right[kSweepMax + 1];
    for (int axis = 0; axis < 3; ++axis) {
      uint8_t* sa = sorted[axis];
      for (int i = 0; i < count; ++i) sa[i] = (uint8_t)i;
      for (int i = 1; i < count; ++i) {  // insertion sort by centroid, ties by primitive index: deterministic
        const uint8_t t = sa[i];
        const float c = pr[t].c[axis];
        int j = i - 1;
        while (j >= 0) {
          const float cj = pr[sa[j]].c[axis];
          if (cj < c || (cj == c && pr[sa[j]].id < pr[t].id)) break;
          sa[j + 1] = sa[j];
          --j;
        }
        sa[j + 1] = t;
      }
      right[count].reset();
      for (int i = count - 1; i > 0; --i) { right[i] = right[i + 1]; right[i].grow(pr[sa[i]].box); }
      Box left;
      left.reset();
      for (int i = 1; i < count; ++i) {  // split: [0,i) | [i,count)
        left.grow(pr[sa[i - 1]].box);
        const float cost = left.half_area() * (float)i + right[i].half_area() * (float)(count - i);
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = i; }
      }
    }
    const float parent_area = box.half_area();
    if (count <= kMaxLeaf) {
      const float leaf_cost = (float)count * parent_area;
      const float split_cost = 1.0f * parent_area + best_cost;
      if (leaf_cost <= split_cost) {
        nodes2[(size_t)id].first = first;
        nodes2[(size_t)id].count = count;
        return id;
      }
    }
    Prim tmp[kSweepMax];
    for (int i = 0; i < count; ++i) tmp[i] = pr[sorted[best_axis][i]];
    for (int i = 0; i < count; ++i) pr[i] = tmp[i];
    Box lbox, rbox, unused;
    lbox.reset(); rbox.reset(); unused.reset();
    for (int i = 0; i < count; ++i) (i < best_split ? lbox : rbox).grow(pr[i].box);
    const int l = build2(first, best_split, lbox, unused);
    const int r = build2(first + best_split, count - best_split, rbox, unused);
    nodes2[(size_t)id].left = l;
    nodes2[(size_t)id].right = r;
    return id;
  }

  // box: bounds of the range's triangles; cb: bounds of their centroids (both known from the parent's bins)
  int build2(int first, int count, const Box& box, const Box& cb) {
    const int id = next_node.fetch_add(1);
    Node2& self = nodes2[(size_t)id];
    self.box = box;
    self.prims = count;
    if (count == 1) {
      self.first = first;
      self.count = count;
      return id;
    }
    if (count <= kSweepMax) return build_small(id, first, count, box);
    // one pass bins the range on all three axes; each bin keeps its triangle and centroid bounds
    Box bb[3][kBins], cbb[3][kBins];
    int bc[3][kBins];
    float lo[3], scale[3];
    bool usable[3];
    for (int axis = 0; axis < 3; ++axis) {
      const float ext = cb.hi[axis] - cb.lo[axis];
      usable[axis] = ext > 0.f;
      lo[axis] = cb.lo[axis];
      scale[axis] = usable[axis] ? (float)kBins / ext : 0.f;
      for (int b = 0; b < kBins; ++b) { bb[axis][b].reset(); cbb[axis][b].reset(); bc[axis][b] = 0; }
    }
    for (int i = first; i < first + count; ++i) {
      const float* c = prims[(size_t)i].c;
      const Box& tb = prims[(size_t)i].box;
      for (int axis = 0; axis < 3; ++axis) {
        if (!usable[axis]) continue;
        int b = (int)((c[axis] - lo[axis]) * scale[axis]);
        b = std::min(std::max(b, 0), kBins - 1);
        bb[axis][b].grow(tb);
        cbb[axis][b].grow(c);
        bc[axis][b]++;
      }
    }
    float best_cost = FLT_MAX;
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
      if (!usable[axis]) continue;
      float right_area[kBins];
      int right_cnt[kBins];
      Box acc;
      acc.reset();
      int c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        acc.grow(bb[axis][b]);
        c += bc[axis][b];
        right_area[b] = acc.half_area();
        right_cnt[b] = c;
      }
      acc.reset();
      c = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        acc.grow(bb[axis][b]);
        c += bc[axis][b];
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
      }
    }
    const float parent_area = box.half_area();
    if (count <= kMaxLeaf) {
      // leaf cost: count triangle tests; split cost: one more node level plus the children
      const float leaf_cost = (float)count * parent_area;
      const float split_cost = best_axis < 0 ? FLT_MAX : 1.0f * parent_area + best_cost;
      if (leaf_cost <= split_cost) {
        self.first = first;
        self.count = count;
        return id;
      }
    }
    int mid;
    Box lbox, lcb, rbox, rcb;
    lbox.reset(); lcb.reset(); rbox.reset(); rcb.reset();
    bool have_child_bounds = false;
    if (best_axis >= 0) {
      const float plo = lo[best_axis], pscale = scale[best_axis];
      auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& q) {
        int b = (int)((q.c[best_axis] - plo) * pscale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= best_bin;
      });
      mid = (int)(it - prims.begin());
      if (mid != first && mid != first + count) {
        for (int b = 0; b < kBins; ++b) {
          if (bc[best_axis][b] == 0) continue;
          if (b <= best_bin) { lbox.grow(bb[best_axis][b]); lcb.grow(cbb[best_axis][b]); }
          else { rbox.grow(bb[best_axis][b]); rcb.grow(cbb[best_axis][b]); }
        }
        have_child_bounds = true;
      }
    }
    if (!have_child_bounds) {
      mid = first + count / 2;  // all centroids coincide: split by position in the list
      for (int i = first; i < first + count; ++i) {
        Box& tbx = i < mid ? lbox : rbox;
        Box& tcb = i < mid ? lcb : rcb;
        tbx.grow(prims[(size_t)i].box);
        tcb.grow(prims[(size_t)i].c);
      }
    }
    const int lcount = mid - first, rcount = first + count - mid;
    int l, r;
    // big subtrees go to their own thread while threads are left
    if (std::min(lcount, rcount) >= (1 << 15) && live_threads.fetch_add(1) < max_threads) {
      // the worker gives its slot back the moment it is done, not when it is joined
      std::thread worker([&] { l = build2(first, lcount, lbox, lcb); live_threads.fetch_sub(1); });
      r = build2(mid, rcount, rbox, rcb);
      worker.join();
    } else {
      if (std::min(lcount, rcount) >= (1 << 15)) live_threads.fetch_sub(1);
      l = build2(first, lcount, lbox, lcb);
      r = build2(mid, rcount, rbox, rcb);
    }
    nodes2[(size_t)id].left = l;
    nodes2[(size_t)id].right = r;
    return id;
  }
};

struct Emitter {
  Builder& b;
  BvhBuildResult& out;
  const std::vector<Item>& items;
  // for the binary node that becomes an 8-wide node: 8-wide nodes strictly below it, triangles in it
  // and below.  Known before emission, so every node's place in the arrays is fixed up front and
  // subtrees can be emitted by different threads into the same layout a single thread would produce.
  std::vector<uint32_t> below;

  // children of the 8-wide node rooted at binary node n2: the cost-optimal forest of up to 8 entries
  int children_of(int n2, int* child) const {
    int nchild = 0;
    const Node2& root = b.nodes2[n2];
    if (root.count > 0) {
      child[nchild++] = n2;
      return nchild;
    }
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
    return nchild;
  }

  uint32_t count_below(int n2) {
    int child[8];
    const int nchild = children_of(n2, child);
    uint32_t total = 0;
    int big[8], nbig = 0;
    for (int i = 0; i < nchild; ++i) {
      if (b.nodes2[child[i]].count > 0) continue;
      if (b.nodes2[child[i]].prims >= (1 << 16)) big[nbig++] = child[i];
      else total += 1u + count_below(child[i]);
    }
    if (nbig > 0) {  // large subtrees side by side
      uint32_t res[8];
      std::vector<std::thread> workers;
      for (int i = 1; i < nbig; ++i) workers.emplace_back([&, i] { res[i] = count_below(big[i]); });
      res[0] = count_below(big[0]);
      for (auto& w : workers) w.join();
      for (int i = 0; i < nbig; ++i) total += 1u + res[i];
    }
    below[(size_t)n2] = total;
    return total;
  }

  // traversal record of a leaf item (layout: the record comment in pt_bvh.h), computed in double
  void write_tri(uint32_t slot_index, uint32_t item_index) {
    const Item& item = items[(size_t)item_index];
    const uint32_t prim = item.a;
    if (box_mode) {  // the caller writes the record of box `prim` here once the order is known
      q4* q = &out.tris[4 * (size_t)slot_index];
      q[0] = q[2] = q[3] = q4{0.f, 0.f, 0.f, 0.f};
      q[1] = q4{0.f, 0.f, 0.f, u2f(prim)};
      return;
    }
    const uint32_t* t = b.idx + 3 * (size_t)prim;
    double v[3][3];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) v[k][c] = (double)b.verts[3 * (size_t)t[(item.order_a >> (2 * k)) & 3] + c];
    out.corner_order[prim] = item.order_a;
    if (item.b != kNoPartner) out.corner_order[item.b] = item.order_b;
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
    double eps = std::ldexp(norm1 + norm2, -20);
    if (item.b != kNoPartner) {
      // the partner is tested as this triangle's image: widen the band by what its real corner is off by
      eps += (norm1 + norm2) * (double)item.delta / std::max((double)out.vmag, 1e-30);
      q[2].w = u2f(1u);
      q[3].w = u2f(item.b);
    }
    q[0].w = (float)eps;
    q[1].x = (float)r1[0]; q[1].y = (float)r1[1]; q[1].z = (float)r1[2];
    q[2].x = (float)r2[0]; q[2].y = (float)r2[1]; q[2].z = (float)r2[2];
    q[3].x = (float)(n[0] * invlen); q[3].y = (float)(n[1] * invlen); q[3].z = (float)(n[2] * invlen);
  }

  // A ray that leaves a surface cannot hit a triangle lying in that surface's plane; traversal drops
  // such candidates up front.  Planes are compared through the unit normal and offset of each triangle.
  void mark_coplanar(uint32_t node_index, uint32_t first, uint32_t count) {
    mark_coplanar(node_index, first, count, out.home_node.data(), 1, out.home_mask.data(), 1);
  }
  // writes node_index to hn[prim * hn_stride] and the mask to hm[prim * hm_stride]
  void mark_coplanar(uint32_t node_index, uint32_t first, uint32_t count, uint32_t* hn, int hn_stride, uint32_t* hm, int hm_stride) {
    if (box_mode) return;
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
      hn[(size_t)prim[i] * hn_stride] = node_index;
      hm[(size_t)prim[i] * hm_stride] = mask;
      if (f2u(out.tris[4 * (size_t)(first + i) + 2].w)) {  // the partner lies in the same plane
        const uint32_t partner = f2u(out.tris[4 * (size_t)(first + i) + 3].w);
        hn[(size_t)partner * hn_stride] = node_index;
        hm[(size_t)partner * hm_stride] = mask;
      }
    }
  }

  bool box_mode = false;  // the items are caller-defined boxes, not triangles

  // ---- wide nodes for small trees (layout: pt_bvh.h)
  std::vector<float> slot_boxes;  // per node and slot: the child's box, lo xyz hi xyz (filled by emit)
  struct WideLeaf {
    const float* box;
    uint32_t first, count;  // items, relative to the packed tree's item order
  };
  // leaf slots of the subtree below packed node i, in item order
  void collect_leaves(uint32_t i, std::vector<WideLeaf>& leaves) const {
    const BvhNode& nd = out.nodes[i];
    uint32_t rank = 0;
    std::vector<uint32_t> kids;
    for (int s = 0; s < 8; ++s) {
      const uint32_t meta = nd.meta[s];
      if (!meta) continue;
      if (((meta & (meta << 1)) & 0x10u) != 0u) {
        kids.push_back(nd.child_base + rank++);
      } else {
        const uint32_t unary = (meta >> 5) & 7u;
        leaves.push_back(WideLeaf{&slot_boxes[48 * (size_t)i + 6 * s], nd.tri_base + (meta & 0x1fu),
                                  (uint32_t)__builtin_popcount(unary)});
      }
    }
    for (uint32_t k : kids) collect_leaves(k, leaves);
  }
  static float round_down(double v) {
    float f = (float)v;
    return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
  }
  static float round_up(double v) {
    float f = (float)v;
    return (double)f < v ? std::nextafterf(f, INFINITY) : f;
  }
  void put_slot(u4* node, int slot, const float* box, uint32_t bits) const {
    uint32_t* w = reinterpret_cast<uint32_t*>(node);
    w[4 * kWideBits + slot] = bits;
    // padded by a few 1e-6 of the scene's size: covers the rounding of t = plane * (1/d) - o * (1/d)
    const double pad = 4e-6 * std::max((double)out.vmag, 1e-3);
    for (int k = 0; k < 3; ++k) {
      w[4 * (kWideBounds + kWidePlane * k) + slot] = f2u(box ? round_down((double)box[k] - pad) : 3e38f);
      w[4 * (kWideBounds + kWidePlane * (3 + k)) + slot] = f2u(box ? round_up((double)box[3 + k] + pad) : -3e38f);
    }
  }
  int wide_node(uint32_t i) {
    u4* node = out.wide_nodes.data() + (size_t)kWideNodeWords * i;
    for (int slot = 0; slot < 16; ++slot) put_slot(node, slot, nullptr, 0u);
    const BvhNode& nd = out.nodes[i];
    std::vector<WideLeaf> leaves;
    collect_leaves(i, leaves);
    uint32_t items = 0;
    for (const WideLeaf& l : leaves) items = std::max(items, l.first + l.count - nd.tri_base);
    if (leaves.size() <= 16 && items <= 24) {  // the whole subtree in one node
      for (size_t k = 0; k < leaves.size(); ++k)
        put_slot(node, (int)k, leaves[k].box, ((1u << leaves[k].count) - 1u) << (leaves[k].first - nd.tri_base));
      node[0] = u4{nd.tri_base << 8, 0u, (uint32_t)std::max<size_t>((leaves.size() + 3) / 4, 1), 0u};
      out.wide_groups = std::max(out.wide_groups, node[0].z);
      mark_coplanar(i, nd.tri_base, items, out.home_wide.data(), 2, out.home_wide.data() + 1, 2);
      return 1;
    }
    int below = 0;
    uint32_t rank = 0, own_items = 0;
    for (int s = 0; s < 8; ++s) {
      const uint32_t meta = nd.meta[s];
      if (!meta) continue;
      const float* box = &slot_boxes[48 * (size_t)i + 6 * s];
      if (((meta & (meta << 1)) & 0x10u) != 0u) {
        put_slot(node, s, box, 1u << (24 + s));
        below = std::max(below, wide_node(nd.child_base + rank++));
      } else {
        const uint32_t unary = (meta >> 5) & 7u;
        put_slot(node, s, box, unary << (meta & 0x1fu));
        own_items = std::max(own_items, (meta & 0x1fu) + (uint32_t)__builtin_popcount(unary));
      }
    }
    node[0] = u4{(uint32_t)nd.imask | (nd.tri_base << 8), nd.child_base, 2u, 0u};
    out.wide_groups = std::max(out.wide_groups, 2u);
    mark_coplanar(i, nd.tri_base, own_items, out.home_wide.data(), 2, out.home_wide.data() + 1, 2);
    return 1 + below;
  }
  void build_wide() {
    out.wide_nodes.assign((size_t)kWideNodeWords * out.nodes.size(), u4{0u, 0u, 0u, 0u});
    out.home_wide.assign(2 * (size_t)out.num_tris, 0u);
    out.wide_groups = 1;
    out.wide_depth = wide_node(0);
  }

  // returns the number of internal-node levels below and including this node
  // block_start: where this node's block of internal children goes; tri_cursor: first triangle slot of
  // this node (its leaf slots first, then its subtrees in slot order) -- depth-first order throughout
  int emit(uint32_t out_index, int n2, uint32_t block_start, uint32_t tri_cursor) {
    int child[8];
    const int nchild = children_of(n2, child);
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
    const uint32_t child_base = block_start;
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
        const double glo = ((double)cn.box.lo[k] - (double)nb.lo[k]) / sc, ghi = ((double)cn.box.hi[k] - (double)nb.lo[k]) / sc;
        double lo = std::floor(glo);
        double hi = std::ceil(ghi);
        qlo[k][s] = (uint8_t)std::min(std::max(lo, 0.0), 255.0);
        qhi[k][s] = (uint8_t)std::min(std::max(hi, 0.0), 255.0);
        if (!slot_boxes.empty()) {  // small tree: remember the child's box itself for the wide nodes
          slot_boxes[48 * (size_t)out_index + 6 * s + k] = cn.box.lo[k];
          slot_boxes[48 * (size_t)out_index + 6 * s + 3 + k] = cn.box.hi[k];
        }
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
    out.nodes[out_index] = node;

    // subtrees in slot order; each one's node blocks and triangles follow those of the ones before it
    int depth_below = 0;
    uint32_t rank = 0;
    uint32_t next_block = block_start + (uint32_t)n_internal, next_tri = tri_cursor + tri_off;
    std::vector<std::thread> workers;
    int worker_depth[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < 8; ++s) {
      const int c = child_in_slot[s];
      if (c < 0 || b.nodes2[c].count > 0) continue;
      const uint32_t my_index = child_base + rank, my_block = next_block, my_tri = next_tri;
      next_block += below[(size_t)c];
      next_tri += (uint32_t)b.nodes2[c].prims;
      if (b.nodes2[c].prims >= (1 << 16) && b.live_threads.fetch_add(1) < b.max_threads) {
        workers.emplace_back([this, my_index, c, my_block, my_tri, s, &worker_depth] {
          worker_depth[s] = emit(my_index, c, my_block, my_tri);
          b.live_threads.fetch_sub(1);
        });
      } else {
        if (b.nodes2[c].prims >= (1 << 16)) b.live_threads.fetch_sub(1);
        depth_below = std::max(depth_below, emit(my_index, c, my_block, my_tri));
      }
      rank++;
    }
    for (auto& w : workers) w.join();
    for (int s = 0; s < 8; ++s) depth_below = std::max(depth_below, worker_depth[s]);
    return 1 + depth_below;
  }
};

int g_pairing = -1;  // -1: environment (PT_BVH_PAIRS, default on)

}  // namespace

void bvh_set_pairing(int on) { g_pairing = on < 0 ? -1 : (on ? 1 : 0); }

static BvhBuildResult build_common(const float* vertices, const uint32_t* indices, int num_tris, const float* boxes);

BvhBuildResult build_bvh8(const float* vertices, const uint32_t* indices, int num_tris) {
  return build_common(vertices, indices, num_tris, nullptr);
}
BvhBuildResult build_bvh8_boxes(const float* boxes, int num_boxes) {
  return build_common(nullptr, nullptr, num_boxes, boxes);
}

// boxes != null: the primitives are boxes (lo xyz, hi xyz each) and the leaf records are left to the caller
static BvhBuildResult build_common(const float* vertices, const uint32_t* indices, int num_tris, const float* boxes) {
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
  if (boxes) {
    for (size_t k = 0; k < 6 * (size_t)num_tris; ++k) out.vmag = std::max(out.vmag, std::fabs(boxes[k]));
  } else {
    for (size_t k = 0; k < 3 * (size_t)num_tris; ++k) {
      const float* v = vertices + 3 * (size_t)indices[k];
      out.vmag = std::max(out.vmag, std::max(std::fabs(v[0]), std::max(std::fabs(v[1]), std::fabs(v[2]))));
    }
  }
  // Leaf items: a triangle and its successor in the input become one item when they are the two halves of
  // a parallelogram (how quads arrive in a triangle list); everything else stays a single triangle.
  const bool pairing = !boxes && (g_pairing >= 0 ? g_pairing != 0 : !(getenv("PT_BVH_PAIRS") && atoi(getenv("PT_BVH_PAIRS")) == 0));
  std::vector<Item> items;
  items.reserve((size_t)num_tris);
  for (int t = 0; t < num_tris; ++t) {
    Item item{(uint32_t)t, kNoPartner, kInputOrder, kInputOrder, 0.f};
    if (pairing && t + 1 < num_tris && try_pair(vertices, indices, (uint32_t)t, (uint32_t)t + 1u, out.vmag, item)) ++t;
    items.push_back(item);
  }
  const int num_items = (int)items.size();
  out.num_items = num_items;
  b.prims.resize((size_t)num_items);
  for (int it = 0; it < num_items; ++it) {
    Box bx;
    bx.reset();
    if (boxes) {
      bx.grow(boxes + 6 * (size_t)it);
      bx.grow(boxes + 6 * (size_t)it + 3);
    } else {
      for (int k = 0; k < 3; ++k) bx.grow(vertices + 3 * (size_t)indices[3 * (size_t)items[(size_t)it].a + k]);
      if (items[(size_t)it].b != kNoPartner)
        for (int k = 0; k < 3; ++k) bx.grow(vertices + 3 * (size_t)indices[3 * (size_t)items[(size_t)it].b + k]);
    }
    Builder::Prim& pr = b.prims[(size_t)it];
    for (int k = 0; k < 3; ++k) {
      pr.c[k] = 0.5f * (bx.lo[k] + bx.hi[k]);
      // pad so that rays grazing a flat or grid-aligned box are not lost to rounding
      const float pad = 1e-6f * std::max(std::max(std::fabs(bx.lo[k]), std::fabs(bx.hi[k])), 1e-3f);
      bx.lo[k] -= pad;
      bx.hi[k] += pad;
    }
    pr.box = bx;
    pr.id = (uint32_t)it;
  }
  b.nodes2.resize(2 * (size_t)num_items);
  {
    const unsigned hw = std::thread::hardware_concurrency();
    const char* env = getenv("PT_BVH_THREADS");
    b.max_threads = env ? std::max(atoi(env), 1) : (int)std::min<unsigned>(std::max(hw, 1u), 16u);
  }
  const bool timing = getenv("PT_BVH_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double>(b2 - a).count();
  };
  const auto t0 = now();
  const int root2 = b.build_root(num_items);
  const auto t1 = now();
  b.nodes2.resize((size_t)b.next_node.load());
  b.order.resize((size_t)num_items);
  for (int t = 0; t < num_items; ++t) b.order[(size_t)t] = b.prims[(size_t)t].id;
  std::vector<Builder::Prim>().swap(b.prims);
  b.dp.resize(b.nodes2.size());
  b.compute_collapse(root2);
  const auto t2 = now();
  out.tris.resize(4 * (size_t)num_items);
  out.home_node.assign((size_t)num_tris, 0u);
  out.home_mask.assign((size_t)num_tris, 0u);
  out.corner_order.assign((size_t)num_tris, kInputOrder);
  Emitter em{b, out, items};
  em.box_mode = boxes != nullptr;
  em.below.assign(b.nodes2.size(), 0u);
  const uint32_t nodes_below_root = em.count_below(root2);
  out.nodes.resize(1 + (size_t)nodes_below_root);
  if (!boxes && out.nodes.size() <= kWideMaxNodes) em.slot_boxes.assign(48 * out.nodes.size(), 0.0f);
  out.depth = em.emit(0, root2, 1u, 0u);
  if (!em.slot_boxes.empty()) em.build_wide();
  if (timing)
    fprintf(stderr, "bvh build: %d triangles in %d leaf items, binary tree %.2f s (%d threads), collapse %.2f s, emit %.2f s\n", num_tris, num_items,
            secs(t0, t1), b.max_threads, secs(t1, t2), secs(t2, now()));
  return out;
}

}  // namespace pt
