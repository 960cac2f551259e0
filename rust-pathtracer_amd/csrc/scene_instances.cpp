// scene_instances.cpp -- the two-level tree of a scene made of instances (host side).
// No counterpart in the reference: its pipeline allows single-level instancing (optix_wrapper.cpp:565-566) but its
// one acceleration structure is built over flat triangles (optix_wrapper.cpp:492-544).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bvh_build.h"
#include "scene.h"

namespace pt {

// world -> object rows of an affine object -> world matrix (three rows of x y z | translation); false if singular
bool invert_affine(const float m[12], double inv[12]) {
  const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (!(std::fabs(det) > 1e-30) || !std::isfinite(det)) return false;
  const double r = 1.0 / det;
  const double w[9] = {(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r,
                       (f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r,
                       (d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r};
  for (int row = 0; row < 3; ++row) {
    for (int k = 0; k < 3; ++k) inv[4 * row + k] = w[3 * row + k];
    inv[4 * row + 3] = -(w[3 * row] * m[3] + w[3 * row + 1] * m[7] + w[3 * row + 2] * m[11]);
  }
  return true;
}

// Two-level tree of an instanced scene in the arrays of a flat one: the top-level tree over the instances'
// world boxes comes first (node 0 is its root, its leaf items are instance records), each mesh's own tree follows
// with its node, item and primitive indices moved to where it landed.
BvhBuildResult build_instanced(const HostScene& hs) {
  BvhBuildResult out;
  const size_t nt = (size_t)hs.num_tris();
  out.num_tris = (int)nt;
  out.home_node.assign(nt, 0u);
  out.home_mask.assign(nt, 0u);
  out.corner_order.assign(nt, 0x24);
  struct Blas { BvhBuildResult r; uint32_t node_base = 0, item_base = 0; float lo[3], hi[3]; };
  std::vector<Blas> blas(hs.meshes.size());
  for (size_t m = 0; m < hs.meshes.size(); ++m) {
    const HostScene::MeshRange& mr = hs.meshes[m];
    blas[m].r = build_bvh8(hs.vertices.data(), hs.indices.data() + 3 * (size_t)mr.first_tri, (int)mr.num_tris);
    for (int k = 0; k < 3; ++k) { blas[m].lo[k] = 3e38f; blas[m].hi[k] = -3e38f; }
    for (size_t j = 0; j < 3 * (size_t)mr.num_tris; ++j) {
      const float* v = &hs.vertices[3 * (size_t)hs.indices[3 * (size_t)mr.first_tri + j]];
      for (int k = 0; k < 3; ++k) { blas[m].lo[k] = std::min(blas[m].lo[k], v[k]); blas[m].hi[k] = std::max(blas[m].hi[k], v[k]); }
    }
  }
  // world boxes of the instances: the eight corners of the mesh's box, padded against rounding
  const size_t ni = hs.instances.size();
  std::vector<float> boxes(6 * ni);
  for (size_t n = 0; n < ni; ++n) {
    const HostScene::Instance& in = hs.instances[n];
    const Blas& bl = blas[in.mesh];
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int corner = 0; corner < 8; ++corner) {
      const float p[3] = {(corner & 1) ? bl.hi[0] : bl.lo[0], (corner & 2) ? bl.hi[1] : bl.lo[1], (corner & 4) ? bl.hi[2] : bl.lo[2]};
      for (int r = 0; r < 3; ++r) {
        const float w = in.xform[4 * r] * p[0] + in.xform[4 * r + 1] * p[1] + in.xform[4 * r + 2] * p[2] + in.xform[4 * r + 3];
        lo[r] = std::min(lo[r], w);
        hi[r] = std::max(hi[r], w);
      }
    }
    for (int k = 0; k < 3; ++k) {
      const float pad = 1e-5f * std::max(std::max(std::fabs(lo[k]), std::fabs(hi[k])), hi[k] - lo[k]) + 1e-7f;
      boxes[6 * n + k] = lo[k] - pad;
      boxes[6 * n + 3 + k] = hi[k] + pad;
    }
  }
  BvhBuildResult top = build_bvh8_boxes(boxes.data(), (int)ni);
  // placement
  uint32_t node_cursor = (uint32_t)top.nodes.size(), item_cursor = (uint32_t)top.num_items;
  int mesh_depth = 0;
  for (Blas& bl : blas) {
    bl.node_base = node_cursor;
    bl.item_base = item_cursor;
    node_cursor += (uint32_t)bl.r.nodes.size();
    item_cursor += (uint32_t)bl.r.num_items;
    mesh_depth = std::max(mesh_depth, bl.r.depth);
  }
  out.nodes = top.nodes;
  out.nodes.reserve(node_cursor);
  out.tris = top.tris;
  out.tris.reserve(4 * (size_t)item_cursor);
  out.inst_records.assign(6 * ni, q4{0.f, 0.f, 0.f, 0.f});
  for (size_t n = 0; n < ni; ++n) {
    const HostScene::Instance& in = hs.instances[n];
    double inv[12];
    if (!invert_affine(in.xform, inv)) memset(inv, 0, sizeof(inv));  // checked when the instance was added
    q4* x = &out.inst_records[6 * n];
    for (int r = 0; r < 3; ++r) x[r] = q4{(float)inv[4 * r], (float)inv[4 * r + 1], (float)inv[4 * r + 2], (float)inv[4 * r + 3]};
    if (in.has_material) {
      const Material& mt = in.material;
      x[3] = q4{mt.base[0], mt.base[1], mt.base[2], mt.roughness};
      x[4] = q4{mt.emission[0], mt.emission[1], mt.emission[2], mt.ior};
      x[5] = q4{mt.transmission, 1.0f, 0.f, 0.f};
    }
  }
  for (size_t it = 0; it < (size_t)top.num_items; ++it) {  // the top level's leaf items become instance records
    q4* q = &out.tris[4 * it];
    const uint32_t n = f2u(q[1].w);
    const Blas& bl = blas[hs.instances[n].mesh];
    for (int r = 0; r < 3; ++r) q[r] = out.inst_records[6 * (size_t)n + r];
    q[3] = q4{u2f(bl.node_base), u2f(n), bl.r.vmag, 0.f};
  }
  for (size_t m = 0; m < blas.size(); ++m) {
    const Blas& bl = blas[m];
    const uint32_t first = hs.meshes[m].first_tri;
    for (BvhNode nd : bl.r.nodes) {
      nd.child_base += bl.node_base;
      nd.tri_base += bl.item_base;
      out.nodes.push_back(nd);
    }
    for (size_t it = 0; it < (size_t)bl.r.num_items; ++it) {
      q4 q[4] = {bl.r.tris[4 * it], bl.r.tris[4 * it + 1], bl.r.tris[4 * it + 2], bl.r.tris[4 * it + 3]};
      q[1].w = u2f(f2u(q[1].w) + first);
      if (f2u(q[2].w)) q[3].w = u2f(f2u(q[3].w) + first);
      for (int k = 0; k < 4; ++k) out.tris.push_back(q[k]);
    }
    for (uint32_t t = 0; t < hs.meshes[m].num_tris; ++t) {
      out.home_node[first + t] = bl.r.home_node[t] + bl.node_base;
      out.home_mask[first + t] = bl.r.home_mask[t];
      out.corner_order[first + t] = bl.r.corner_order[t];
    }
  }
  out.vmag = top.vmag;
  out.num_items = (int)item_cursor;
  out.depth = top.depth + mesh_depth + 2;  // entering an instance sets two entries of the top level aside
  return out;
}

}  // namespace pt
