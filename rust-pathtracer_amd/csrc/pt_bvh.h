// pt_bvh.h -- software ray casting for gfx950 (a7): traversal of the compressed 8-wide BVH
// with a short per-lane stack, and a watertight two-sided ray/triangle test.
//
// Replaces what the reference gets from optixTrace + the driver-built GAS
// (optix_device.cu:508-520, 976-986; optix_wrapper.cpp:492-544): closest hit returning the
// primitive index and barycentrics (b1,b2) of v1,v2, or first-hit termination for shadow
// rays.  The ray interval is open: (tmin, tmax).
#pragma once
#include "pt_types.h"

namespace pt {

PT_HD f3 ldq(const q4& q) { return mk3(q.x, q.y, q.z); }
PT_HD f3 ldv(const q4& q) { return mk3(q.x, q.y, q.z); }

struct Hit {
  int prim;  // -1: miss
  float t, b1, b2;
};

// Traversal stack entry: {base index, hit bits}.  Node group: bits 24..31 are the still
// unvisited child slots in front-to-back priority order, bits 0..7 the node's imask.
// One entry per BVH level at most.
struct StackEntry {
  uint32_t base, bits;
};

// Per-ray constants of the watertight test (Woop, Benthin, Wald, JCGT 2013).  The shear
// "translate, permute so the dominant axis is z, shear x/y" is folded into three row
// vectors so no per-vertex component selection is needed:
//   Ax = dot(A, rx), Ay = dot(A, ry), Az = dot(A, rz)
// Every vertex goes through the same instruction sequence for a given ray, so vertices
// shared by two triangles get identical (Ax, Ay) and the edge functions below agree on
// shared edges.
struct RayShear {
  f3 rx, ry, rz;
};

PT_HD RayShear make_shear(f3 d) {
  float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  int kx = kz == 2 ? 0 : kz + 1;
  int ky = kx == 2 ? 0 : kx + 1;
  float dz = comp(d, kz);
  if (dz < 0.0f) { int t = kx; kx = ky; ky = t; }
  float inv = rcp(dz);
  float sx = comp(d, kx) * inv, sy = comp(d, ky) * inv;
  RayShear r;
  r.rx = mk3(kx == 0 ? 1.0f : (kz == 0 ? -sx : 0.0f), kx == 1 ? 1.0f : (kz == 1 ? -sx : 0.0f),
             kx == 2 ? 1.0f : (kz == 2 ? -sx : 0.0f));
  r.ry = mk3(ky == 0 ? 1.0f : (kz == 0 ? -sy : 0.0f), ky == 1 ? 1.0f : (kz == 1 ? -sy : 0.0f),
             ky == 2 ? 1.0f : (kz == 2 ? -sy : 0.0f));
  r.rz = mk3(kz == 0 ? inv : 0.0f, kz == 1 ? inv : 0.0f, kz == 2 ? inv : 0.0f);
  return r;
}

// Edge functions must not be contracted into FMAs: U for edge (B,C) in one triangle has
// to be the exact negation of the same edge seen from the neighbouring triangle.
PT_HD void edge_functions(float Ax, float Ay, float Bx, float By, float Cx, float Cy, float& U, float& V, float& W) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  float cxby = Cx * By, cybx = Cy * Bx;
  float axcy = Ax * Cy, aycx = Ay * Cx;
  float bxay = Bx * Ay, byax = By * Ax;
  U = cxby - cybx;
  V = axcy - aycx;
  W = bxay - byax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
}

// Sheared coordinate of a vertex: one multiply and two explicit FMAs in a fixed order.  Left to
// the compiler, a*b + c*d + e*f may be fused differently at different call sites (either
// product of the first sum can be the rounded one), and a vertex shared by two triangles would
// then get two slightly different images -- which is all it takes to open a crack along the
// shared edge.
PT_HD float shear_dot(f3 a, f3 r) { return __builtin_fmaf(a.z, r.z, __builtin_fmaf(a.y, r.y, a.x * r.x)); }

// Returns true and updates hit when the triangle is hit inside (tmin, hit.t).
PT_HD bool intersect_triangle(const RayShear& rs, f3 o, f3 v0, f3 v1, f3 v2, float tmin, int prim, Hit& hit) {
  const f3 A = v0 - o, B = v1 - o, C = v2 - o;
  const float Ax = shear_dot(A, rs.rx), Ay = shear_dot(A, rs.ry);
  const float Bx = shear_dot(B, rs.rx), By = shear_dot(B, rs.ry);
  const float Cx = shear_dot(C, rs.rx), Cy = shear_dot(C, rs.ry);
  float U, V, W;
  edge_functions(Ax, Ay, Bx, By, Cx, Cy, U, V, W);
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  float det = U + V + W;
  if (det == 0.0f) return false;
  const float Az = dot(A, rs.rz), Bz = dot(B, rs.rz), Cz = dot(C, rs.rz);
  float T = U * Az + V * Bz + W * Cz;
  if (det < 0.0f) { T = -T; det = -det; }
  if (!(T > tmin * det) || !(T < hit.t * det)) return false;
  const float r = rcp(det);
  hit.t = T * r;
  hit.b1 = fabsf(V) * r;
  hit.b2 = fabsf(W) * r;
  hit.prim = prim;
  return true;
}

// Traversal record of a leaf item, 64 bytes: a precomputed affine map to "unit triangle" space.
//   q0 = v0.xyz, eps     q1 = r1.xyz, primitive index bits     q2 = r2.xyz, pair code     q3 = n.xyz, partner
// r1, r2 are the dual basis of the edges (r1.(v1-v0) = 1, r1.(v2-v0) = 0, ...), n the unit normal,
// so for a point p: b1 = r1.(p-v0), b2 = r2.(p-v0), distance from the plane = n.(p-v0).
// eps bounds the rounding error of b1, b2 per unit of coordinate magnitude.
// Pair code != 0: the item is a parallelogram, this triangle plus the partner primitive, which is this
// triangle turned by 180 degrees about the middle of the edge v1-v2 (the shared diagonal; v0 is the corner
// off it).  The partner's shading record lists its corners as (image of v0, image of v1, image of v2), so
// a point with weights (b0, b1, b2) here and b0 < 0 has weights (-b0, 1 - b1, 1 - b2) there.
constexpr int kTriWords = 4;

// Fast ray/triangle test: solve for t on the plane, evaluate the barycentrics directly.  It decides
// alone when the hit point is clearly inside or clearly outside; a point within the error band
// of an edge goes to the watertight test on the original vertices (intersect_triangle), so shared
// edges and vertices keep the watertight test's answer and no ray slips between two triangles.
// Returns 0 miss, 1 hit (hit updated), 2 undecided.
// dlen: an upper bound of |d| (1 for the unit directions of world-space rays; a ray taken into an instance's object
// space keeps its parameter and so has a direction of another length).
PT_HD int fast_triangle(const q4& q0, const q4& q1, const q4& q2, const q4& q3, f3 o, f3 d, float rmag, float tmin,
                        Hit& hit, float dlen = 1.0f) {
  // every kernel must get the same bits out of this test: products and sums are fused only where written
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const f3 A = o - ldq(q0);
  const f3 n = ldq(q3);
  const float t = -shear_dot(n, A) * rcp(shear_dot(n, d));  // parallel ray: t is inf or nan and the next test fails
  // The interval test is exact on this t.  A ray that starts on the plane itself (the reference's 1e-3 offset puts
  // a path that leaves the glass box through its bottom face exactly on the floor) has t = +-1e-9: which side it
  // falls on is rounding noise in any formulation, so this one's sign decides and no second opinion is asked for.
  if (!(t > tmin) || !(t < hit.t)) return 0;
  const f3 r1 = ldq(q1), r2 = ldq(q2);
  float b1 = __builtin_fmaf(t, shear_dot(r1, d), shear_dot(r1, A));
  float b2 = __builtin_fmaf(t, shear_dot(r2, d), shear_dot(r2, A));
  const float b0 = 1.0f - b1 - b2;
  const float eps = q0.w * (rmag + fabsf(t) * dlen);
  float lo = fminf(fminf(b0, b1), b2);
  uint32_t prim = f2u(q1.w);
  const uint32_t code = f2u(q2.w);
  if (code != 0u && b0 < -eps) {  // clearly beyond the diagonal of a parallelogram: the point's weights in the other half
    b1 = 1.0f - b1;
    b2 = 1.0f - b2;
    lo = fminf(fminf(-b0, b1), b2);
    prim = f2u(q3.w);
  }
  if (!(lo >= -eps)) return 0;
  if (lo <= eps) return 2;
  hit.t = t;
  hit.b1 = b1;
  hit.b2 = b2;
  hit.prim = (int)prim;
  return 1;
}

// The undecided case of fast_triangle: the watertight test on the original vertices of the item's
// triangle(s) decides.  Testing a triangle the point is not near costs time only.
PT_HD bool watertight_item(const SceneView& sc, const q4& q1, const q4& q2, const q4& q3, f3 o, f3 d, float tmin, Hit& hit) {
#if PT_ON_DEVICE
  // This runs for one ray in thousands.  Its per-ray constants depend on the direction alone, and the compiler, left
  // to itself, computes them ahead of the traversal loop for every ray (some 60 selects, a dozen registers held
  // through the loop); the empty statement ties the direction to this place.
  asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
#endif
  const RayShear rs = make_shear(d);
  const int prim = (int)f2u(q1.w);
  const q4* rec = reinterpret_cast<const q4*>(sc.shade + prim);
  bool found = intersect_triangle(rs, o, ldv(rec[0]), ldv(rec[1]), ldv(rec[2]), tmin, prim, hit);
  if (f2u(q2.w) != 0u) {
    const int partner = (int)f2u(q3.w);
    const q4* rec2 = reinterpret_cast<const q4*>(sc.shade + partner);
    found |= intersect_triangle(rs, o, ldv(rec2[0]), ldv(rec2[1]), ldv(rec2[2]), tmin, partner, hit);
  }
  return found;
}

// Hits carry barycentrics in the corner order of the shading record (ShadeRec::corner_order); these two
// convert to and from the order in which the scene's triangles were given.
PT_HD void to_input_order(uint32_t corner_order, float& b1, float& b2) {
  float w[3];
  const float b[3] = {1.0f - b1 - b2, b1, b2};
  for (int m = 0; m < 3; ++m) w[(corner_order >> (2 * m)) & 3u] = b[m];
  b1 = w[1];
  b2 = w[2];
}
PT_HD void to_record_order(uint32_t corner_order, float& b1, float& b2) {
  const float w[3] = {1.0f - b1 - b2, b1, b2};
  const float r1 = w[(corner_order >> 2) & 3u], r2 = w[(corner_order >> 4) & 3u];
  b1 = r1;
  b2 = r2;
}

PT_HD float safe_inv(float d) {
  float a = fabsf(d) < 1e-20f ? (d < 0.0f ? -1e-20f : 1e-20f) : d;
  return rcp(a);
}

// Ray constants for the child-box tests of one node.
struct RayBox {
  f3 o, idir;
  uint32_t octinv;  // bit2: d.x >= 0, bit1: d.y >= 0, bit0: d.z >= 0
};

// A node as the five 16-byte words it is loaded with.
struct NodeWords {
  u4 q[5];
};
PT_HD NodeWords load_node(const BvhNode* nodes, uint32_t idx) {
  const u4* p = reinterpret_cast<const u4*>(nodes) + 8u * idx;
  NodeWords n;
  n.q[0] = p[0]; n.q[1] = p[1]; n.q[2] = p[2]; n.q[3] = p[3]; n.q[4] = p[4];
  return n;
}
PT_HD float byte_f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xffu); }

// Slab test of the node's 8 quantised child boxes; returns the hit bits
// (children: 24 + (slot ^ octinv); triangles: their offsets in the node's triangle range).
PT_HD uint32_t intersect_children(const NodeWords& n, const RayBox& rb, float tmin, float tmax) {
  const uint32_t ew = n.q[0].w;  // ex | ey<<8 | ez<<16 | imask<<24
  const float sx = u2f((ew & 0xffu) << 23) * rb.idir.x;
  const float sy = u2f(((ew >> 8) & 0xffu) << 23) * rb.idir.y;
  const float sz = u2f(((ew >> 16) & 0xffu) << 23) * rb.idir.z;
  const float bx = (u2f(n.q[0].x) - rb.o.x) * rb.idir.x;
  const float by = (u2f(n.q[0].y) - rb.o.y) * rb.idir.y;
  const float bz = (u2f(n.q[0].z) - rb.o.z) * rb.idir.z;
  const bool px = (rb.octinv & 4u) != 0u, py = (rb.octinv & 2u) != 0u, pz = (rb.octinv & 1u) != 0u;
  // words: q[2] = qlox[0..7] qloy[0..7]; q[3] = qloz[0..7] qhix[0..7]; q[4] = qhiy[0..7] qhiz[0..7]
  const uint32_t lox[2] = {n.q[2].x, n.q[2].y}, loy[2] = {n.q[2].z, n.q[2].w}, loz[2] = {n.q[3].x, n.q[3].y};
  const uint32_t hix[2] = {n.q[3].z, n.q[3].w}, hiy[2] = {n.q[4].x, n.q[4].y}, hiz[2] = {n.q[4].z, n.q[4].w};
  const uint32_t meta[2] = {n.q[1].z, n.q[1].w};
  const uint32_t octinv4 = rb.octinv * 0x01010101u;
  uint32_t mask = 0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const uint32_t m4 = meta[half];
    const uint32_t is_inner4 = (m4 & (m4 << 1)) & 0x10101010u;
    const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
    const uint32_t bit_index4 = (m4 ^ (octinv4 & inner_mask4)) & 0x1f1f1f1fu;
    const uint32_t child_bits4 = (m4 >> 5) & 0x07070707u;
    const uint32_t nxw = px ? lox[half] : hix[half], fxw = px ? hix[half] : lox[half];
    const uint32_t nyw = py ? loy[half] : hiy[half], fyw = py ? hiy[half] : loy[half];
    const uint32_t nzw = pz ? loz[half] : hiz[half], fzw = pz ? hiz[half] : loz[half];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tnx = byte_f(nxw, j) * sx + bx, tfx = byte_f(fxw, j) * sx + bx;
      const float tny = byte_f(nyw, j) * sy + by, tfy = byte_f(fyw, j) * sy + by;
      const float tnz = byte_f(nzw, j) * sz + bz, tfz = byte_f(fzw, j) * sz + bz;
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
      const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax));
      if (tn <= tf) mask |= ((child_bits4 >> (8 * j)) & 0xffu) << ((bit_index4 >> (8 * j)) & 0xffu);
    }
  }
  return mask;
}

// Wide nodes: the form in which a small tree is kept in LDS by the fused kernels (built on the host,
// bvh_build.cpp: build_wide).  A node is kWideNodeWords u4 and holds up to 16 slots in groups of four:
//   u4[0]      = imask | first leaf item << 8, first internal child, number of slot groups in use (1..4), 0
//                (kernels run SceneView::wide_groups groups on every node: unused slots are never hit)
//   u4[1..4]   = per slot, the traversal bits the child contributes when its box is hit (leaf: its item bits;
//                internal: bit 24 + slot, moved to 24 + (slot ^ octinv) after the test)
//   u4[5..28]  = the child boxes in world coordinates as floats, 16 slots each: lo x, lo y, lo z, hi x, hi y, hi z
//                (unquantised, padded outwards; an empty slot is (+big, -big))
// Node indices are those of the packed tree.  A subtree with at most 16 leaf slots and 24 items is flattened
// into its root (no internal children; its descendants' entries are unused), otherwise a node keeps its 8
// slots.  Boxes are never tighter than the children, so hits are those of the packed tree; only the
// candidates tested for nothing differ.
constexpr uint32_t kWideNodeWords = 32;
constexpr uint32_t kWideBits = 1, kWideBounds = 5, kWidePlane = 4;  // u4 offsets / u4 per plane

// Ray constants of the wide test: t = plane * idir + oi with oi = -o * idir.
struct RayWide {
  f3 idir, oi;
  uint32_t octinv;
};

// far_bound = false: the ray has no far end worth testing against (tmax beyond any box: a closest-hit ray before its
// first hit), one instruction less per slot.
PT_HD uint32_t intersect_children_wide(const u4* n, const RayWide& rw, float tmin, float tmax, uint32_t groups, bool far_bound = true) {
  const bool px = (rw.octinv & 4u) != 0u, py = (rw.octinv & 2u) != 0u, pz = (rw.octinv & 1u) != 0u;
  const u4* b = n + kWideBounds;
  const u4* nx = b + (px ? 0 : 3) * kWidePlane, *fx = b + (px ? 3 : 0) * kWidePlane;
  const u4* ny = b + (py ? 1 : 4) * kWidePlane, *fy = b + (py ? 4 : 1) * kWidePlane;
  const u4* nz = b + (pz ? 2 : 5) * kWidePlane, *fz = b + (pz ? 5 : 2) * kWidePlane;
  uint32_t mask = 0;
  // one group of four slots per trip; the counter is hidden from the optimiser (an empty asm), which otherwise
  // unrolls or software-pipelines the loop and spills dozens of registers doing so
  uint32_t g = 0;
#pragma unroll 1
  do {
    const u4 w = n[kWideBits + g];
    const u4 ax = nx[g], bxw = fx[g], ay = ny[g], byw = fy[g], az = nz[g], bzw = fz[g];
    const uint32_t wj[4] = {w.x, w.y, w.z, w.w};
    const uint32_t nxj[4] = {ax.x, ax.y, ax.z, ax.w}, fxj[4] = {bxw.x, bxw.y, bxw.z, bxw.w};
    const uint32_t nyj[4] = {ay.x, ay.y, ay.z, ay.w}, fyj[4] = {byw.x, byw.y, byw.z, byw.w};
    const uint32_t nzj[4] = {az.x, az.y, az.z, az.w}, fzj[4] = {bzw.x, bzw.y, bzw.z, bzw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tnx = u2f(nxj[j]) * rw.idir.x + rw.oi.x, tfx = u2f(fxj[j]) * rw.idir.x + rw.oi.x;
      const float tny = u2f(nyj[j]) * rw.idir.y + rw.oi.y, tfy = u2f(fyj[j]) * rw.idir.y + rw.oi.y;
      const float tnz = u2f(nzj[j]) * rw.idir.z + rw.oi.z, tfz = u2f(fzj[j]) * rw.idir.z + rw.oi.z;
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
      const float tf = far_bound ? fminf(fminf(tfx, tfy), fminf(tfz, tmax)) : fminf(fminf(tfx, tfy), tfz);
      if (tn <= tf) mask |= wj[j];
    }
    ++g;
#if PT_ON_DEVICE
    asm volatile("" : "+s"(g));
#endif
  } while (g < groups);
  if (mask & 0xff000000u) {  // internal children: slot s is visited in the order of s ^ octinv
    uint32_t top = mask >> 24;
    if (px) top = ((top & 0xf0u) >> 4) | ((top & 0x0fu) << 4);
    if (py) top = ((top & 0xccu) >> 2) | ((top & 0x33u) << 2);
    if (pz) top = ((top & 0xaau) >> 1) | ((top & 0x55u) << 1);
    mask = (mask & 0x00ffffffu) | (top << 24);
  }
  return mask;
}

// Stack policies: push(level, entry) / pop(level).
struct LocalStack {
  StackEntry e[64];
  PT_HD void push(int sp, StackEntry v) { e[sp] = v; }
  PT_HD StackEntry pop(int sp) const { return e[sp]; }
};

// Closest hit (any_hit = false) or first hit (any_hit = true; hit.prim >= 0 means occluded).
// origin: 1 + the primitive the ray leaves (its origin lies on that primitive's plane, pushed off towards
// the side it travels to), or 0.  Triangles in that plane cannot be hit and are not tested; passing 0 for
// such a ray gives the same result, only slower.
template <class Stack>
PT_HD Hit trace_ray(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, bool any_hit, Stack& stack, int stack_limit,
                    uint32_t origin = 0u) {
  Hit hit{-1, tmax, 0.f, 0.f};
  uint32_t home_node = 0xffffffffu, home_mask = 0u;
  if (origin) {
    if (sc.wide) {  // the wide tree has its own node numbering of where a primitive lives
      home_node = sc.home_wide[2u * (origin - 1u)];
      home_mask = sc.home_wide[2u * (origin - 1u) + 1u];
    } else {
      const ShadeRec* rec = sc.shade + (origin - 1u);
      home_node = rec->home_node;
      home_mask = rec->home_mask;
    }
  }
  RayBox rb;
  rb.o = o;
  rb.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  rb.octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);

  const float rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.vmag;

  StackEntry cur{0u, 0x80000000u};
  int sp = 0;
  for (;;) {
    StackEntry tg;
    PT_TIC(t_node);
    if (cur.bits & 0xff000000u) {
      PT_PROF(any_hit ? PF_SH_NODE : PF_NODE);
      const uint32_t hits_imask = cur.bits;
      const int bit = 31 - __builtin_clz(hits_imask);
      cur.bits &= ~(1u << bit);
      if ((cur.bits & 0xff000000u) && sp < stack_limit) stack.push(sp++, cur);
      const uint32_t slot = (uint32_t)(bit - 24) ^ rb.octinv;
      const uint32_t rel = (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
      const uint32_t node_index = cur.base + rel;
      uint32_t hm;
      if (sc.wide) {
        const u4* n = sc.wide + kWideNodeWords * node_index;
        const u4 head = n[0];
        // -o / d is recomputed at every visit (three multiplies) rather than kept in registers across the
        // traversal; the empty asm keeps the compiler from hoisting it out of the loop again
        f3 oo = o;
#if PT_ON_DEVICE
        asm volatile("" : "+v"(oo.x), "+v"(oo.y), "+v"(oo.z));
#endif
        RayWide rw;
        rw.idir = rb.idir;
        rw.oi = mk3(-oo.x * rb.idir.x, -oo.y * rb.idir.y, -oo.z * rb.idir.z);
        rw.octinv = rb.octinv;
        hm = intersect_children_wide(n, rw, tmin, hit.t, sc.wide_groups);  // the same trip count on every lane
        cur.base = head.y;
        cur.bits = (hm & 0xff000000u) | (head.x & 0xffu);
        tg.base = head.x >> 8;
      } else {
        const NodeWords n = load_node(sc.nodes, node_index);
        hm = intersect_children(n, rb, tmin, hit.t);
        cur.base = n.q[1].x;
        cur.bits = (hm & 0xff000000u) | (n.q[0].w >> 24);
        tg.base = n.q[1].y;
      }
      tg.bits = hm & 0x00ffffffu;
      if (node_index == home_node) tg.bits &= ~home_mask;
    } else {
      tg = cur;
      cur.base = 0u;
      cur.bits = 0u;
    }
    PT_TOC(t_node, PT_T_NODE);
    PT_TIC(t_tri);
    while (tg.bits) {
      PT_PROF(any_hit ? PF_SH_TRI : PF_TRI);
      const int ti = __builtin_ctz(tg.bits);
      tg.bits &= tg.bits - 1u;
      const q4* tp = sc.tris + (uint32_t)kTriWords * (tg.base + (uint32_t)ti);
      const q4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
      int res = fast_triangle(q0, q1, q2, q3, o, d, rmag, tmin, hit);
      if (res == 2) {  // rare: within rounding of an edge -- the watertight test on the original vertices decides
        PT_PROF(PF_FALLBACK);
        res = watertight_item(sc, q1, q2, q3, o, d, tmin, hit) ? 1 : 0;
      }
      if (res == 1 && any_hit) return hit;
    }
    PT_TOC(t_tri, PT_T_TRI);
    if ((cur.bits & 0xff000000u) == 0u) {
      if (sp == 0) break;
      PT_PROF(any_hit ? PF_SH_POP : PF_POP);
      cur = stack.pop(--sp);
    }
  }
  return hit;
}

// trace_ray for a wide tree that is one node (up to 16 leaf slots and 24 items: the reference's scene is one): what
// trace_ray does for such a tree -- the same slab test, the same item tests in the same order, the same hit -- without
// what it carries for trees in general (the loop over nodes with its stack, the child ranks, the order of internal
// children): about 35 vector instructions and half a dozen registers less per query.
// unbounded: the caller's tmax is beyond every box of the scene (the slab test then skips its far clamp).
PT_HD Hit trace_ray_one_node(const SceneView& sc, f3 o, f3 d, float tmax, bool any_hit, uint32_t origin, bool unbounded = false) {
  Hit hit{-1, tmax, 0.f, 0.f};
  RayWide rw;
  rw.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  rw.oi = mk3(-o.x * rw.idir.x, -o.y * rw.idir.y, -o.z * rw.idir.z);
  rw.octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
  const float rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.vmag;
  const u4* n = sc.wide;
  uint32_t bits = intersect_children_wide(n, rw, 0.0f, tmax, sc.wide_groups, !unbounded) & 0x00ffffffu;
  if (origin) bits &= ~sc.home_wide[2u * (origin - 1u) + 1u];  // every primitive's home is this node
  const uint32_t base = n[0].x >> 8;
  while (bits) {
    PT_PROF(any_hit ? PF_SH_TRI : PF_TRI);
    const int ti = __builtin_ctz(bits);
    bits &= bits - 1u;
    const q4* tp = sc.tris + (uint32_t)kTriWords * (base + (uint32_t)ti);
    const q4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
    int res = fast_triangle(q0, q1, q2, q3, o, d, rmag, 0.0f, hit);
    if (res == 2) {
      PT_PROF(PF_FALLBACK);
      res = watertight_item(sc, q1, q2, q3, o, d, 0.0f, hit) ? 1 : 0;
    }
    if (res == 1 && any_hit) return hit;
  }
  return hit;
}

// The same traversal as trace_ray, cut into single steps so that a kernel can interleave rays: a lane
// whose ray has finished takes the next ray while its neighbours are still traversing theirs.
// A ray is in one of three states: triangles of a visited node wait to be tested (wants_tri), child
// nodes wait to be visited (wants_node), or nothing is left (done).
//
// INST: the scene is a top-level tree over instances of meshes (SceneView::inst_records).  A leaf item of the
// top level is an instance record,
//   q0..q2 = rows of the world -> object matrix (xyz | translation)
//   q3     = root node of the mesh's tree, instance index, largest |coordinate| of the mesh, 0
// and testing it means entering the instance: what is left of the top-level node goes on the stack, the ray is
// taken to object space (direction not renormalised, so distances along the ray stay the same numbers) and the
// mesh's tree is walked with it; when the stack is back to where it was, the world ray resumes.  Rays get no
// origin-plane hint in such scenes.
constexpr uint32_t kNoInstance = 0xffffffffu;
template <bool INST>
struct InstancePart {};
template <>
struct InstancePart<true> {
  f3 oo, od;          // the ray in the space of the instance it is in
  float dlen;         // |od.x| + |od.y| + |od.z|: how far a unit of the ray's parameter goes there, at most
  uint32_t inst;      // that instance, kNoInstance at the top level
  uint32_t hit_inst;  // the instance of the closest hit so far
  int ret_sp;         // stack level at which the instance was entered
};

template <bool INST>
struct TraceStateT : InstancePart<INST> {
  // the ray itself (origin, direction, tmin) stays with the caller and is passed to every step
  f3 idir;
  uint32_t octinv;
  float rmag;
  Hit hit;
  StackEntry cur, tg;
  int sp;
  uint32_t home_node, home_mask;

  PT_HD void set_ray(f3 o, f3 d, float vmag) {
    idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    octinv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
    rmag = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + vmag;
  }
  PT_HD void begin(const SceneView& sc, f3 o, f3 d, float tmax, uint32_t origin) {
    hit = Hit{-1, tmax, 0.f, 0.f};
    home_node = 0xffffffffu;
    home_mask = 0u;
    if (!INST && origin) {
      const ShadeRec* rec = sc.shade + (origin - 1u);
      home_node = rec->home_node;
      home_mask = rec->home_mask;
    }
    set_ray(o, d, sc.vmag);
    cur = StackEntry{0u, 0x80000000u};
    tg = StackEntry{0u, 0u};
    sp = 0;
    if constexpr (INST) {
      this->inst = kNoInstance;
      this->hit_inst = kNoInstance;
      this->ret_sp = 0;
      this->oo = o;
      this->od = d;
      this->dlen = 1.0f;
    }
  }
  PT_HD bool wants_tri() const { return tg.bits != 0u; }
  PT_HD bool wants_node() const { return tg.bits == 0u && (cur.bits != 0u || sp > 0); }  // cur.bits != 0: children left
  PT_HD bool done() const { return tg.bits == 0u && cur.bits == 0u && sp == 0; }
  PT_HD void finish() { tg.bits = 0u; cur.bits = 0u; sp = 0; }

  // visit one child node (o, d: the world ray)
  template <class Stack>
  PT_HD void node_step(const SceneView& sc, f3 o, f3 d, float tmin, Stack& stack, int stack_limit) {
    if (cur.bits == 0u) {
      if constexpr (INST) {
        if (this->inst != kNoInstance && sp == this->ret_sp) {  // the mesh's tree is done: back to the world ray
          this->inst = kNoInstance;
          set_ray(o, d, sc.vmag);
          if (sp == 0) return;
        }
      }
      cur = stack.pop(--sp);  // only entries with something left to visit are ever pushed
      if constexpr (INST) {
        if ((cur.bits & 0xff000000u) == 0u) {  // instance records of a top-level node that were set aside
          tg = cur;
          cur.bits = 0u;
          return;
        }
      }
    }
    const uint32_t hits_imask = cur.bits;
    const int bit = 31 - __builtin_clz(hits_imask);
    cur.bits &= ~(1u << bit);
    if ((cur.bits & 0xff000000u) && sp < stack_limit) stack.push(sp++, cur);
    const uint32_t slot = (uint32_t)(bit - 24) ^ octinv;
    const uint32_t rel = (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
    const uint32_t node_index = cur.base + rel;
    const NodeWords n = load_node(sc.nodes, node_index);
    RayBox rb;
    rb.o = o;
    if constexpr (INST) {
      if (this->inst != kNoInstance) rb.o = this->oo;
    }
    rb.idir = idir;
    rb.octinv = octinv;
    const uint32_t hm = intersect_children(n, rb, tmin, hit.t);
    cur.base = n.q[1].x;
    cur.bits = hm & 0xff000000u;
    if (cur.bits) cur.bits |= n.q[0].w >> 24;
    tg.base = n.q[1].y;
    tg.bits = hm & 0x00ffffffu;
    if (node_index == home_node) tg.bits &= ~home_mask;
  }

  // test one waiting leaf item; returns true when an any-hit ray has found its occluder
  template <class Stack>
  PT_HD bool tri_step(const SceneView& sc, f3 o, f3 d, float tmin, bool any_hit, Stack& stack, int stack_limit) {
    const int ti = __builtin_ctz(tg.bits);
    tg.bits &= tg.bits - 1u;
    const q4* tp = sc.tris + (uint32_t)kTriWords * (tg.base + (uint32_t)ti);
    const q4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
#if PT_ON_DEVICE
    // The record comes from device memory here: all four words in one round trip.  Left alone, the compiler
    // sinks the loads of the two barycentric rows behind the plane-distance test, a second dependent miss per item.
    asm volatile("" ::"v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
#endif
    if constexpr (INST) {
      if (this->inst == kNoInstance) {  // an instance record: enter it
        if (tg.bits && sp < stack_limit) stack.push(sp++, StackEntry{tg.base, tg.bits});
        if ((cur.bits & 0xff000000u) && sp < stack_limit) stack.push(sp++, cur);
        this->ret_sp = sp;
        this->inst = f2u(q3.y);
        this->oo = mk3(q0.x * o.x + q0.y * o.y + q0.z * o.z + q0.w, q1.x * o.x + q1.y * o.y + q1.z * o.z + q1.w,
                       q2.x * o.x + q2.y * o.y + q2.z * o.z + q2.w);
        this->od = mk3(q0.x * d.x + q0.y * d.y + q0.z * d.z, q1.x * d.x + q1.y * d.y + q1.z * d.z,
                       q2.x * d.x + q2.y * d.y + q2.z * d.z);
        set_ray(this->oo, this->od, q3.z);
        this->dlen = fabsf(this->od.x) + fabsf(this->od.y) + fabsf(this->od.z);
        cur = StackEntry{f2u(q3.x), 0x80000000u};
        tg.bits = 0u;
        return false;
      }
      int res = fast_triangle(q0, q1, q2, q3, this->oo, this->od, rmag, tmin, hit, this->dlen);
      if (res == 2) res = watertight_item(sc, q1, q2, q3, this->oo, this->od, tmin, hit) ? 1 : 0;
      if (res == 1) this->hit_inst = this->inst;
      return res == 1 && any_hit;
    } else {
      int res = fast_triangle(q0, q1, q2, q3, o, d, rmag, tmin, hit);
      if (res == 2) res = watertight_item(sc, q1, q2, q3, o, d, tmin, hit) ? 1 : 0;
      return res == 1 && any_hit;
    }
  }
};
using TraceState = TraceStateT<false>;

}  // namespace pt
