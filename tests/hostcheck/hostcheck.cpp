// hostcheck.cpp -- TEST HARNESS: compiles the integrator's device headers (pt_bvh.h,
// pt_shade.h, ...) and the host BVH builder with a plain C++ compiler so that the device
// LOGIC can be exercised against the oracle on machines without a GPU (and under gdb /
// host sanitizers).  It is never part of libpt_hip.so and nothing in the product links it.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#include "../../rust-pathtracer_amd/csrc/bvh_build.h"
#include "../../rust-pathtracer_amd/csrc/pt_bvh.h"
#include "../../rust-pathtracer_amd/csrc/pt_demosaic_tile.h"
#include "../../rust-pathtracer_amd/csrc/pt_shade.h"
#include "../../rust-pathtracer_amd/csrc/scene.h"

using namespace pt;

namespace {
HostScene g_hs;
BvhBuildResult g_bvh;
std::vector<ShadeRec> g_recs;
SceneView g_sv;
CameraView g_cam;
std::vector<GuideRegion> g_regions;
std::vector<GuideLobe> g_lobes;
std::vector<GuideRegionDev> g_regions_dev;
std::vector<GuideLobeDev> g_lobes_dev;
std::vector<GuideRegionDev> g_cells_dev;
std::vector<uint32_t> g_grid;
GuideView g_gv;
int g_guiding = 0;
bool g_ready = false;
int g_origin_skip = 1;  // pass the primitive a ray leaves to trace_ray (must not change any result)

void finish_scene() {
  const int nt = g_hs.num_tris();
  if (g_hs.instanced()) g_bvh = build_instanced(g_hs);
  else g_bvh = build_bvh8(g_hs.vertices.data(), g_hs.indices.data(), nt);
  g_recs.assign((size_t)(nt > 0 ? nt : 1), ShadeRec{});
  for (int i = 0; i < nt; ++i) {
    ShadeRec& r = g_recs[(size_t)i];
    const Material& m = g_hs.materials[(size_t)i];
    const uint32_t co = g_bvh.corner_order[(size_t)i];
    r.corner_order = co;
    for (int k = 0; k < 3; ++k) {
      r.v0[k] = g_hs.vertices[3 * (size_t)g_hs.indices[3 * (size_t)i + (co & 3u)] + k];
      r.v1[k] = g_hs.vertices[3 * (size_t)g_hs.indices[3 * (size_t)i + ((co >> 2) & 3u)] + k];
      r.v2[k] = g_hs.vertices[3 * (size_t)g_hs.indices[3 * (size_t)i + ((co >> 4) & 3u)] + k];
      r.base[k] = m.base[k];
      r.emission[k] = m.emission[k];
    }
    const float* n = &g_hs.normals[3 * (size_t)i];
    float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len < 1e-20f) len = 1e-20f;
    for (int k = 0; k < 3; ++k) r.ng[k] = n[k] * (1.0f / len);
    r.roughness = m.roughness; r.ior = m.ior; r.transmission = m.transmission;
    r.home_node = g_bvh.home_node[(size_t)i]; r.home_mask = g_bvh.home_mask[(size_t)i];
  }
  g_sv.nodes = g_bvh.nodes.data();
  g_sv.wide = nullptr;
  g_sv.tris = g_bvh.tris.data();
  g_sv.shade = g_recs.data();
  g_sv.inst_records = g_bvh.inst_records.empty() ? nullptr : g_bvh.inst_records.data();
  g_sv.num_tris = (uint32_t)nt;
  g_sv.num_items = (uint32_t)g_bvh.num_items;
  g_sv.vmag = g_bvh.vmag;
  for (int k = 0; k < 3; ++k) {
    g_sv.light_pos[k] = g_hs.light_pos[k]; g_sv.light_emit[k] = g_hs.light_emit[k]; g_sv.light_normal[k] = g_hs.light_normal[k];
  }
  g_sv.light_half[0] = g_hs.light_half[0]; g_sv.light_half[1] = g_hs.light_half[1];
  g_ready = true;
}
void ensure() {
  if (!g_ready) { g_hs = make_cornell(); finish_scene(); }
}
void set_camera(int w, int h) {
  // same arithmetic as renderer.cpp set_camera() with the default arguments
  const float eye[3] = {0.f, 3.f, 1.f}, look[3] = {0.f, 0.f, 1.f}, up[3] = {0.f, 0.f, 1.f};
  float W[3] = {look[0] - eye[0], look[1] - eye[1], look[2] - eye[2]};
  auto norm = [](float v[3]) { float l = sqrtf(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); float inv = 1.0f / l; v[0]*=inv; v[1]*=inv; v[2]*=inv; };
  norm(W);
  float U[3] = {W[1]*up[2]-W[2]*up[1], W[2]*up[0]-W[0]*up[2], W[0]*up[1]-W[1]*up[0]};
  norm(U);
  float V[3] = {U[1]*W[2]-U[2]*W[1], U[2]*W[0]-U[0]*W[2], U[0]*W[1]-U[1]*W[0]};
  const float th = tanf(45.0f * float(M_PI) / 180.0f * 0.5f), aspect = float(w) / float(h);
  for (int k = 0; k < 3; ++k) { g_cam.eye[k] = eye[k]; g_cam.w[k] = W[k]; g_cam.u[k] = th * aspect * U[k]; g_cam.v[k] = th * V[k]; }
}
}  // namespace

extern "C" {

void hc_scene_cornell() { g_hs = make_cornell(); finish_scene(); }
void hc_scene_synthetic(uint32_t seed, int n, int subdiv) { g_hs = make_synthetic(seed, n, subdiv); finish_scene(); }
// the same picture as instances of a room and a unit icosphere, the current scene written out flat, and a scene
// of caller meshes and placements (xforms: 12 floats per instance, object -> world rows)
void hc_scene_synthetic_instanced(uint32_t seed, int n, int subdiv) { g_hs = make_synthetic_instanced(seed, n, subdiv); finish_scene(); }
void hc_scene_flatten() { ensure(); g_hs = g_hs.flattened(); finish_scene(); }
int hc_scene_instances(const float* v, int nv, const uint32_t* idx, int nt, int n_inst, const float* xforms) {
  HostScene s = make_from_triangles(v, nv, idx, nt, nullptr);
  s.meshes.push_back(HostScene::MeshRange{0u, (uint32_t)nt});
  for (int i = 0; i < n_inst; ++i) {
    HostScene::Instance in{};
    in.mesh = 0;
    memcpy(in.xform, xforms + 12 * i, sizeof(in.xform));
    double inv[12];
    if (!invert_affine(in.xform, inv)) return -1;
    s.instances.push_back(in);
  }
  g_hs = s;
  finish_scene();
  return n_inst;
}
int hc_num_instances() { ensure(); return (int)g_hs.instances.size(); }
void hc_scene_triangles(const float* v, int nv, const uint32_t* idx, int nt, const float* m11) {
  g_hs = make_from_triangles(v, nv, idx, nt, m11); finish_scene();
}
void hc_scene_info(int* ntris, int* nnodes, int* depth) {
  ensure();
  *ntris = g_hs.num_tris(); *nnodes = (int)g_bvh.nodes.size(); *depth = g_bvh.depth;
}
void hc_scene_get(float* v, uint32_t* idx, float* nrm, float* m11) {
  ensure();
  if (v) memcpy(v, g_hs.vertices.data(), g_hs.vertices.size() * 4);
  if (idx) memcpy(idx, g_hs.indices.data(), g_hs.indices.size() * 4);
  if (nrm) memcpy(nrm, g_hs.normals.data(), g_hs.normals.size() * 4);
  if (m11) memcpy(m11, g_hs.materials.data(), g_hs.materials.size() * sizeof(Material));
}
void hc_set_origin_skip(int on) { g_origin_skip = on; }
// traverse the unpacked form of the nodes (what the fused kernels keep in LDS) instead of the packed one
void hc_set_wide_nodes(int on) {
  ensure();
  const bool use = on && !g_bvh.wide_nodes.empty();
  g_sv.wide = use ? g_bvh.wide_nodes.data() : nullptr;
  g_sv.home_wide = use ? g_bvh.home_wide.data() : nullptr;
  g_sv.wide_groups = g_bvh.wide_groups;
}
int hc_has_wide_nodes() { ensure(); return g_bvh.wide_nodes.empty() ? 0 : 1; }
int hc_wide_depth() { ensure(); return g_bvh.wide_depth; }
// rebuilds the current scene's tree with or without parallelogram items
void hc_set_pairing(int on) { bvh_set_pairing(on); ensure(); finish_scene(); }
int hc_num_items() { ensure(); return g_bvh.num_items; }
// FNV-1a over the built BVH arrays (nodes, triangle records, home masks)
unsigned long long hc_bvh_hash() {
  ensure();
  unsigned long long h = 1469598103934665603ull;
  auto feed = [&](const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
  };
  feed(g_bvh.nodes.data(), g_bvh.nodes.size() * sizeof(BvhNode));
  feed(g_bvh.tris.data(), g_bvh.tris.size() * sizeof(q4));
  feed(g_bvh.home_node.data(), g_bvh.home_node.size() * 4);
  feed(g_bvh.home_mask.data(), g_bvh.home_mask.size() * 4);
  return h;
}
int hc_num_vertices() { ensure(); return (int)(g_hs.vertices.size() / 3); }

void hc_trace_rays(int n, const float* o, const float* d, const float* tmin, const float* tmax, int any_hit,
                   int* prim, float* t, float* b1, float* b2) {
  ensure();
  for (int i = 0; i < n; ++i) {
    LocalStack st;
    Hit h = trace_ray(g_sv, mk3(o[3*i], o[3*i+1], o[3*i+2]), mk3(d[3*i], d[3*i+1], d[3*i+2]), tmin[i], tmax[i],
                      any_hit != 0, st, 64);
    if (any_hit) { prim[i] = h.prim >= 0 ? 1 : 0; }
    else {
      if (h.prim >= 0) to_input_order(g_sv.shade[h.prim].corner_order, h.b1, h.b2);
      prim[i] = h.prim; t[i] = h.t; b1[i] = h.b1; b2[i] = h.b2;
    }
  }
}

// Instanced scenes: the step-wise traversal the staged kernels run (TraceStateT<true>); prim is the index into the
// meshes' concatenated triangles, inst the instance of the hit.
void hc_trace_rays_instanced(int n, const float* o, const float* d, const float* tmin, const float* tmax, int any_hit,
                             int* prim, float* t, float* b1, float* b2, int* inst) {
  ensure();
  for (int i = 0; i < n; ++i) {
    LocalStack st;
    const f3 ro = mk3(o[3*i], o[3*i+1], o[3*i+2]), rd = mk3(d[3*i], d[3*i+1], d[3*i+2]);
    TraceStateT<true> ts;
    ts.begin(g_sv, ro, rd, tmax[i], 0u);
    while (!ts.done()) {
      if (ts.wants_node()) ts.node_step(g_sv, ro, rd, tmin[i], st, 64);
      if (ts.wants_tri() && ts.tri_step(g_sv, ro, rd, tmin[i], any_hit != 0, st, 64)) ts.finish();
    }
    Hit h = ts.hit;
    if (any_hit) { prim[i] = h.prim >= 0 ? 1 : 0; }
    else {
      if (h.prim >= 0) to_input_order(g_sv.shade[h.prim].corner_order, h.b1, h.b2);
      prim[i] = h.prim; t[i] = h.t; b1[i] = h.b1; b2[i] = h.b2; inst[i] = h.prim >= 0 ? (int)ts.hit_inst : -1;
    }
  }
}

void hc_guide_set(const float* regions8, int nr, const float* lobes8, int nl, const uint32_t* grid, const int* res,
                  const float* gmin, const float* cell, int enabled) {
  g_regions.resize((size_t)nr); g_lobes.resize((size_t)nl);
  memcpy(g_regions.data(), regions8, (size_t)nr * 32);
  memcpy(g_lobes.data(), lobes8, (size_t)nl * 32);
  g_grid.assign(grid, grid + (size_t)res[0] * res[1] * res[2]);
  g_regions_dev.resize((size_t)nr); g_lobes_dev.resize((size_t)nl);
  make_guide_records(g_regions.data(), (uint32_t)nr, g_lobes.data(), (uint32_t)nl, g_regions_dev.data(), g_lobes_dev.data());
  g_cells_dev.resize(g_grid.size());
  make_guide_cells(g_grid.data(), g_grid.size(), g_regions_dev.data(), (uint32_t)nr, g_cells_dev.data());
  g_gv.regions = g_regions_dev.data(); g_gv.lobes = g_lobes_dev.data(); g_gv.grid = g_grid.data();
  g_gv.cells = g_cells_dev.data();
  for (int k = 0; k < 3; ++k) {
    g_gv.res[k] = res[k]; g_gv.gmin[k] = gmin[k]; g_gv.cell[k] = cell[k];
    g_gv.inv_cell[k] = cell[k] > 0.f ? 1.0f / cell[k] : 0.0f;
  }
  g_guiding = enabled;
}

// The per-record step of the device kernel that bins training statistics (pt_guiding.h: bin_train_record), run over
// records in the layout guiding_map_train_samples returns: each gets the digest the capture would have written, then
// goes through the same validation, assignment and fixed-point addends; torn[i] != 0 spoils record i's digest.
void hc_bin_records(const float* rec10, int n, const float* bmin, const float* bmax, const int* res, const float* axes,
                    const int* torn, long long* table) {
  BinField f;
  for (int k = 0; k < 3; ++k) { f.bmin[k] = bmin[k]; f.bmax[k] = bmax[k]; f.res[k] = res[k]; }
  for (int i = 0; i < n; ++i) {
    TrainRec r;
    memcpy(&r, rec10 + 10 * (size_t)i, sizeof(r));
    r.is_delta = (r.is_delta & 1u) | train_digest(r);
    if (torn && torn[i]) r.contrib[1] += 1.0f;  // payload and digest no longer belong together
    uint32_t bin;
    long long add[kBinWords];
    if (!bin_train_record(r, f, axes, bin, add)) continue;
    for (int k = 0; k < kBinWords; ++k) table[(size_t)bin * kBinWords + k] += add[k];
  }
}

// The two forms of the passes over a region's lobes (pt_guiding.h): one load per lobe and pass, or the region's
// lobes loaded in batches.  lobes8: n <= 4 lobes in the snapshot layout; returns the chosen lobe and the mixture pdf
// of direction wi in both forms.
void hc_guide_lobe_passes(const float* lobes8, int n, float u, const float* wi, int* idx_loop, int* idx_batch,
                          float* pdf_loop, float* pdf_batch) {
  std::vector<GuideLobe> in((size_t)n);
  memcpy(in.data(), lobes8, (size_t)n * 32);
  GuideRegion reg{{0, 0, 0}, {1, 1, 1}, 0u, (uint32_t)n};
  GuideRegionDev rd;
  std::vector<GuideLobeDev> dev((size_t)n);
  make_guide_records(&reg, 1u, in.data(), (uint32_t)n, &rd, dev.data());
  const f3 w = mk3(wi[0], wi[1], wi[2]);
  *idx_loop = guiding_choose_lobe(dev.data(), (uint32_t)n, rd.sum, u);
  *pdf_loop = guiding_mixture_pdf(dev.data(), (uint32_t)n, rd.sum, w);
  GuideBatch gb;
  GuideAxes ga;
  guiding_load_batch(dev.data(), (uint32_t)n, gb);
  guiding_load_axes(dev.data(), (uint32_t)n, ga);
  *idx_batch = guiding_choose_lobe_batch(gb, (uint32_t)n, rd.sum, u);
  *pdf_batch = guiding_mixture_pdf_batch(ga, gb, (uint32_t)n, rd.sum, w);
}

struct HcRec {
  float radiance[3], throughput[3], origin[3], direction[3];
  unsigned long long seed; int depth, done; float prev_pdf_bsdf; int prev_pdf_valid; int px, py, sample, rng_dim;
};
void hc_shade(int n, HcRec* recs, const int* prim, const float* b1, const float* b2, int frame, int max_depth, int* shadow) {
  ensure();
  for (int i = 0; i < n; ++i) {
    HcRec& r = recs[i];
    PathState ps;
    ps.o = mk3(r.origin[0], r.origin[1], r.origin[2]);
    ps.d = mk3(r.direction[0], r.direction[1], r.direction[2]);
    ps.T = mk3(r.throughput[0], r.throughput[1], r.throughput[2]);
    ps.sum = mk3(0.f); ps.seed = r.seed; ps.prev_pdf = r.prev_pdf_bsdf;
    const uint32_t dim = r.rng_dim < 2 ? (uint32_t)r.rng_dim : 2u;
    ps.flags = ((uint32_t)r.depth & kDepthMask) | (dim << kDimShift) | (r.prev_pdf_valid ? kPrevValid : 0u) | kAlive;
    RandCtx rc = make_rand_ctx(r.px, r.py, r.sample, frame);
    Hit h{prim[i], 0.f, b1[i], b2[i]};
    if (h.prim >= 0) to_record_order(g_sv.shade[h.prim].corner_order, h.b1, h.b2);
    ShadowRay sr;
    ShadeParams sp{frame, max_depth, g_guiding};
    TrainVertex tv;
    shade_hit<true, false>(g_sv, g_gv, sp, rc, ps, h, sr, tv);
    if (sr.valid) {
      LocalStack st;
      Hit o = trace_ray(g_sv, sr.o, sr.d, 0.0f, sr.tmax, true, st, 64);
      if (o.prim < 0) ps.sum += sr.contrib;
    }
    shadow[i] = sr.valid ? 1 : 0;
    r.radiance[0] = ps.sum.x; r.radiance[1] = ps.sum.y; r.radiance[2] = ps.sum.z;
    r.throughput[0] = ps.T.x; r.throughput[1] = ps.T.y; r.throughput[2] = ps.T.z;
    r.origin[0] = ps.o.x; r.origin[1] = ps.o.y; r.origin[2] = ps.o.z;
    r.direction[0] = ps.d.x; r.direction[1] = ps.d.y; r.direction[2] = ps.d.z;
    r.seed = ps.seed; r.depth = (int)ps_depth(ps); r.done = (ps.flags & kAlive) ? 0 : 1;
    r.prev_pdf_bsdf = ps.prev_pdf; r.prev_pdf_valid = (ps.flags & kPrevValid) ? 1 : 0; r.rng_dim = (int)ps_dim(ps);
  }
}

// The per-pixel sample loop of the wavefront kernels, run sequentially per pixel.
// First number drawn from the stream of the sample range that starts at `begin`, for pixels 0..n-1.
void hc_first_draws(int frame, int begin, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    uint64_t s = chunk_seed(frame, (uint32_t)i, begin);
    out[i] = rnd(s);
  }
}

void hc_render(int w, int h, int sample_begin, int sample_count, int max_depth, int frame, int strict, int nthreads,
               float* out_sum_rgb, unsigned long long* rays) {
  ensure();
  set_camera(w, h);
  std::atomic<int> next_row{0};
  std::atomic<unsigned long long> r0{0}, r1{0};
  auto work = [&]() {
    unsigned long long a = 0, b = 0;
    for (;;) {
      int y = next_row.fetch_add(1);
      if (y >= h) break;
      for (int x = 0; x < w; ++x) {
        const uint32_t pixel = (uint32_t)(y * w + x);
        float acc[3] = {0, 0, 0};
        uint64_t seed = 0;
        for (int local = 0; local < sample_count; ++local) {
          const int s = strict ? (local & 3) : sample_begin + local;
          uint64_t start = strict ? ((local & 3) == 0 ? pixel_seed(frame, pixel) : seed)
                                  : (local == 0 ? chunk_seed(frame, pixel, sample_begin) : seed);
          PathState ps;
          RandCtx rc = make_rand_ctx(x, y, s, frame);
          generate_path(g_cam, rc, s, w, h, start, ps);
          ShadeParams sp{frame, max_depth, g_guiding};
          while (ps.flags & kAlive) {
            LocalStack st;
            Hit hit = trace_ray(g_sv, ps.o, ps.d, 0.0f, 1e16f, false, st, 64, g_origin_skip ? ps.origin : 0u);
            a++;
            ShadowRay sr;
            TrainVertex tv;
            shade_hit<true, false>(g_sv, g_gv, sp, rc, ps, hit, sr, tv);
            if (sr.valid) {
              b++;
              Hit o = trace_ray(g_sv, sr.o, sr.d, 0.0f, sr.tmax, true, st, 64, g_origin_skip ? sr.origin : 0u);
              if (o.prim < 0) ps.sum += sr.contrib;
            }
          }
          seed = ps.seed;
          acc[0] += ps.sum.x; acc[1] += ps.sum.y; acc[2] += ps.sum.z;
        }
        for (int k = 0; k < 3; ++k) out_sum_rgb[3 * (size_t)pixel + k] = acc[k];
      }
    }
    r0 += a; r1 += b;
  };
  std::vector<std::thread> th;
  for (int i = 0; i < (nthreads > 1 ? nthreads : 1); ++i) th.emplace_back(work);
  for (auto& t : th) t.join();
  if (rays) { rays[0] = r0; rays[1] = r1; }
}

}  // extern "C"

// The interior-tile routine of the gradient demosaic (pt_demosaic_tile.h), tile by tile as the kernel runs it:
// writes rgb (and covered = 1) for every pixel of a tile whose apron lies inside the image; returns their number.
namespace {
template <int P>
void demosaic_interior_tiles(const float* raw, int w, int h, float* out, unsigned char* covered, long* count) {
  const dm::TileGrid tg = dm::tile_grid(w, h);
  std::vector<float> planes(4 * (size_t)dm::kPlane);
  float* s_raw = planes.data();
  float* s_g = s_raw + dm::kPlane;
  float* s_rg = s_g + dm::kPlane;
  float* s_bg = s_rg + dm::kPlane;
  for (int by = 1; by < tg.iy1; ++by)
    for (int bx = 1; bx < tg.ix1; ++bx) {
      const int x0 = bx * dm::kTileW, y0 = by * dm::kTileH;
      std::fill(planes.begin(), planes.end(), 0.0f);
      for (int k = 0; k < dm::kPlane; ++k) {
        const int yy = k / dm::kStride, xx = k % dm::kStride;
        s_raw[k] = raw[(size_t)(y0 - dm::kApron + yy) * w + (x0 - dm::kApron + xx)];
      }
      for (int q = 0; q < dm::kCellsW * dm::kCellsH; ++q)
        dm::green_cell<P>(s_raw, s_g, (2 + 2 * (q / dm::kCellsW)) * dm::kStride + 2 + 2 * (q % dm::kCellsW));
      for (int q = 0; q < dm::kCellsW * dm::kCellsH; ++q)
        dm::chroma_cell<P>(s_raw, s_g, s_rg, s_bg, (2 + 2 * (q / dm::kCellsW)) * dm::kStride + 2 + 2 * (q % dm::kCellsW));
      for (int q = 0; q < dm::kOutCellsW * dm::kOutCellsH; ++q) {
        const int qy = q / dm::kOutCellsW, qx = q % dm::kOutCellsW;
        float row0[6], row1[6];
        dm::combine_cell(s_g, s_rg, s_bg, (dm::kApron + 2 * qy) * dm::kStride + dm::kApron + 2 * qx, row0, row1);
        for (int r = 0; r < 2; ++r) {
          const size_t i = (size_t)(y0 + 2 * qy + r) * w + (x0 + 2 * qx);
          memcpy(out + 3 * i, r ? row1 : row0, 6 * sizeof(float));
          covered[i] = covered[i + 1] = 1;
          *count += 2;
        }
      }
    }
}
}  // namespace
extern "C" long hc_demosaic_interior(const float* raw, int w, int h, int pattern, float* out, unsigned char* covered) {
  long count = 0;
  switch (pattern) {
    case 0: demosaic_interior_tiles<0>(raw, w, h, out, covered, &count); break;
    case 1: demosaic_interior_tiles<1>(raw, w, h, out, covered, &count); break;
    case 2: demosaic_interior_tiles<2>(raw, w, h, out, covered, &count); break;
    default: demosaic_interior_tiles<3>(raw, w, h, out, covered, &count); break;
  }
  return count;
}
// the tile grid the kernel launches for a w x h image: tiles across, down, interior bounds, border tiles
extern "C" void hc_demosaic_tile_grid(int w, int h, int* five) {
  const dm::TileGrid tg = dm::tile_grid(w, h);
  five[0] = tg.gw; five[1] = tg.gh; five[2] = tg.ix1; five[3] = tg.iy1; five[4] = tg.n_edge;
}
