// renderer.cpp -- host runtime and exported C ABI of libpt_hip.so.
//
// Mirrors the responsibilities of the reference's wrapper (optix/optix_wrapper.cpp): a lazily
// created singleton renderer per resolution, the built-in scene and camera, the render
// entry points and the guiding upload/readback calls -- with the OptiX pipeline replaced by
// the HIP wavefront kernels of pt_kernels.hip.
#include "renderer.h"

#include "pt_guiding.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace pt {
namespace {

// ---- global settings (the reference keeps the same kind of file-scope state:
// optix_wrapper.cpp:147-149, 923-925)
State* g_handle = nullptr;
int g_last_w = 0, g_last_h = 0;
int g_frame = 0;           // Params.frame of the most recent render call
bool g_frame_pinned = false;
int g_mode = -1;           // -1: read PT_MODE on first use
int g_pipeline = -1;       // -1: read PT_PIPELINE on first use
int g_device = -1;         // -1: read PT_DEVICE on first use
int g_guiding_enabled = 0;
uint32_t g_train_capacity = 0;
int g_grid_res[3] = {0, 0, 0};
std::vector<pgl_region> g_snap_regions;
std::vector<pgl_lobe> g_snap_lobes;
uint64_t g_snap_version = 0;
HostScene g_scene;
bool g_scene_init = false;
uint64_t g_scene_version = 1;
int g_timing_every = 0;
int g_gpus = -1;                 // sample shards per render call; -1: read PT_GPUS on first use
std::vector<State*> g_shards;    // shards 1, 2, ... (shard 0 is g_handle)
std::mutex g_stats_mutex;        // ray totals are added from every shard's host thread
KernelStat g_kstats[PT_K_COUNT];
uint64_t g_rays_total[2] = {0, 0};
int g_fused_bounces = 32;  // bounces per fused launch (1080p, 1024 spp: 16 389.7 ms, 24 382.0, 32 382.7, 48 382.2, 64 388.4)
int g_defer_shadow = getenv("PT_DEFER_SHADOW") ? atoi(getenv("PT_DEFER_SHADOW")) : -1;  // -1: chosen per scene
int g_one_launch = getenv("PT_PERSIST") ? (atoi(getenv("PT_PERSIST")) != 0) : 1;
int g_adaptive_round = getenv("PT_ADAPTIVE_ROUND") ? atoi(getenv("PT_ADAPTIVE_ROUND")) : -32;
int g_adaptive_chunks = getenv("PT_ADAPTIVE_CHUNKS") ? atoi(getenv("PT_ADAPTIVE_CHUNKS")) : 0;  // 0: as many as fit
constexpr size_t kTargetSlots = 8u << 20;       // path slots a renderer starts with (pixels x chunks); adaptive renders stay within them
constexpr size_t kFusedTargetSlots = 32u << 20;  // ... and what a fixed-count render of a scene that runs fused grows to

int env_choice(const char* name, const char* a, const char* b, int dflt) {
  const char* v = getenv(name);
  if (!v) return dflt;
  if (!strcmp(v, a)) return 0;
  if (!strcmp(v, b)) return 1;
  return dflt;
}
int mode() {
  if (g_mode < 0) g_mode = env_choice("PT_MODE", "strict", "converge", PT_MODE_CONVERGE);
  return g_mode;
}
// -1 = not chosen by the caller or PT_PIPELINE: pick per scene (see effective_pipeline)
int pipeline() {
  if (g_pipeline == -1) g_pipeline = env_choice("PT_PIPELINE", "staged", "fused", -2);  // -2: choose per scene
  return g_pipeline;
}
int device_index() {
  if (g_device < 0) {
    const char* v = getenv("PT_DEVICE");
    g_device = v ? atoi(v) : 0;
  }
  return g_device;
}
HostScene& scene() {
  if (!g_scene_init) {
    g_scene = make_cornell();
    g_scene_init = true;
  }
  return g_scene;
}

template <typename T>
T* dev_alloc(size_t n) {
  T* p = nullptr;
  HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)));
  return p;
}
template <typename T>
void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}
template <typename T>
void upload(T*& d, const std::vector<T>& h) {
  dev_free(d);
  d = dev_alloc<T>(h.size());
  if (!h.empty()) HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
}

// optix_wrapper.cpp:949-976 (and 813-833)
void set_camera(State& s, const float eye[3], const float lookat[3], const float up[3], float vfov_deg) {
  auto norm = [](float v[3]) {
    float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (len <= 0.f) { v[0] = v[1] = v[2] = 0.f; return; }
    float inv = 1.0f / len;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  };
  auto cross = [](const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  };
  float W[3] = {lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]};
  norm(W);
  float U[3];
  cross(W, up, U);
  norm(U);
  float V[3];
  cross(U, W, V);
  const float fovY = vfov_deg * float(M_PI) / 180.0f;
  const float aspect = (s.height == 0) ? 1.f : float(s.width) / float(s.height);
  const float th = std::tan(fovY * 0.5f);
  for (int k = 0; k < 3; ++k) {
    s.cam.eye[k] = eye[k];
    s.cam.w[k] = W[k];
    s.cam.u[k] = th * aspect * U[k];
    s.cam.v[k] = th * V[k];
  }
}

// The host-side BVH of the current scene, built once per scene version and shared by every
// renderer (a resolution change re-creates device state but not the tree).
BvhBuildResult g_bvh;
uint64_t g_bvh_version = 0;

// The current scene's tree (built on first use per scene version) and its per-primitive shading records.
const BvhBuildResult& current_bvh() {
  HostScene& hs = scene();
  if (g_bvh_version != g_scene_version) {
    if (hs.instanced()) g_bvh = build_instanced(hs);
    else g_bvh = build_bvh8(hs.vertices.data(), hs.indices.data(), hs.num_tris());
    g_bvh_version = g_scene_version;
  }
  return g_bvh;
}
std::vector<ShadeRec> shade_records(const BvhBuildResult& bvh) {
  HostScene& hs = scene();
  const int nt = hs.num_tris();
  std::vector<ShadeRec> recs((size_t)std::max(nt, 1));
  memset(recs.data(), 0, recs.size() * sizeof(ShadeRec));
  for (int i = 0; i < nt; ++i) {
    ShadeRec& r = recs[(size_t)i];
    const Material& m = hs.materials[(size_t)i];
    const uint32_t co = bvh.corner_order[(size_t)i];  // corners in the order the traversal records use
    r.corner_order = co;
    for (int k = 0; k < 3; ++k) {
      r.v0[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + (co & 3u)] + k];
      r.v1[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + ((co >> 2) & 3u)] + k];
      r.v2[k] = hs.vertices[3 * (size_t)hs.indices[3 * (size_t)i + ((co >> 4) & 3u)] + k];
      r.base[k] = m.base[k];
      r.emission[k] = m.emission[k];
    }
    // optix_device.cu:421: Ng = normalize3(normals[prim]), hoisted to upload time
    const float* n = &hs.normals[3 * (size_t)i];
    const float len = std::max(std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-20f);
    const float inv = 1.0f / len;
    for (int k = 0; k < 3; ++k) r.ng[k] = n[k] * inv;
    r.home_node = bvh.home_node[(size_t)i];
    r.home_mask = bvh.home_mask[(size_t)i];
    r.roughness = m.roughness;
    r.ior = m.ior;
    r.transmission = m.transmission;
  }
  return recs;
}

// Fill in everything of State::scene but the three device pointers.
void describe_scene(State& s, const BvhBuildResult& bvh) {
  HostScene& hs = scene();
  const int nt = hs.num_tris();
  s.scene.num_tris = (uint32_t)nt;
  s.scene.num_items = (uint32_t)bvh.num_items;
  s.scene.vmag = bvh.vmag;
  for (int k = 0; k < 3; ++k) {
    s.scene.light_pos[k] = hs.light_pos[k];
    s.scene.light_emit[k] = hs.light_emit[k];
    s.scene.light_normal[k] = hs.light_normal[k];
  }
  s.scene.light_half[0] = hs.light_half[0];
  s.scene.light_half[1] = hs.light_half[1];
  s.bvh_depth = std::max(bvh.depth, 1);
  s.num_nodes = (int)bvh.nodes.size();
  s.bvh_bytes = bvh.nodes.size() * sizeof(BvhNode) + bvh.tris.size() * sizeof(q4);
  s.num_instances = (int)(bvh.inst_records.size() / 6);
  s.scene_version = g_scene_version;
}

void upload_scene(State& s) {
  const BvhBuildResult& bvh = current_bvh();
  const std::vector<ShadeRec> recs = shade_records(bvh);
  upload(s.d_nodes, bvh.nodes);
  upload(s.d_tris, bvh.tris);
  upload(s.d_shade, recs);
  upload(s.d_wide, bvh.wide_nodes);
  upload(s.d_home_wide, bvh.home_wide);
  upload(s.d_inst, bvh.inst_records);
  s.scene.inst_records = bvh.inst_records.empty() ? nullptr : s.d_inst;
  s.scene.wide = nullptr;
  s.scene.wide_src = bvh.wide_nodes.empty() ? nullptr : s.d_wide;
  s.scene.home_wide = bvh.wide_nodes.empty() ? nullptr : s.d_home_wide;
  s.scene.wide_groups = bvh.wide_groups;
  s.scene.wide_one_node = (!bvh.wide_nodes.empty() && bvh.wide_depth == 1) ? 1u : 0u;
  s.scene.nodes = s.d_nodes;
  s.scene.tris = s.d_tris;
  s.scene.shade = s.d_shade;
  describe_scene(s, bvh);
}

// optix_wrapper.cpp:1168-1253: convert the snapshot, build the region grid on the host, upload.
void upload_guide(State& s) {
  s.guide_version = g_snap_version;
  s.guide_valid = false;
  const uint32_t nr = (uint32_t)g_snap_regions.size(), nl = (uint32_t)g_snap_lobes.size();
  if (nr == 0) return;
  std::vector<GuideRegion> hreg(nr);
  std::vector<GuideLobe> hlobe(nl);
  static_assert(sizeof(GuideRegion) == sizeof(pgl_region) && sizeof(GuideLobe) == sizeof(pgl_lobe), "layout");
  memcpy(hreg.data(), g_snap_regions.data(), nr * sizeof(GuideRegion));
  if (nl) memcpy(hlobe.data(), g_snap_lobes.data(), nl * sizeof(GuideLobe));
  float bmin[3] = {1e30f, 1e30f, 1e30f}, bmax[3] = {-1e30f, -1e30f, -1e30f};
  for (uint32_t i = 0; i < nr; ++i)
    for (int k = 0; k < 3; ++k) {
      bmin[k] = std::fmin(bmin[k], hreg[i].bmin[k]);
      bmax[k] = std::fmax(bmax[k], hreg[i].bmax[k]);
    }
  int res[3];
  for (int k = 0; k < 3; ++k) res[k] = g_grid_res[k] <= 0 ? 1 : g_grid_res[k];
  float cell[3];
  for (int k = 0; k < 3; ++k) cell[k] = (bmax[k] - bmin[k]) / (float)res[k];
  std::vector<uint32_t> grid((size_t)res[0] * (size_t)res[1] * (size_t)res[2], 0xFFFFFFFFu);
  for (uint32_t r = 0; r < nr; ++r) {
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      // a field that is flat along an axis has zero-size cells there: everything falls into cell 0
      lo[k] = cell[k] > 0.f ? (int)std::floor((hreg[r].bmin[k] - bmin[k]) / cell[k]) : 0;
      hi[k] = cell[k] > 0.f ? (int)std::floor((hreg[r].bmax[k] - bmin[k]) / cell[k]) : 0;
      lo[k] = std::max(0, std::min(res[k] - 1, lo[k]));
      hi[k] = std::max(0, std::min(res[k] - 1, hi[k]));
    }
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int x = lo[0]; x <= hi[0]; ++x) {
          const size_t idx = ((size_t)z * (size_t)res[1] + (size_t)y) * (size_t)res[0] + (size_t)x;
          if (grid[idx] == 0xFFFFFFFFu) grid[idx] = r;  // first region wins
        }
  }
  std::vector<GuideRegionDev> dreg(nr);
  std::vector<GuideLobeDev> dlobe(nl);
  make_guide_records(hreg.data(), nr, hlobe.data(), nl, dreg.data(), dlobe.data());
  upload(s.d_regions, dreg);
  upload(s.d_lobes, dlobe);
  upload(s.d_grid, grid);
  std::vector<GuideRegionDev> cells(grid.size());
  make_guide_cells(grid.data(), grid.size(), dreg.data(), nr, cells.data());
  upload(s.d_cells, cells);
  s.guide.cells = s.d_cells;
  s.guide.regions = s.d_regions;
  s.guide.lobes = s.d_lobes;
  s.guide.grid = s.d_grid;
  for (int k = 0; k < 3; ++k) {
    s.guide.res[k] = res[k];
    s.guide.gmin[k] = bmin[k];
    s.guide.cell[k] = cell[k];
    s.guide.inv_cell[k] = cell[k] > 0.f ? 1.0f / cell[k] : 0.0f;
  }
  s.guide_valid = nl > 0;
}

// Chunks (slots per pixel) a renderer of px pixels has room for.
int max_chunks_for(size_t px) {
  static const size_t target_slots = getenv("PT_TARGET_SLOTS") ? (size_t)atoll(getenv("PT_TARGET_SLOTS")) : kTargetSlots;
  return (int)std::max<size_t>(1, std::min<size_t>(16, target_slots / std::max<size_t>(px, 1)));
}

// Slots per pixel an adaptive render of a scene that runs fused may take: 16 M slots (1080p: 8 chunks, of 4 samples
// each in a round of 32; 4 / 8 / 16 chunks at threshold 0.01: 334.3 / 327.9 / 330.2 ms -- shorter slots shorten the
// end of every round, where waves run on with a few lanes, until refilling costs more than that).
int adaptive_room(size_t px) { return (int)std::max<size_t>(1, std::min<size_t>(16, ((size_t)16u << 20) / std::max<size_t>(px, 1))); }

// The per-slot arrays for `chunks` slots per pixel (whatever they held is gone).
constexpr size_t kSlotBytes = 9 * 16 + 5 * 4;  // nine 16-byte words, started, hit_inst and three queue entries
void alloc_slots(State& s, int chunks) {
  dev_free(s.st_o); dev_free(s.st_d); dev_free(s.st_s); dev_free(s.st_m); dev_free(s.acc); dev_free(s.hits);
  dev_free(s.sh_o); dev_free(s.sh_d); dev_free(s.sh_c); dev_free(s.sh_count); dev_free(s.started); dev_free(s.hit_inst);
  dev_free(s.queue[0]); dev_free(s.queue[1]); dev_free(s.queue_tmp); dev_free(s.tiles);
  s.max_chunks = chunks;
  const size_t slots = (size_t)s.width * (size_t)s.height * (size_t)chunks;
  s.st_o = dev_alloc<q4>(slots); s.st_d = dev_alloc<q4>(slots); s.st_s = dev_alloc<q4>(slots);
  s.st_m = dev_alloc<u4>(slots); s.acc = dev_alloc<q4>(slots); s.hits = dev_alloc<q4>(slots);
  s.sh_o = dev_alloc<q4>(slots); s.sh_d = dev_alloc<q4>(slots); s.sh_c = dev_alloc<q4>(slots);
  s.sh_count = dev_alloc<uint32_t>(slots / 256 + 2);
  s.started = dev_alloc<int>(slots);
  s.hit_inst = dev_alloc<uint32_t>(slots);
  s.queue[0] = dev_alloc<uint32_t>(slots);
  s.queue[1] = dev_alloc<uint32_t>(slots);
  s.queue_tmp = dev_alloc<uint32_t>(slots);
  s.tiles = dev_alloc<uint32_t>(slots / 1024 + 2);
  if (s.acc2) {  // the adaptive renders' sum of squares is per slot too
    dev_free(s.acc2);
    s.acc2 = dev_alloc<q4>(slots);
  }
}

// How many of `want` slots per pixel a renderer that holds `have` may grow to: what a quarter of the memory that
// is free (its own arrays counted as free) has room for, and never fewer than it holds.
int chunks_that_fit(size_t px, int want, int have, size_t free_bytes) {
  const size_t own = px * (size_t)have * kSlotBytes;
  while (want > have && px * (size_t)want * kSlotBytes > (free_bytes + own) / 4) --want;
  return want;
}

// The state arrays hold `chunks` slots per pixel after this, or as many as a quarter of the free memory allows.
int fit_chunks(State& s, int chunks) {
  if (s.borrowed) return std::min(chunks, s.max_chunks);
  if (chunks > s.max_chunks) {
    const size_t px = std::max<size_t>((size_t)s.width * (size_t)s.height, 1);
    size_t free_b = 0, total_b = 0;
    HIP_CHECK(hipSetDevice(s.device));
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    chunks = chunks_that_fit(px, chunks, s.max_chunks, free_b);
    if (chunks > s.max_chunks) {
      HIP_CHECK(hipStreamSynchronize(s.stream));
      alloc_slots(s, chunks);
    }
  }
  return chunks;
}

// Trees up to this size run fused by default, larger ones staged (measured on spheres in the room at 1080p, 256
// samples: 0.14 MiB of tree fused 9.9 against staged 7.7 Grays/s, 0.55 MiB 5.5 against 6.2, 1.4 MiB 5.0 against 5.8).
constexpr size_t kFusedTreeBytes = (size_t)256u << 10;
bool staged_by_default(const State& s) { return s.scene.inst_records != nullptr || s.bvh_bytes > kFusedTreeBytes; }

// Chunks (slots per pixel) of a fixed-count render.  Scenes that run fused: up to 8 chunks of at least 32 samples (the
// last one then cut into short ones, planned_layout).  With the launch-after-launch kernel more slots were better
// (1080p, 1024 spp, queue compacted by every launch: 8 M slots 404.7 ms, 16 M 395.7, 32 M 390.2: the more slots, the
// later the launches run out of full waves); the one-launch kernel only cares how long a slot is -- every change of
// slot costs a lane a few dozen instructions, the end of the render is taken care of by the short final slots --
// and runs 8 chunks of 128 samples 0.5 % faster than 16 of 64 (297.5 against 299.4 ms; 4 of 256 the same).  Scenes that run staged (larger trees, instances) do better the more of
// their samples are in flight at once -- every launch is one bounce of every live path, so fewer, larger launches
// leave fewer half-empty waves behind: up to 64 chunks of as little as one sample, 144 M slots (22 GB of the
// GPU's 288) at most and never more than a quarter of what is free.  The state arrays grow when a render needs it.
int planned_chunks(size_t px, int sample_count, bool staged_scene) {
  static const int chunk_override = getenv("PT_CHUNKS") ? atoi(getenv("PT_CHUNKS")) : 0;
  static const size_t target_env = getenv("PT_TARGET_SLOTS") ? (size_t)atoll(getenv("PT_TARGET_SLOTS")) : 0;
  static const int min_chunk_env = getenv("PT_MIN_CHUNK_SAMPLES") ? std::max(atoi(getenv("PT_MIN_CHUNK_SAMPLES")), 1) : 0;
  px = std::max<size_t>(px, 1);
  static const int most_env = getenv("PT_MAX_CHUNKS") ? atoi(getenv("PT_MAX_CHUNKS")) : 0;
  const int most = most_env > 0 ? most_env : (staged_scene ? 64 : 8);
  const size_t target = target_env ? target_env : (staged_scene ? (size_t)144 << 20 : kFusedTargetSlots);
  const int min_chunk = min_chunk_env ? min_chunk_env : (staged_scene ? 1 : 32);
  int chunks = (int)std::max<size_t>(1, std::min<size_t>((size_t)most, target / px));
  if (chunk_override > 0) chunks = std::min(chunk_override, most);
  while (chunks > 1 && sample_count / chunks < min_chunk) --chunks;
  return chunks;
}
// How a pixel's sample range is cut into slots: long_n chunks of long_len samples, then chunks of short_len.
struct ChunkPlan {
  int chunks, long_n, long_len, short_len;
};
ChunkPlan uniform_plan(int sample_count, int n) {
  n = std::max(n, 1);
  const int len = (sample_count + n - 1) / n;
  return ChunkPlan{n, n, len, len};
}
// A fixed-count render of a scene that runs fused ends with short slots: the one-launch kernel is over when the last
// lane has finished the slot it holds, and from the moment the queue is empty the GPU runs emptier and emptier for
// as long as the longest slots take (1080p, section clock of the profiling build: 9.5 ms of 369 with 64-sample slots
// everywhere, 5 ms with 32-sample slots, 1.5 ms with 8-sample slots -- which cost as much again in slot changes when
// every slot is that short).  So the last of the n chunks is cut into PT_TAIL_SPLIT (4) short ones, which the queue,
// chunk after chunk, lists last.
ChunkPlan layout_from(int sample_count, int n, bool split_tail) {
  static const int split = getenv("PT_TAIL_SPLIT") ? atoi(getenv("PT_TAIL_SPLIT")) : 4;
  const ChunkPlan u = uniform_plan(sample_count, n);
  if (!split_tail || n < 2 || split < 2) return u;
  const int rem = sample_count - (n - 1) * u.long_len;  // what the last chunk holds
  const int short_len = std::max((rem + split - 1) / split, 4);
  const int n_short = rem > 0 ? (rem + short_len - 1) / short_len : 0;
  if (n_short < 2) return u;
  return ChunkPlan{n - 1 + n_short, n - 1, u.long_len, short_len};
}
ChunkPlan planned_layout(size_t px, int sample_count, bool staged_scene) {
  return layout_from(sample_count, planned_chunks(px, sample_count, staged_scene), !staged_scene);
}
ChunkPlan plan_for_render(State& s, int sample_count, bool strict) {
  if (strict) return uniform_plan(sample_count, 1);
  const size_t px = (size_t)s.width * (size_t)s.height;
  const bool staged_scene = !s.borrowed && staged_by_default(s);
  const int n = planned_chunks(px, sample_count, staged_scene);
  if (s.borrowed) {  // arrays of a fixed size: the plan of the same render with arrays that grow, as far as it fits
    const int n_fit = std::min(n, s.max_chunks);
    const ChunkPlan want = layout_from(sample_count, n_fit, true);
    return want.chunks <= s.max_chunks_split ? want : uniform_plan(sample_count, n_fit);
  }
  const ChunkPlan want = layout_from(sample_count, n, !staged_scene);
  const int got = fit_chunks(s, want.chunks);
  if (got == want.chunks) return want;
  return uniform_plan(sample_count, std::min(got, n));  // less room than that
}

// optix_wrapper.cpp:784-920
State* make_state(int W, int H, int device = -1) {
  State* s = new State();
  s->device = device >= 0 ? device : device_index();
  HIP_CHECK(hipSetDevice(s->device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, s->device));
  s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  s->width = W;
  s->height = H;
  const float eye[3] = {0.f, 3.f, 1.f}, look[3] = {0.f, 0.f, 1.f}, up[3] = {0.f, 0.f, 1.f};
  set_camera(*s, eye, look, up, 45.f);
  HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  upload_scene(*s);
  const size_t px = (size_t)W * (size_t)H;
  // A pixel's sample range may be rendered as several chunks side by side (one slot each);
  // state is sized for the most chunks a render of this resolution will use.
  s->base_chunks = max_chunks_for(px);
  alloc_slots(*s, s->base_chunks);
  s->counters = dev_alloc<uint32_t>(CNT_WORDS);
  s->ray_counters = dev_alloc<unsigned long long>(2);
  HIP_CHECK(hipMemset(s->ray_counters, 0, 2 * sizeof(unsigned long long)));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s->h_counters), 3 * CNT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
  s->d_out_rgb = dev_alloc<float>(px * 3);
  s->d_out_bayer = dev_alloc<float>(px);
  for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreateWithFlags(&s->ev_slot[i], hipEventDisableTiming));
  HIP_CHECK(hipEventCreate(&s->ev_begin));
  HIP_CHECK(hipEventCreate(&s->ev_end));
  if (g_train_capacity > 0) {  // optix_wrapper.cpp:874-880
    s->train_capacity = g_train_capacity;
    s->d_train = dev_alloc<TrainRec>(g_train_capacity);
    s->d_train_idx = dev_alloc<uint32_t>(1);
    HIP_CHECK(hipMemset(s->d_train_idx, 0, sizeof(uint32_t)));
    HIP_CHECK(hipMemset(s->d_train, 0xff, sizeof(TrainRec) * (size_t)g_train_capacity));  // every slot: never written
  }
  if (!g_snap_regions.empty()) upload_guide(*s);
  return s;
}

void drop_shards() {
  for (State* sh : g_shards) delete sh;
  g_shards.clear();
}

// optix_wrapper.cpp:1087-1100
State& ensure_handle(int w, int h) {
  if (!g_handle || w != g_last_w || h != g_last_h) {
    delete g_handle;
    drop_shards();
    g_handle = make_state(w, h);
    g_last_w = w;
    g_last_h = h;
    if (!g_frame_pinned) g_frame = 0;  // a fresh renderer starts at frame 0 (optix_wrapper.cpp:907)
  }
  State& s = *g_handle;
  HIP_CHECK(hipSetDevice(s.device));
  if (s.scene_version != g_scene_version) upload_scene(s);
  if (s.guide_version != g_snap_version) upload_guide(s);
  return s;
}

struct Timer {
  State& s;
  int kernel;
  bool on;
  hipEvent_t a = nullptr, b = nullptr;
  Timer(State& st, int k, bool enabled, uint64_t work) : s(st), kernel(k), on(enabled && !st.quiet) {
    if (st.quiet) return;  // shards other than 0 run on other host threads: they leave the statistics alone
    g_kstats[k].launches++;
    g_kstats[k].work += work;
    if (on) {
      HIP_CHECK(hipEventCreate(&a));
      HIP_CHECK(hipEventCreate(&b));
      HIP_CHECK(hipEventRecord(a, s.stream));
    }
  }
  ~Timer() {
    if (on) {
      HIP_CHECK(hipEventRecord(b, s.stream));
      s.timed.push_back(TimedLaunch{kernel, a, b});
    }
  }
};

void resolve_timers(State& s) {
  for (TimedLaunch& t : s.timed) {
    float ms = 0.f;
    HIP_CHECK(hipEventSynchronize(t.stop));
    HIP_CHECK(hipEventElapsedTime(&ms, t.start, t.stop));
    g_kstats[t.kernel].total_ms += ms;
    g_kstats[t.kernel].timed++;
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  s.timed.clear();
}

RenderParams make_params(State& s, int sample_begin, int sample_count, int max_depth, int frame, bool strict, bool grow = true) {
  RenderParams p;
  memset(&p, 0, sizeof(p));
  p.scene = s.scene;
  p.guide = s.guide;
  p.cam = s.cam;
  p.width = s.width;
  p.height = s.height;
  p.frame = frame;
  p.max_depth = std::max(1, std::min(max_depth, 255));
  p.sample_begin = sample_begin;
  p.sample_count = sample_count;
  p.strict = strict ? 1 : 0;
  p.npx = (uint32_t)((size_t)s.width * (size_t)s.height);
  auto inv32 = [](uint32_t b) { return b > 1u ? (uint32_t)(((1ull << 32) + b - 1u) / b) : 0u; };  // 0: divide by one
  p.npx_inv = inv32(p.npx);
  p.width_inv = inv32((uint32_t)s.width);
  // chunks: strict mode replays one stream per pixel; converge mode splits the range (plan_for_render)
  const ChunkPlan plan = grow ? plan_for_render(s, sample_count, strict) : uniform_plan(sample_count, 1);
  p.chunks = plan.chunks;
  p.chunk_spp = plan.long_len;
  p.chunk_long = plan.long_n;
  p.chunk_short = plan.short_len;
  p.guiding = (g_guiding_enabled && s.guide_valid) ? 1 : 0;
  p.stack_depth = s.bvh_depth;
  p.scene_nodes = (uint32_t)s.num_nodes;
  {
    static const int lds_scene = getenv("PT_LDS_SCENE") ? atoi(getenv("PT_LDS_SCENE")) : 1;
    p.scene_in_lds = lds_scene;
    // Shadow rays from a per-wave queue pay where a traversal is long enough to be worth moving between lanes and
    // the scene is small enough to stay close: trees of more than one node up to about a thousand triangles
    // (spheres in the room at 1080p, fused kernel, queued against in place: 92 triangles 17.3 against 15.4 Grays/s,
    // 652: 16.9 against 15.4, 1 612: 9.8 against 10.1, 3 212: 9.4 against 9.8; the one-node built-in scene 52.8
    // against 55.2) -- unless the scene fits into LDS beside what the in-place kernel keeps there, which the queues'
    // 29 KB per block rule out: then in place and from LDS (32 triangles in 5 nodes: 28.3 against 24.8 Grays/s).
    p.defer_shadow = 0;
    if (g_defer_shadow >= 0) p.defer_shadow = g_defer_shadow;
    else if (!s.scene.wide_one_node && s.scene.num_tris <= 1024u && !persist_in_place_stages_scene(p, 256)) p.defer_shadow = 1;
    static const int refill = getenv("PT_REFILL_LANES") ? atoi(getenv("PT_REFILL_LANES")) : 16;
    p.refill_lanes = std::min(std::max(refill, 1), 64);
    static const int shared_env = getenv("PT_SHARED_RANGES") ? atoi(getenv("PT_SHARED_RANGES")) : -1;
    // long rays: a tree beyond the caches, or a thousand instances (whose tree is small however much it stands for);
    // the kernels use it in launches large enough for ranges of 512 rays and more
    p.shared_ranges = shared_env >= 0 ? shared_env : ((s.bvh_bytes > (size_t)(4u << 20) || s.num_instances >= 1000) ? 1 : 0);
  }
  p.st_o = s.st_o; p.st_d = s.st_d; p.st_s = s.st_s; p.st_m = s.st_m; p.acc = s.acc;
  p.started = s.started;
  p.queue[0] = s.queue[0]; p.queue[1] = s.queue[1];
  p.queue_tmp = s.queue_tmp;
  p.tiles = s.tiles;
  p.counters = s.counters;
  p.ray_counters = s.ray_counters;
  p.hits = s.hits;
  p.hit_inst = s.hit_inst;
  p.sh_o = s.sh_o; p.sh_d = s.sh_d; p.sh_c = s.sh_c;
  p.sh_count = s.sh_count;
  // optix_device.cu:432: training records are captured whenever the ring exists
  p.train = s.d_train;
  p.train_write_idx = s.d_train_idx;
  p.train_capacity = s.train_capacity;
  return p;
}

// The batch loop (optix_wrapper.cpp:1002-1021) re-thought for a wavefront: iterations are
// enqueued back to back and the live-slot count is read back once per batch of iterations,
// one batch behind the GPU, instead of one blocking sync per 4 samples.
struct RenderLoop {
  State& s;
  RenderParams p;
  LaunchCfg cfg, cfg_fused, cfg_persist;
  bool fused;
  bool tail_fused = false;
  uint64_t iter = 0;

  RenderLoop(State& st, const RenderParams& params) : s(st), p(params) {
    const size_t slots = (size_t)p.npx * (size_t)p.chunks;
    cfg.threads = 256;
    // grid-stride kernels: several blocks per resident slot even out the tail.  The staged kernels hand
    // every wave one contiguous range of the queue and do better with fewer, longer ranges.
    static const int env_blocks = getenv("PT_BLOCKS_PER_CU") ? atoi(getenv("PT_BLOCKS_PER_CU")) : 0;
    auto blocks_for = [&](int per_cu) {
      const int bpc = std::max(env_blocks > 0 ? env_blocks : per_cu, 1);
      return std::max((int)std::min<size_t>((slots + 255) / 256, (size_t)s.num_cus * (size_t)bpc), 1);
    };
    cfg.blocks = blocks_for(16);
    cfg.stream = s.stream;
    cfg_fused = cfg;
    cfg_fused.blocks = blocks_for(32);
    // the one-launch kernel: as many blocks as stay resident (4 per CU at 128 registers and under 40 KB of LDS each)
    static const int persist_bpc = getenv("PT_PERSIST_BLOCKS_PER_CU") ? std::max(atoi(getenv("PT_PERSIST_BLOCKS_PER_CU")), 1) : 4;
    cfg_persist = cfg;
    cfg_persist.blocks = std::max((int)std::min<size_t>((slots + 255) / 256, (size_t)s.num_cus * (size_t)persist_bpc), 1);
    // Small scenes are instruction-bound: keep paths in registers (fused).  Scenes whose BVH lives in
    // HBM are latency-bound: the lean staged kernels run at twice the occupancy and win.
    const int chosen = pipeline();
    fused = chosen >= 0 ? chosen == PT_PIPELINE_FUSED : s.bvh_bytes <= kFusedTreeBytes;
    // Auto mode only: once few paths are left, staged iterations (one bounce, four launches) are mostly
    // launch overhead; the fused kernel then finishes the stragglers, many bounces per launch.  Both
    // pipelines work on the same state and give the same bits.
    tail_fused = chosen < 0 && !fused;
    if (s.scene.inst_records) {  // instances are walked by the staged traversal kernels only
      fused = false;
      tail_fused = false;
    }
    if (s.borrowed) {  // no hit / shadow records exist: the fused kernels only
      fused = true;
      tail_fused = false;
    }
  }

  // Clear the per-slot state of a new render.
  void reset_slots() {
    const size_t slots = (size_t)p.npx * (size_t)p.chunks;
    HIP_CHECK(hipMemsetAsync(s.st_m, 0, slots * sizeof(u4), s.stream));
    HIP_CHECK(hipMemsetAsync(s.st_s, 0, slots * sizeof(q4), s.stream));
    HIP_CHECK(hipMemsetAsync(s.acc, 0, slots * sizeof(q4), s.stream));
    HIP_CHECK(hipMemsetAsync(s.started, 0, slots * sizeof(int), s.stream));
    if (p.acc2) HIP_CHECK(hipMemsetAsync(p.acc2, 0, slots * sizeof(q4), s.stream));
  }

  // Queue 0 holds `n` slot ids (already written on the stream); run until every slot has finished.
  void drain(uint32_t n) {
    if (n == 0) return;
    uint32_t* init = s.h_counters + 2 * CNT_WORDS;  // pinned staging, third slot
    memset(init, 0, CNT_WORDS * sizeof(uint32_t));
    init[CNT_Q0] = n;
    HIP_CHECK(hipMemcpyAsync(s.counters, init, CNT_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
    static const int fused_bounces = getenv("PT_FUSED_BOUNCES") ? atoi(getenv("PT_FUSED_BOUNCES")) : g_fused_bounces;
    // Renders without per-sample sums take the whole queue in one launch; pt_set_one_launch(0) / PT_PERSIST=0: they
    // too go launch after launch (32 bounces of every slot per launch, queue compacted).
    const bool persist = g_one_launch != 0 && bounce_persist_usable(p);
    bool use_fused = fused, done = false;
    int qin = 0, slot = 0;
    int cin = CNT_Q0;      // the counter that holds the length of queue[qin] (the staged kernels keep it in counters[qin])
    int live[2] = {0, 0};  // per read-back slot: the counter that held the number of live slots when it was taken
    bool have_prev = false;
    for (;;) {
      const int iters_per_batch = use_fused ? 8 : 16;
      for (int b = 0; b < iters_per_batch; ++b, ++iter) {
        const bool timed = !s.borrowed && g_timing_every > 0 && (iter % (uint64_t)g_timing_every) == 0;
        if (use_fused && persist) {
          // One launch takes the whole queue to its end (k_bounce_persist): nothing to read back afterwards.
          HIP_CHECK(hipMemsetAsync(s.counters + CNT_GRAB, 0, sizeof(uint32_t), s.stream));
          Timer t(s, PT_K_BOUNCE, timed, 0);
          launch_bounce_persist(p, qin, cin, cfg_persist);
          done = true;
          break;
        }
        if (use_fused) {
          // The slot queue is compacted by every launch (PT_COMPACT_EVERY=n: by every n-th, a divisor of the batch
          // length): a finished slot left in the queue costs more than its state read -- it idles a lane of a wave
          // that runs all its bounces for the live slots beside it.  Adaptive renders, whose every round ends with
          // such a tail, gain 5 %, a fixed-count render 1 %.
          static const int compact_every = getenv("PT_COMPACT_EVERY") ? std::max(atoi(getenv("PT_COMPACT_EVERY")), 1) : 1;
          const int compact = ((b + 1) % compact_every == 0 || b == iters_per_batch - 1) ? 1 : 0;
          const int cout = next_queue_counter(cin);
          const QueueCounters cnt{cin, cout, compact ? next_queue_counter(cout) : cout};
          Timer t(s, PT_K_BOUNCE, timed, 0);
          launch_bounce(p, qin, std::max(fused_bounces, 1), compact, cnt, cfg_fused);
          if (compact) {
            qin ^= 1;
            cin = cout;
          }
          continue;
        }
        { Timer t(s, PT_K_GENERATE, timed, 0); launch_generate(p, qin, cfg); }
        { Timer t(s, PT_K_EXTEND, timed, 0); launch_extend(p, qin ^ 1, cfg); }
        { Timer t(s, PT_K_SHADE, timed, 0); launch_shade(p, qin ^ 1, cfg); }
        { Timer t(s, PT_K_OCCLUDE, timed, 0); launch_occlude(p, qin ^ 1, cfg); }
        qin ^= 1;
        cin = qin;  // CNT_Q0 / CNT_Q1; the other one has been cleared by k_occlude, CNT_Q2 is untouched
      }
      if (done) break;
      // Only the counter of the queue the next launch reads says how many slots are live: the others
      // hold earlier lengths until a later launch clears them.
      if (s.borrowed) {  // no pinned memory and no events: a blocking copy after every batch
        uint32_t c[CNT_WORDS];
        HIP_CHECK(hipMemcpy(c, s.counters, sizeof(c), hipMemcpyDeviceToHost));
        if (c[cin] == 0u) break;
        continue;
      }
      HIP_CHECK(hipMemcpyAsync(s.h_counters + slot * CNT_WORDS, s.counters, CNT_WORDS * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, s.stream));
      HIP_CHECK(hipEventRecord(s.ev_slot[slot], s.stream));
      live[slot] = cin;
      if (have_prev) {
        const int prev = slot ^ 1;
        HIP_CHECK(hipEventSynchronize(s.ev_slot[prev]));
        const uint32_t left = s.h_counters[prev * CNT_WORDS + live[prev]];
        if (left == 0u) break;
        if (tail_fused && left < n / 16u) use_fused = true;
      }
      have_prev = true;
      slot ^= 1;
    }
  }

  void begin() {
    if (!s.borrowed) HIP_CHECK(hipEventRecord(s.ev_begin, s.stream));
  }
  void end() {
    if (!s.borrowed) HIP_CHECK(hipEventRecord(s.ev_end, s.stream));
    HIP_CHECK(hipStreamSynchronize(s.stream));
    float ms = 0.f;
    if (!s.borrowed) HIP_CHECK(hipEventElapsedTime(&ms, s.ev_begin, s.ev_end));
    s.last_render_ms = ms;
    resolve_timers(s);
    unsigned long long rc[2];
    HIP_CHECK(hipMemcpy(rc, s.ray_counters, sizeof(rc), hipMemcpyDeviceToHost));
    {
      std::lock_guard<std::mutex> lock(g_stats_mutex);
      g_rays_total[0] += rc[0];
      g_rays_total[1] += rc[1];
    }
    HIP_CHECK(hipMemset(s.ray_counters, 0, sizeof(rc)));
  }
};

void render_core(State& s, const RenderParams& params) {
  RenderLoop loop(s, params);
  // a fixed-count render that runs fused from its first launch: every slot is untouched when its one launch takes it,
  // and none runs again
  loop.p.fresh_queue = loop.fused ? 1 : 0;
  loop.p.last_pass = 1;
  s.last_chunks = loop.p.chunks;
  s.adaptive_rounds = 0;
  const size_t slots = (size_t)loop.p.npx * (size_t)loop.p.chunks;
  loop.begin();
  loop.reset_slots();
  launch_iota(s.queue[0], (uint32_t)slots, s.stream);
  if (params.sample_count > 0) loop.drain((uint32_t)slots);
  loop.end();
}
void render_core(State& s, int sample_begin, int sample_count, int max_depth, int frame, bool strict) {
  render_core(s, make_params(s, sample_begin, sample_count, max_depth, frame, strict));
}

// Adaptive sampling (optix_wrapper.cpp:1002-1021 with optix_device.cu:1011-1060), made independent of
// scheduling: all active pixels take `round` samples, then the noise test runs as its own pass and
// compacts the active list.  The first test happens at the first round boundary at or past
// min_adaptive_samples.  One slot per pixel; the variance estimate covers all samples so far.
void ensure_adaptive(State& s) {
  if (s.acc2) return;
  const size_t px = (size_t)s.width * (size_t)s.height;
  s.acc2 = dev_alloc<q4>(px * (size_t)s.max_chunks);
  s.adaptive.active[0] = dev_alloc<uint32_t>(px);
  s.adaptive.active[1] = dev_alloc<uint32_t>(px);
  s.adaptive.count = dev_alloc<uint32_t>(1);
  s.adaptive.noise_map = dev_alloc<uint32_t>(px);
  s.adaptive.own_noise = dev_alloc<float>(px);
  s.adaptive.total = dev_alloc<int>(px);
}

// Statistics shared between the ranks of a sample-sharded adaptive render: at every round boundary each
// rank packs (sum, samples, sum of squares) of all pixels into `stats`, `reduce` adds the ranks' buffers
// together in place, and the noise test runs on the totals, so every rank keeps the same pixels active.
struct SharedStats {
  float* stats;     // device, 8 floats per pixel
  int total_spp;    // samples per pixel of all ranks together
  int (*reduce)(void* user);
  void* user;
};

// sample_begin > 0: this call's samples are the global samples [sample_begin, sample_begin + spp) of every
// pixel (one rank's share).  Without `shared` the noise test runs on this share's own statistics.
int render_adaptive(State& s, int sample_begin, int spp, int max_depth, int frame, bool strict, float threshold,
                    int min_samples, const SharedStats* shared = nullptr) {
  ensure_adaptive(s);
  const uint32_t npx = (uint32_t)((size_t)s.width * (size_t)s.height);
  if (shared && !s.shared_acc) {
    s.shared_acc = dev_alloc<q4>(npx);
    s.shared_acc2 = dev_alloc<q4>(npx);
    s.shared_n = dev_alloc<int>(npx);
  }
  // the round schedule counts samples of all ranks; a rank's target is its proportional part of it
  const int total = shared ? std::max(shared->total_spp, spp) : spp;
  auto local_target = [&](int t) {
    return shared ? (int)std::min<long long>(spp, ((long long)t * spp + total - 1) / total) : t;
  };
  // strict mode replays the reference's 4-sample launches, so it is tested at that cadence
  const bool doubling = !strict && g_adaptive_round < 0;
  const int round = strict ? 4 : std::max(std::abs(g_adaptive_round), 1);
  auto next_target = [&](int d) { return doubling ? (d == 0 ? round : 2 * d) : d + round; };
  // A pixel's samples are rendered as several chunks side by side (one slot each) where that keeps enough
  // paths in flight; every chunk takes the same number of samples per round, so the chunk count has to
  // divide this rank's share of a round (and of the first target, which may be several rounds away).
  int first = next_target(0);
  while (first < min_samples && first < total) first = next_target(first);
  int chunks = 1;
  if (!strict) {
    // scenes that run staged keep more slots in flight (chunks_for_render); a function of scene and resolution,
    // not of what earlier renders left allocated, so a render gives the same result whatever came before it
    const bool staged_scene = !s.borrowed && staged_by_default(s);
    const int room = staged_scene ? (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)144 << 20) / std::max<uint32_t>(npx, 1u)))
                                  : (s.borrowed ? s.base_chunks : adaptive_room(npx));
    const int least = staged_scene ? 1 : 4;  // samples a chunk takes per round, at least
    const int cap = g_adaptive_chunks > 0 ? std::min(g_adaptive_chunks, room) : room;
    const int step = local_target(std::min(round, total)), start = local_target(std::min(first, total));
    while (2 * chunks <= cap && step % (2 * chunks) == 0 && start % (2 * chunks) == 0 && step / (2 * chunks) >= least) chunks *= 2;
    while (chunks > 1 && fit_chunks(s, chunks) < chunks) chunks /= 2;  // less memory than that: fewer chunks
  }
  RenderParams p = make_params(s, sample_begin, spp, max_depth, frame, strict, false);  // after the arrays may have grown
  p.chunks = chunks;
  p.chunk_spp = (spp + chunks - 1) / chunks;
  p.chunk_long = chunks;  // chunks of one length: the rounds divide by them
  p.chunk_short = p.chunk_spp;
  // Scenes that run fused: the last chunk takes a smaller part of every long round (about a third of what
  // the others take).  A round is one launch over the active pixels' slots, chunk after chunk, and it is over when
  // the last lane has finished the slot it holds: with the last chunk's slots short, the GPU runs empty for a
  // third as long at the end of every round.  The parts are fixed before the first round (the schedule of rounds
  // does not depend on what the noise test finds): per round and chunk how many samples, and so every chunk's
  // range of the pixel's samples.
  std::vector<int> part_a, part_b;  // cumulative samples after each round: the first chunks - 1 chunks, the last one
  if (!strict && chunks >= 4 && !(!s.borrowed && staged_by_default(s))) {
    static const int uneven_from = getenv("PT_ADAPTIVE_UNEVEN") ? atoi(getenv("PT_ADAPTIVE_UNEVEN")) : 16;
    int d = 0, d_mine = 0, a_sum = 0, b_sum = 0;
    bool ok = uneven_from > 0;
    while (ok && d < total) {  // the rounds of the loop below, this rank's share of each
      int t = next_target(d);
      while (t < min_samples && t < total) t = next_target(t);
      t = std::min(t, total);
      const int mine = local_target(t), quota = mine - d_mine;
      if (quota < 0 || quota % chunks != 0) { ok = false; break; }
      int a = quota / chunks, b = a;
      if (a >= uneven_from) {
        a = (int)((4LL * quota) / (4LL * chunks - 3));  // chunks - 1 parts of a and one of about a / 4 (2, 8, 16: slower) ...
        b = quota - (chunks - 1) * a;                   // ... which takes what the division leaves
      }
      a_sum += a; b_sum += b;
      part_a.push_back(a_sum); part_b.push_back(b_sum);
      d = t;
      d_mine = mine;
    }
    if (ok && !part_a.empty() && (chunks - 1) * a_sum + b_sum == spp && b_sum > 0 && a_sum != b_sum) {
      p.chunk_spp = a_sum;
      p.chunk_long = chunks - 1;
      p.chunk_short = b_sum;
    } else {
      part_a.clear(); part_b.clear();
    }
  }
  p.acc2 = s.acc2;
  RenderLoop loop(s, p);
  s.last_chunks = chunks;
  auto exchange = [&]() -> int {
    launch_pack_stats(s.acc, s.acc2, s.started, chunks, shared->stats, npx, s.stream);
    HIP_CHECK(hipStreamSynchronize(s.stream));
    const int rc = shared->reduce ? shared->reduce(shared->user) : 0;
    HIP_CHECK(hipSetDevice(s.device));
    return rc;
  };
  loop.begin();
  loop.reset_slots();
  HIP_CHECK(hipMemsetAsync(s.adaptive.noise_map, 0, (size_t)npx * sizeof(uint32_t), s.stream));
  launch_iota(s.adaptive.active[0], npx, s.stream);
  uint32_t n_active = npx;
  int which = 0, done = 0, done_local = 0, rc = 0;
  s.adaptive_rounds = 0;
  static const bool trace_rounds = getenv("PT_ADAPTIVE_TRACE") != nullptr;  // per-round wall times on stderr
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_round = 0.0;
  if (trace_rounds) { HIP_CHECK(hipStreamSynchronize(s.stream)); t_round = now_ms(); }
  while (n_active > 0 && done < total) {
    int target = next_target(done);
    while (target < min_samples && target < total) target = next_target(target);  // no test before min_samples
    target = std::min(target, total);
    const int mine = local_target(target);
    if (mine > done_local) {
      loop.p.round_limit = (mine + chunks - 1) / chunks;  // per slot; the last chunk of a pixel may hold fewer
      loop.p.round_limit_b = loop.p.round_limit;
      loop.p.round_split = chunks;
      if (!part_a.empty()) {  // the parts fixed above
        const size_t r = std::min((size_t)s.adaptive_rounds, part_a.size() - 1);
        loop.p.round_limit = part_a[r];
        loop.p.round_limit_b = part_b[r];
        loop.p.round_split = chunks - 1;
      }
      launch_expand_active(s.adaptive.active[which], n_active, chunks, npx, s.queue[0], s.stream);
      loop.drain(n_active * (uint32_t)chunks);
      done_local = mine;
    }
    done = target;
    s.adaptive_rounds++;
    if (done >= total) break;
    const q4 *t_acc = s.acc, *t_acc2 = s.acc2;
    const int* t_n = s.started;
    int t_chunks = chunks;
    if (shared) {
      if ((rc = exchange()) != 0) break;
      launch_unpack_stats(shared->stats, s.shared_acc, s.shared_acc2, s.shared_n, npx, s.stream);
      t_acc = s.shared_acc; t_acc2 = s.shared_acc2; t_n = s.shared_n;
      t_chunks = 1;
    }
    HIP_CHECK(hipMemsetAsync(s.adaptive.count, 0, sizeof(uint32_t), s.stream));
    launch_noise_round(s.adaptive, which, n_active, t_acc, t_acc2, t_n, t_chunks, s.width, s.height, s.bayer_pattern,
                       min_samples, total, threshold, s.stream);
    HIP_CHECK(hipMemcpyAsync(s.h_counters + 2 * CNT_WORDS, s.adaptive.count, sizeof(uint32_t), hipMemcpyDeviceToHost,
                             s.stream));
    HIP_CHECK(hipStreamSynchronize(s.stream));
    if (trace_rounds) {
      const double t = now_ms();
      fprintf(stderr, "[pt_hip] adaptive round to %d samples: %u pixels x %d chunks, %.2f ms, %u stay active\n", done, n_active,
              chunks, t - t_round, s.h_counters[2 * CNT_WORDS]);
      t_round = t;
    }
    n_active = s.h_counters[2 * CNT_WORDS];
    which ^= 1;
  }
  if (shared && rc == 0) rc = exchange();  // the totals of the finished render
  loop.end();
  return rc;
}

int shard_setting() {
  if (g_gpus < 0) {
    const char* v = getenv("PT_GPUS");
    g_gpus = v ? std::max(atoi(v), 1) : 1;
  }
  return g_gpus;
}
// Shard d > 0 of a render at the resolution of shard 0: its own State on GPU (first device + d) modulo the number
// of GPUs -- with fewer GPUs than shards several shards share one, which changes the speed and nothing else.
State& ensure_shard(int d, const State& first) {
  int ndev = 1;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  if ((int)g_shards.size() < d) g_shards.resize((size_t)d, nullptr);
  State*& sh = g_shards[(size_t)d - 1];
  const int device = (first.device + d) % std::max(ndev, 1);
  if (sh && (sh->width != first.width || sh->height != first.height || sh->device != device)) {
    delete sh;
    sh = nullptr;
  }
  if (!sh) {
    sh = make_state(first.width, first.height, device);
    sh->quiet = true;
    int can = 0;  // direct copies over xGMI where the GPUs allow it; the copy works either way
    if (device != first.device && hipDeviceCanAccessPeer(&can, first.device, device) == hipSuccess && can) {
      HIP_CHECK(hipSetDevice(first.device));
      (void)hipDeviceEnablePeerAccess(device, 0);
      (void)hipGetLastError();
    }
  }
  HIP_CHECK(hipSetDevice(sh->device));
  if (sh->scene_version != g_scene_version) upload_scene(*sh);
  if (sh->guide_version != g_snap_version) upload_guide(*sh);
  sh->cam = first.cam;
  sh->bayer_pattern = first.bayer_pattern;
  return *sh;
}

// A meeting point for the shards' host threads (C++17 has no std::barrier).
struct ThreadBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0;
  uint64_t generation = 0;
  explicit ThreadBarrier(int count) : n(count) {}
  void wait() {
    std::unique_lock<std::mutex> lock(m);
    const uint64_t g = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != g; });
    }
  }
};

// One render call split by sample index over n shards, one host thread and one GPU each (optix_wrapper.cpp:1102-1115
// stays the only call the host makes): shard d renders samples [d*spp/n, (d+1)*spp/n) of every pixel exactly as
// rank d of a multi-process render would, the shards' per-pixel totals are copied to shard 0's GPU and added there in
// shard order, and shard 0 divides by the number of samples taken.
// Adaptive renders run ONE noise test on the statistics of all shards together (the progressive reduce of
// pt_render_adaptive_shared_device, with peer copies in place of a collective): at every round boundary the shards
// meet, shard 0 adds the peers' packed statistics to its own in shard order and hands the totals back, and every
// shard keeps the same pixels active -- so min_adaptive_samples and the threshold mean what they mean on one GPU,
// whatever the number of shards.
void render_mean_sharded(State& s, int n, int spp, int max_depth, float noise_threshold, int min_adaptive_samples) {
  const uint32_t npx = (uint32_t)((size_t)s.width * (size_t)s.height);
  const bool adaptive = noise_threshold > 0.0f;
  (void)current_bvh();  // built once, before the threads need it
  std::vector<State*> st((size_t)n);
  st[0] = &s;
  for (int d = 1; d < n; ++d) st[(size_t)d] = &ensure_shard(d, s);
  // Everything that allocates happens here, one shard after the other: shards that share a GPU would otherwise each
  // size their path state for a quarter of the same free memory at the same moment.
  std::vector<int> on_device((size_t)n, 0);
  for (int d = 0; d < n; ++d)
    for (int e = 0; e < n; ++e) on_device[(size_t)d] += st[(size_t)e]->device == st[(size_t)d]->device ? 1 : 0;
  for (int d = 0; d < n; ++d) {
    State* sd = st[(size_t)d];
    HIP_CHECK(hipSetDevice(sd->device));
    if (!sd->d_sum4) sd->d_sum4 = dev_alloc<q4>(npx);
    if (adaptive) {
      ensure_adaptive(*sd);
      if (!sd->d_shard_stats) sd->d_shard_stats = dev_alloc<float>((size_t)npx * 8);
      HIP_CHECK(hipMemsetAsync(sd->d_shard_stats, 0, (size_t)npx * 8 * sizeof(float), sd->stream));
    } else {
      const int begin = (int)((long long)d * spp / n), count = (int)((long long)(d + 1) * spp / n) - begin;
      const int want = planned_layout(npx, count, staged_by_default(*sd)).chunks;
      fit_chunks(*sd, std::max(1, want / on_device[(size_t)d]));  // its part of what the GPU has room for
    }
  }
  HIP_CHECK(hipSetDevice(s.device));
  if (adaptive && !s.d_stats_in) s.d_stats_in = dev_alloc<float>((size_t)npx * 8);
  if (!adaptive && s.peer_capacity < (size_t)(n - 1) * npx) {
    dev_free(s.d_peer);
    s.peer_capacity = (size_t)(n - 1) * npx;
    s.d_peer = dev_alloc<q4>(s.peer_capacity);
  }
  const int frame = g_frame;
  ThreadBarrier meet(n);
  struct ReduceCtx {
    std::vector<State*>* st;
    ThreadBarrier* meet;
    int d, n;
    uint32_t npx;
  };
  // called by every shard's thread once its statistics are packed and its stream is idle
  auto reduce = [](void* user) -> int {
    ReduceCtx& c = *static_cast<ReduceCtx*>(user);
    c.meet->wait();
    if (c.d == 0) {
      State& first = *(*c.st)[0];
      const size_t bytes = (size_t)c.npx * 8 * sizeof(float);
      HIP_CHECK(hipSetDevice(first.device));
      for (int e = 1; e < c.n; ++e) {  // shard order: the totals do not depend on which shard arrived first
        State& peer = *(*c.st)[(size_t)e];
        HIP_CHECK(hipMemcpyPeerAsync(first.d_stats_in, first.device, peer.d_shard_stats, peer.device, bytes, first.stream));
        launch_add_stats(first.d_shard_stats, first.d_stats_in, (size_t)c.npx * 8, first.stream);
      }
      for (int e = 1; e < c.n; ++e) {
        State& peer = *(*c.st)[(size_t)e];
        HIP_CHECK(hipMemcpyPeerAsync(peer.d_shard_stats, peer.device, first.d_shard_stats, first.device, bytes, first.stream));
      }
      HIP_CHECK(hipStreamSynchronize(first.stream));
    }
    c.meet->wait();
    return 0;
  };
  std::vector<ReduceCtx> ctx((size_t)n);
  auto work = [&](int d) {
    State& sd = *st[(size_t)d];
    HIP_CHECK(hipSetDevice(sd.device));
    const int begin = (int)((long long)d * spp / n), count = (int)((long long)(d + 1) * spp / n) - begin;
    if (adaptive) {
      ctx[(size_t)d] = ReduceCtx{&st, &meet, d, n, npx};
      const SharedStats shared{sd.d_shard_stats, spp, reduce, &ctx[(size_t)d]};
      render_adaptive(sd, begin, count, max_depth, frame, false, noise_threshold, min_adaptive_samples, &shared);
    } else {
      render_core(sd, begin, count, max_depth, frame, false);
      launch_shard_sums(sd.acc, sd.started, sd.d_sum4, npx, sd.last_chunks, sd.stream);
    }
    HIP_CHECK(hipStreamSynchronize(sd.stream));
  };
  std::vector<std::thread> threads;
  for (int d = 1; d < n; ++d) threads.emplace_back(work, d);
  work(0);
  for (std::thread& t : threads) t.join();
  HIP_CHECK(hipSetDevice(s.device));
  s.counts_from_stats = adaptive;
  if (adaptive) {  // every shard holds the totals of the finished render
    launch_stats_mean(s.d_shard_stats, s.d_out_rgb, npx, s.stream);
    return;
  }
  for (int d = 1; d < n; ++d)
    HIP_CHECK(hipMemcpyPeerAsync(s.d_peer + (size_t)(d - 1) * npx, s.device, st[(size_t)d]->d_sum4, st[(size_t)d]->device,
                                 (size_t)npx * sizeof(q4), s.stream));
  launch_reduce_shards(s.d_sum4, s.d_peer, n - 1, s.d_out_rgb, npx, s.stream);
}

int render_mean(State& s, int spp, int max_depth, float noise_threshold, int min_adaptive_samples) {
  g_frame++;  // optix_wrapper.cpp:983
  const bool strict = mode() == PT_MODE_STRICT;
  // strict mode replays the reference's launches one pixel stream at a time: never sharded
  const int shards = strict ? 1 : std::min(shard_setting(), std::max(spp, 1));
  s.counts_from_stats = false;
  if (shards > 1) {
    render_mean_sharded(s, shards, spp, max_depth, noise_threshold, min_adaptive_samples);
    return 0;
  }
  if (noise_threshold > 0.0f && spp > 0)
    render_adaptive(s, 0, spp, max_depth, g_frame, strict, noise_threshold, min_adaptive_samples);
  else
    render_core(s, 0, spp, max_depth, g_frame, strict);
  launch_finalize(s.acc, s.started, s.d_out_rgb, (uint32_t)((size_t)s.width * (size_t)s.height), s.last_chunks, 0, s.stream);
  return 0;
}

template <typename T>
struct DevBuf {
  T* d = nullptr;
  size_t n = 0;
  explicit DevBuf(size_t count) : n(count) { d = dev_alloc<T>(count); }
  DevBuf(const T* h, size_t count) : n(count) {
    d = dev_alloc<T>(count);
    if (count) HIP_CHECK(hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice));
  }
  ~DevBuf() { dev_free(d); }
  void download(T* h) const {
    if (n) HIP_CHECK(hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost));
  }
};
State& dbg_state() { return ensure_handle(g_last_w > 0 ? g_last_w : 16, g_last_h > 0 ? g_last_h : 16); }

// ---- allocation-free rendering (pt_render_accumulate_device_noalloc)
// All device memory the render needs besides the caller's output -- path state, slot queues, counters and
// the scene -- lives in one array of the library's code object, so the call never touches the allocator.
// Each GPU the library runs on holds its own copy of the array from the moment the code object is loaded
// there until the process ends.
constexpr size_t kNaMaxSlots = PT_NOALLOC_MAX_SLOTS;
constexpr size_t kNaMaxSceneBytes = PT_NOALLOC_MAX_SCENE_BYTES;
constexpr size_t kNaSmall = 256;  // counters, ray counters: one aligned line each
// per slot: origin, direction, sample sum, rng/flags, finished-sample sum (16 B each), samples started, two queues
constexpr size_t kNaScratchBytes = kNaMaxSlots * (5 * 16 + 4 + 2 * 4) + kNaMaxSceneBytes + 2 * kNaSmall;
static __device__ __align__(256) unsigned char scratch[kNaScratchBytes];

struct NoAllocDevice {
  bool ready = false;
  unsigned char* base = nullptr;  // device address of `scratch` on this GPU
  unsigned char* scene_mem = nullptr;
  State state;
  uint32_t h_counters[3 * CNT_WORDS] = {0};
};
constexpr int kNaMaxDevices = 64;
NoAllocDevice g_noalloc[kNaMaxDevices];

// First call on a device (that device current): find this GPU's copy of the array and carve it up.
NoAllocDevice& noalloc_device(int device) {
  NoAllocDevice& na = g_noalloc[device];
  if (na.ready) return na;
  void* ptr = nullptr;
  HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  na.base = static_cast<unsigned char*>(ptr);
  unsigned char* cur = na.base;
  auto take = [&](size_t bytes) {
    unsigned char* r = cur;
    cur += (bytes + 255u) & ~(size_t)255u;
    return r;
  };
  State& s = na.state;
  s.borrowed = true;
  s.device = device;
  s.stream = nullptr;  // the null stream: creating one would allocate
  s.st_o = reinterpret_cast<q4*>(take(kNaMaxSlots * 16));
  s.st_d = reinterpret_cast<q4*>(take(kNaMaxSlots * 16));
  s.st_s = reinterpret_cast<q4*>(take(kNaMaxSlots * 16));
  s.st_m = reinterpret_cast<u4*>(take(kNaMaxSlots * 16));
  s.acc = reinterpret_cast<q4*>(take(kNaMaxSlots * 16));
  s.started = reinterpret_cast<int*>(take(kNaMaxSlots * 4));
  s.queue[0] = reinterpret_cast<uint32_t*>(take(kNaMaxSlots * 4));
  s.queue[1] = reinterpret_cast<uint32_t*>(take(kNaMaxSlots * 4));
  s.counters = reinterpret_cast<uint32_t*>(take(kNaSmall));
  s.ray_counters = reinterpret_cast<unsigned long long*>(take(kNaSmall));
  na.scene_mem = take(kNaMaxSceneBytes);
  if ((size_t)(cur - na.base) > kNaScratchBytes) {
    fprintf(stderr, "[pt_hip] internal error: scratch layout exceeds the array\n");
    std::abort();
  }
  s.h_counters = na.h_counters;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  s.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_CHECK(hipMemsetAsync(s.ray_counters, 0, 2 * sizeof(unsigned long long), s.stream));
  na.ready = true;
  return na;
}

}  // namespace

State::~State() {
  if (borrowed) return;  // nothing here is this State's to release
  if (stream) (void)hipStreamSynchronize(stream);
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  for (TimedLaunch& t : timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
  dev_free(d_nodes); dev_free(d_tris); dev_free(d_shade); dev_free(d_wide); dev_free(d_home_wide);
  dev_free(st_o); dev_free(st_d); dev_free(st_s); dev_free(st_m); dev_free(acc); dev_free(hits);
  dev_free(sh_o); dev_free(sh_d); dev_free(sh_c); dev_free(sh_count); dev_free(started); dev_free(hit_inst); dev_free(d_inst);
  dev_free(queue[0]); dev_free(queue[1]); dev_free(queue_tmp); dev_free(tiles); dev_free(counters); dev_free(ray_counters);
  dev_free(acc2); dev_free(adaptive.active[0]); dev_free(adaptive.active[1]); dev_free(adaptive.count);
  dev_free(adaptive.noise_map); dev_free(adaptive.own_noise); dev_free(adaptive.total);
  dev_free(shared_acc); dev_free(shared_acc2); dev_free(shared_n);
  dev_free(d_out_rgb); dev_free(d_out_bayer); dev_free(d_sum4); dev_free(d_peer); dev_free(d_shard_stats); dev_free(d_stats_in);
  dev_free(d_train); dev_free(d_train_idx); dev_free(d_regions); dev_free(d_lobes); dev_free(d_grid); dev_free(d_cells);
  if (h_counters) (void)hipHostFree(h_counters);
  for (int i = 0; i < 2; ++i) if (ev_slot[i]) (void)hipEventDestroy(ev_slot[i]);
  if (ev_begin) (void)hipEventDestroy(ev_begin);
  if (ev_end) (void)hipEventDestroy(ev_end);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (stream) (void)hipStreamDestroy(stream);
}

}  // namespace pt

using namespace pt;

// ======================================================================== Part 1
extern "C" {

int optix_render_rgb(int w, int h, int spp, int max_depth, float noise_threshold, int min_adaptive_samples, float* out_rgb) {
  State& s = ensure_handle(w, h);
  if (!out_rgb) return -1;  // optix_wrapper.cpp:980
  render_mean(s, spp, max_depth, noise_threshold, min_adaptive_samples);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  HIP_CHECK(hipMemcpy(out_rgb, s.d_out_rgb, (size_t)w * (size_t)h * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int optix_render_bayer_f32(int w, int h, int spp, int max_depth, float noise_threshold, int min_adaptive_samples,
                           int pattern, float* out_raw) {
  State& s = ensure_handle(w, h);
  optix_ctx_set_bayer(&s, pattern);  // optix_wrapper.cpp:1113
  if (!out_raw) return -1;
  render_mean(s, spp, max_depth, noise_threshold, min_adaptive_samples);
  launch_mosaic(s.d_out_rgb, s.d_out_bayer, w, h, s.bayer_pattern, 1.0f, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  HIP_CHECK(hipMemcpy(out_raw, s.d_out_bayer, (size_t)w * (size_t)h * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

void* optix_alloc_host(size_t bytes) {
  HIP_CHECK(hipSetDevice(device_index()));
  void* p = nullptr;
  HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
  return p;
}
void optix_free_host(void* ptr) {
  if (ptr) HIP_CHECK(hipHostFree(ptr));
}
void optix_synchronize(void) {
  if (g_handle) {
    HIP_CHECK(hipStreamSynchronize(g_handle->stream));
    HIP_CHECK(hipStreamSynchronize(g_handle->copy_stream));
  }
  for (State* sh : g_shards)
    if (sh) HIP_CHECK(hipStreamSynchronize(sh->stream));
}

void* optix_ctx_create(int width, int height) { return make_state(width, height); }
void optix_ctx_destroy(void* handle) {
  State* s = static_cast<State*>(handle);
  if (s == g_handle) {
    g_handle = nullptr;
    drop_shards();
  }
  delete s;
}
void optix_ctx_set_bayer(void* handle, int pattern) { static_cast<State*>(handle)->bayer_pattern = pattern; }
void optix_ctx_set_camera(void* handle, float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy,
                          float uz, float vfov_deg) {
  const float eye[3] = {ex, ey, ez}, look[3] = {lx, ly, lz}, up[3] = {ux, uy, uz};
  set_camera(*static_cast<State*>(handle), eye, look, up, vfov_deg);
}

// ---- guiding (optix_wrapper.cpp:1168-1299)
void guiding_upload_snapshot(const struct pgl_region* regions, uint32_t n_regions, const struct pgl_lobe* lobes,
                             uint32_t n_lobes) {
  g_snap_regions.assign(regions, regions + n_regions);
  g_snap_lobes.assign(lobes, lobes + n_lobes);
  g_snap_version++;
  if (g_handle) {
    HIP_CHECK(hipSetDevice(g_handle->device));
    upload_guide(*g_handle);
  }
}
void guiding_set_enabled(int enabled) { g_guiding_enabled = enabled; }
void guiding_set_train_buffer_size(uint32_t n) { g_train_capacity = n; }
void guiding_set_grid_res(int x, int y, int z) {
  g_grid_res[0] = x; g_grid_res[1] = y; g_grid_res[2] = z;
}
void guiding_map_train_samples(const TrainSample** samples, uint32_t* count) {
  static std::vector<TrainSample> host;
  if (samples) *samples = nullptr;
  if (count) *count = 0;
  if (!g_handle) return;
  static_assert(sizeof(TrainSample) == sizeof(TrainRec), "layout");
  // one ring per sample shard (pt_set_device_count): the records of all of them, shard 0 first
  std::vector<State*> states{g_handle};
  for (State* sh : g_shards)
    if (sh) states.push_back(sh);
  host.clear();
  std::vector<TrainSample> ring;
  for (State* stp : states) {
    State& st = *stp;
    if (!st.d_train || !st.d_train_idx || st.train_capacity == 0) continue;
    HIP_CHECK(hipSetDevice(st.device));
    ring.resize(st.train_capacity);
    HIP_CHECK(hipMemcpy(ring.data(), st.d_train, sizeof(TrainSample) * (size_t)st.train_capacity, hipMemcpyDeviceToHost));
    uint32_t n = 0;
    HIP_CHECK(hipMemcpy(&n, st.d_train_idx, sizeof(uint32_t), hipMemcpyDeviceToHost));
    // vertices land in hashed slots (see capture_train): keep the slots that were written, in ring order;
    // never more than were captured or than the ring holds (the reference reports the raw counter)
    const size_t limit = host.size() + std::min(n, st.train_capacity);
    for (size_t i = 0; i < ring.size() && host.size() < limit; ++i) {
      if (ring[i].is_delta == kTrainEmpty) continue;
      TrainRec rec;
      memcpy(&rec, &ring[i], sizeof(rec));
      if ((rec.is_delta & ~1u) != train_digest(rec)) continue;  // two vertices were written here at once
      ring[i].is_delta &= 1u;
      host.push_back(ring[i]);
    }
  }
  HIP_CHECK(hipSetDevice(g_handle->device));
  if (samples) *samples = host.data();
  if (count) *count = (uint32_t)host.size();
}
// Training statistics taken on the device (include/pt_api.h).  d_bins lives on the current renderer's GPU.
static int bin_state_ring(State& st, const BinField& bf, const std::vector<float>& axes, long long* d_bins) {
  if (!st.d_train || st.train_capacity == 0) return -1;
  HIP_CHECK(hipSetDevice(st.device));
  float* d_axes = dev_alloc<float>(axes.size());
  HIP_CHECK(hipMemcpyAsync(d_axes, axes.data(), axes.size() * sizeof(float), hipMemcpyHostToDevice, st.stream));
  launch_bin_train(reinterpret_cast<const TrainRec*>(st.d_train), st.train_capacity, bf, d_axes, d_bins, st.stream);
  HIP_CHECK(hipStreamSynchronize(st.stream));
  dev_free(d_axes);
  return 0;
}
static bool bin_layout(pgl_field_t field, BinField& bf, std::vector<float>& axes, uint32_t& regions) {
  if (pt_pgl_field_layout(field, bf.bmin, bf.bmax, bf.res, nullptr, &regions) != 0 || regions == 0) return false;
  axes.resize((size_t)regions * kBinLobes * 3);
  return pt_pgl_field_layout(field, nullptr, nullptr, nullptr, axes.data(), nullptr) == 0;
}
int pt_guiding_bin_train_samples_device(pgl_field_t field, long long* d_bins) {
  static_assert(kBinLobes == PT_PGL_BIN_LOBES && kBinWords == PT_PGL_BIN_WORDS && kBinShift == PT_PGL_BIN_SHIFT, "bin table layout");
  if (!g_handle || !field || !d_bins) return -1;
  BinField bf;
  std::vector<float> axes;
  uint32_t regions = 0;
  if (!bin_layout(field, bf, axes, regions)) return -1;
  return bin_state_ring(*g_handle, bf, axes, d_bins);
}
int pt_pgl_field_update_from_ring(pgl_field_t field) {
  if (!g_handle || !field) return -1;
  BinField bf;
  std::vector<float> axes;
  uint32_t regions = 0;
  if (!bin_layout(field, bf, axes, regions)) return -1;
  const size_t words = (size_t)regions * kBinLobes * kBinWords;
  std::vector<long long> total(words, 0), part(words);
  std::vector<State*> states{g_handle};
  for (State* sh : g_shards)
    if (sh) states.push_back(sh);
  int binned = 0;
  for (State* st : states) {  // every sample shard has its own ring on its own GPU: integer tables add up exactly
    if (!st->d_train || st->train_capacity == 0) continue;
    HIP_CHECK(hipSetDevice(st->device));
    long long* d_bins = dev_alloc<long long>(words);
    HIP_CHECK(hipMemsetAsync(d_bins, 0, words * sizeof(long long), st->stream));
    if (bin_state_ring(*st, bf, axes, d_bins) == 0) {
      HIP_CHECK(hipMemcpy(part.data(), d_bins, words * sizeof(long long), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < words; ++i) total[i] += part[i];
      binned++;
    }
    dev_free(d_bins);
  }
  HIP_CHECK(hipSetDevice(g_handle->device));
  if (!binned) return -1;
  pt_pgl_field_update_binned(field, total.data());
  guiding_reset_train_write_idx();
  return 0;
}
void guiding_reset_train_write_idx(void) {
  if (!g_handle) return;
  std::vector<State*> states{g_handle};
  for (State* sh : g_shards)
    if (sh) states.push_back(sh);
  for (State* st : states) {
    if (!st->d_train_idx) continue;
    HIP_CHECK(hipSetDevice(st->device));
    HIP_CHECK(hipMemset(st->d_train_idx, 0, sizeof(uint32_t)));
    if (st->d_train) HIP_CHECK(hipMemset(st->d_train, 0xff, sizeof(TrainRec) * (size_t)st->train_capacity));
  }
  HIP_CHECK(hipSetDevice(g_handle->device));
}

// ======================================================================== Part 2
void pt_set_mode(int m) { g_mode = m == PT_MODE_STRICT ? PT_MODE_STRICT : PT_MODE_CONVERGE; }
int pt_get_mode(void) { return mode(); }
void pt_set_pipeline(int p) {
  g_pipeline = p == PT_PIPELINE_STAGED ? PT_PIPELINE_STAGED : (p == PT_PIPELINE_FUSED ? PT_PIPELINE_FUSED : -2);
}
int pt_get_pipeline(void) { return pipeline() < 0 ? PT_PIPELINE_AUTO : pipeline(); }
void pt_set_device(int d) {
  if (d != device_index()) {
    delete g_handle;
    g_handle = nullptr;
    drop_shards();
  }
  g_device = d;
}
void pt_set_device_count(int n) {
  g_gpus = n > 0 ? n : 1;
  if ((int)g_shards.size() > g_gpus - 1) {  // shards no longer used give their memory back
    for (size_t d = (size_t)g_gpus - 1; d < g_shards.size(); ++d) delete g_shards[d];
    g_shards.resize((size_t)g_gpus - 1);
  }
}
int pt_get_device_count(void) { return shard_setting(); }
void pt_shard_range(int spp, int n_shards, int shard, int* begin, int* count) {
  const int n = n_shards > 0 ? n_shards : 1, d = std::min(std::max(shard, 0), n - 1);
  const int b = (int)((long long)d * spp / n), e = (int)((long long)(d + 1) * spp / n);
  if (begin) *begin = b;
  if (count) *count = e - b;
}
void pt_set_frame(int frame) {
  g_frame = frame;
  g_frame_pinned = true;
}
int pt_get_frame(void) { return g_frame; }

int pt_render_accumulate_device(int w, int h, int sample_begin, int sample_count, int max_depth, int frame,
                                float* d_sum_rgb) {
  State& s = ensure_handle(w, h);
  if (!d_sum_rgb) return -1;
  render_core(s, sample_begin, sample_count, max_depth, frame, false);
  launch_finalize(s.acc, s.started, d_sum_rgb, (uint32_t)((size_t)w * (size_t)h), s.last_chunks, 1, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  return 0;
}

int pt_render_accumulate_device_noalloc(int w, int h, int sample_begin, int sample_count, int max_depth, int frame,
                                        float* d_sum_rgb) {
  if (!d_sum_rgb) return -1;
  const int device = device_index();
  if (w <= 0 || h <= 0 || sample_count < 0 || device < 0 || device >= kNaMaxDevices) {
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: bad argument (w %d, h %d, samples %d, device %d)\n", w, h,
            sample_count, device);
    return -2;
  }
  HIP_CHECK(hipSetDevice(device));
  NoAllocDevice& na = noalloc_device(device);
  State& s = na.state;
  const size_t px = (size_t)w * (size_t)h;
  if (px > kNaMaxSlots) {
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: %d x %d = %zu pixels exceed the scratch array's %zu path slots\n",
            w, h, px, kNaMaxSlots);
    return -2;
  }
  if (g_guiding_enabled && !g_snap_regions.empty()) {
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: guided sampling needs the field in device memory "
                    "(use pt_render_accumulate_device, or guiding_set_enabled(0))\n");
    return -2;
  }
  if (scene().instanced()) {
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: scenes made of instances need the staged kernels' hit "
                    "records (use pt_render_accumulate_device)\n");
    return -2;
  }
  // scene: small scenes only, copied into the array when the scene has changed
  const BvhBuildResult& bvh = current_bvh();
  const size_t nodes_b = (bvh.nodes.size() * sizeof(BvhNode) + 255u) & ~(size_t)255u;
  const size_t tris_b = (bvh.tris.size() * sizeof(q4) + 255u) & ~(size_t)255u;
  const size_t shade_b = ((size_t)std::max(scene().num_tris(), 1) * sizeof(ShadeRec) + 255u) & ~(size_t)255u;
  const size_t wide_b = (bvh.wide_nodes.size() * sizeof(u4) + 255u) & ~(size_t)255u;
  const size_t exact_b = wide_b + bvh.home_wide.size() * sizeof(uint32_t);
  if (nodes_b + tris_b + shade_b + exact_b > kNaMaxSceneBytes) {
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: the scene (%zu bytes of nodes, items and shading records) "
                    "exceeds the %zu bytes reserved for it\n", nodes_b + tris_b + shade_b + exact_b, kNaMaxSceneBytes);
    return -2;
  }
  if (s.scene_version != g_scene_version) {
    const std::vector<ShadeRec> recs = shade_records(bvh);
    BvhNode* dn = reinterpret_cast<BvhNode*>(na.scene_mem);
    q4* dt = reinterpret_cast<q4*>(na.scene_mem + nodes_b);
    ShadeRec* ds = reinterpret_cast<ShadeRec*>(na.scene_mem + nodes_b + tris_b);
    HIP_CHECK(hipMemcpy(dn, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dt, bvh.tris.data(), bvh.tris.size() * sizeof(q4), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(ds, recs.data(), recs.size() * sizeof(ShadeRec), hipMemcpyHostToDevice));
    u4* dw = reinterpret_cast<u4*>(na.scene_mem + nodes_b + tris_b + shade_b);
    uint32_t* dh = reinterpret_cast<uint32_t*>(na.scene_mem + nodes_b + tris_b + shade_b + wide_b);
    if (!bvh.wide_nodes.empty()) {
      HIP_CHECK(hipMemcpy(dw, bvh.wide_nodes.data(), bvh.wide_nodes.size() * sizeof(u4), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dh, bvh.home_wide.data(), bvh.home_wide.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    s.scene.wide = nullptr;
    s.scene.wide_src = bvh.wide_nodes.empty() ? nullptr : dw;
    s.scene.home_wide = bvh.wide_nodes.empty() ? nullptr : dh;
    s.scene.wide_groups = bvh.wide_groups;
  s.scene.wide_one_node = (!bvh.wide_nodes.empty() && bvh.wide_depth == 1) ? 1u : 0u;
    s.scene.nodes = dn;
    s.scene.tris = dt;
    s.scene.shade = ds;
    s.scene.inst_records = nullptr;
    describe_scene(s, bvh);
  }
  s.width = w;
  s.height = h;
  s.max_chunks = s.base_chunks = max_chunks_for(px);  // the same split into chunks as pt_render_accumulate_device ...
  s.max_chunks_split = (int)std::max<size_t>(1, std::min<size_t>((size_t)s.max_chunks + 3, kNaMaxSlots / std::max<size_t>(px, 1)));
  const float eye[3] = {0.f, 3.f, 1.f}, look[3] = {0.f, 0.f, 1.f}, up[3] = {0.f, 0.f, 1.f};
  set_camera(s, eye, look, up, 45.f);
  RenderParams p = make_params(s, sample_begin, sample_count, max_depth, frame, false);
  p.guiding = 0;
  p.train = nullptr;
  p.train_write_idx = nullptr;
  p.train_capacity = 0;
  if (px * (size_t)p.chunks > kNaMaxSlots) {  // ... so the two agree bit for bit; more than fit: refuse
    fprintf(stderr, "[pt_hip] pt_render_accumulate_device_noalloc: %d x %d pixels in %d chunks of samples = %zu path slots "
                    "exceed the scratch array's %zu\n", w, h, p.chunks, px * (size_t)p.chunks, kNaMaxSlots);
    return -2;
  }
  render_core(s, p);
  launch_finalize(s.acc, s.started, d_sum_rgb, (uint32_t)px, s.last_chunks, 1, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  return 0;
}

int pt_render_adaptive_accumulate_device(int w, int h, int sample_begin, int sample_count, int max_depth, int frame,
                                         float noise_threshold, int min_adaptive_samples, float* d_sum_rgb,
                                         float* d_count) {
  State& s = ensure_handle(w, h);
  if (!d_sum_rgb || !d_count) return -1;
  if (noise_threshold > 0.0f && sample_count > 0)
    render_adaptive(s, sample_begin, sample_count, max_depth, frame, false, noise_threshold, min_adaptive_samples);
  else
    render_core(s, sample_begin, sample_count, max_depth, frame, false);
  const uint32_t npx = (uint32_t)((size_t)w * (size_t)h);
  launch_finalize(s.acc, s.started, d_sum_rgb, npx, s.last_chunks, 1, s.stream);
  launch_add_counts(s.started, d_count, npx, s.last_chunks, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  return 0;
}

int pt_render_adaptive_shared_device(int w, int h, int sample_begin, int sample_count, int total_spp, int max_depth,
                                     int frame, float noise_threshold, int min_adaptive_samples, float* d_stats,
                                     pt_reduce_fn reduce, void* user) {
  State& s = ensure_handle(w, h);
  if (!d_stats || sample_count <= 0 || total_spp < sample_count || !(noise_threshold > 0.0f)) return -1;
  const SharedStats shared{d_stats, total_spp, reduce, user};
  return render_adaptive(s, sample_begin, sample_count, max_depth, frame, false, noise_threshold, min_adaptive_samples,
                         &shared);
}

int pt_mosaic_device(const float* d_rgb, int w, int h, int pattern, float divisor, float* d_raw) {
  State& s = ensure_handle(w, h);
  if (!d_rgb || !d_raw) return -1;
  launch_mosaic(d_rgb, d_raw, w, h, pattern, divisor, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  return 0;
}

int pt_demosaic_device(const float* d_raw, int w, int h, int pattern, int algo, float* d_out_rgb) {
  if (!d_raw || !d_out_rgb || w <= 0 || h <= 0) return -1;
  HIP_CHECK(hipSetDevice(device_index()));
  const size_t n = (size_t)w * (size_t)h;
  float* tmp = nullptr;
#ifdef PT_DEMOSAIC_PASSES
  static float* s_tmp = nullptr;  // comparison builds only: scratch of the pass-by-pass form, grown on demand
  static size_t s_tmp_n = 0;
  if (getenv("PT_DEMOSAIC_PASSES") && s_tmp_n < 3 * n) {
    dev_free(s_tmp);
    s_tmp = dev_alloc<float>(3 * n);
    s_tmp_n = 3 * n;
  }
  tmp = s_tmp;
#endif
  hipStream_t st = g_handle ? g_handle->stream : nullptr;
  static hipEvent_t a = nullptr, b = nullptr;  // created once: a call costs two event records and the kernel
  if (!a) {
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
  }
  HIP_CHECK(hipEventRecord(a, st));
  launch_demosaic(d_raw, d_out_rgb, tmp, w, h, pattern, algo, st);
  HIP_CHECK(hipEventRecord(b, st));
  HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  g_kstats[PT_K_DEMOSAIC].total_ms += ms;
  g_kstats[PT_K_DEMOSAIC].timed++;
  g_kstats[PT_K_DEMOSAIC].launches++;
  g_kstats[PT_K_DEMOSAIC].work += n;
  return 0;
}

int pt_demosaic_f32(const float* raw, int w, int h, int pattern, int algo, float* out_rgb) {
  if (!raw || !out_rgb || w <= 0 || h <= 0) return -1;
  HIP_CHECK(hipSetDevice(device_index()));
  const size_t n = (size_t)w * (size_t)h;
  float* d_raw = dev_alloc<float>(n);
  float* d_out = dev_alloc<float>(3 * n);
  HIP_CHECK(hipMemcpy(d_raw, raw, n * sizeof(float), hipMemcpyHostToDevice));
  pt_demosaic_device(d_raw, w, h, pattern, algo, d_out);
  HIP_CHECK(hipMemcpy(out_rgb, d_out, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
  dev_free(d_raw);
  dev_free(d_out);
  return 0;
}

void pt_get_ray_counts(uint64_t* radiance_rays, uint64_t* shadow_rays) {
  if (radiance_rays) *radiance_rays = g_rays_total[0];
  if (shadow_rays) *shadow_rays = g_rays_total[1];
}
void pt_reset_ray_counts(void) { g_rays_total[0] = g_rays_total[1] = 0; }
void pt_set_kernel_timing(int every_nth_launch) { g_timing_every = every_nth_launch; }
void pt_get_kernel_stats(int kernel, double* total_ms, uint64_t* timed_launches, uint64_t* launches, uint64_t* work_items) {
  if (kernel < 0 || kernel >= PT_K_COUNT) return;
  if (total_ms) *total_ms = g_kstats[kernel].total_ms;
  if (timed_launches) *timed_launches = g_kstats[kernel].timed;
  if (launches) *launches = g_kstats[kernel].launches;
  if (work_items) *work_items = g_kstats[kernel].work;
}
void pt_reset_kernel_stats(void) {
  for (KernelStat& k : g_kstats) k = KernelStat();
}
void pt_set_defer_shadow(int on) { g_defer_shadow = on < 0 ? -1 : (on ? 1 : 0); }
int pt_get_defer_shadow(void) { return g_defer_shadow; }
void pt_set_one_launch(int on) { g_one_launch = on ? 1 : 0; }
int pt_get_one_launch(void) { return g_one_launch; }
void pt_set_adaptive_round(int samples) { g_adaptive_round = samples != 0 ? samples : -32; }
int pt_get_adaptive_round(void) { return g_adaptive_round; }
void pt_set_adaptive_chunks(int chunks) { g_adaptive_chunks = chunks > 0 ? chunks : 0; }
int pt_get_adaptive_chunks(void) { return g_adaptive_chunks; }
int pt_get_sample_counts(int* out_counts, int* out_rounds) {
  if (!g_handle) return -1;
  State& s = *g_handle;
  HIP_CHECK(hipSetDevice(s.device));
  HIP_CHECK(hipStreamSynchronize(s.stream));
  const size_t px = (size_t)s.width * (size_t)s.height;
  if (out_counts && s.counts_from_stats && s.d_shard_stats) {  // sample shards with one noise test: the totals of all shards
    std::vector<float> tmp(px * 8);
    HIP_CHECK(hipMemcpy(tmp.data(), s.d_shard_stats, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < px; ++i) out_counts[i] = (int)(tmp[8 * i + 3] + 0.5f);
  } else if (out_counts) {
    std::vector<int> tmp(px * (size_t)s.last_chunks);
    HIP_CHECK(hipMemcpy(tmp.data(), s.started, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < px; ++i) {
      int c = 0;
      for (int k = 0; k < s.last_chunks; ++k) c += tmp[(size_t)k * px + i];
      out_counts[i] = c;
    }
  }
  if (out_rounds) *out_rounds = s.adaptive_rounds;
  return 0;
}
int pt_dbg_profile_counters(uint64_t* out64) {
  if (!g_handle) return 0;
  HIP_CHECK(hipSetDevice(g_handle->device));
  HIP_CHECK(hipStreamSynchronize(g_handle->stream));
  return read_profile_counters(reinterpret_cast<unsigned long long*>(out64));
}
int pt_dbg_fused_occupancy(int adaptive, unsigned long long* lds_bytes) {
  if (!g_handle) return -1;
  State& s = *g_handle;
  HIP_CHECK(hipSetDevice(s.device));
  RenderParams p = make_params(s, 0, 64, 32, 1, false, false);
  if (adaptive) p.acc2 = reinterpret_cast<q4*>(s.st_o);  // only tested for being set
  else p.fresh_queue = 1;                                 // a fixed-count render starts from untouched slots
  size_t lds = 0;
  const int n = bounce_occupancy(p, 256, &lds, g_one_launch != 0);
  if (lds_bytes) *lds_bytes = lds;
  return n;
}
double pt_last_render_ms(void) { return g_handle ? g_handle->last_render_ms : 0.0; }
int pt_last_render_chunks(void) { return g_handle ? g_handle->last_chunks : 0; }
int pt_dbg_chunk_plan(int w, int h, int sample_count, int staged_scene, int have_chunks, unsigned long long free_bytes) {
  if (w <= 0 || h <= 0 || sample_count < 0 || have_chunks < 1) return -1;
  const size_t px = (size_t)w * (size_t)h;
  const int n = planned_chunks(px, sample_count, staged_scene != 0);
  const int want = planned_layout(px, sample_count, staged_scene != 0).chunks;  // with the last chunk cut (plan_for_render)
  const int got = chunks_that_fit(px, want, have_chunks, (size_t)free_bytes);
  return got == want ? want : std::min(got, n);
}

void pt_scene_cornell(void) {
  g_scene = make_cornell();
  g_scene_init = true;
  g_scene_version++;
}
void pt_scene_synthetic(uint32_t seed, int n_objects, int subdiv) {
  g_scene = make_synthetic(seed, n_objects, subdiv);
  g_scene_init = true;
  g_scene_version++;
}
int pt_scene_triangles(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                       const float* materials11) {
  // the arrays are caller memory: check them before the BVH builder indexes with them
  const char* why = nullptr;
  if (n_vertices < 0 || n_triangles < 0) why = "negative count";
  else if ((n_vertices > 0 && !vertices) || (n_triangles > 0 && !indices)) why = "null array";
  for (size_t k = 0; !why && k < 3 * (size_t)n_triangles; ++k)
    if (indices[k] >= (uint32_t)n_vertices) why = "vertex index out of range";
  for (size_t k = 0; !why && k < 3 * (size_t)n_vertices; ++k)
    if (!std::isfinite(vertices[k])) why = "vertex coordinate not finite";
  for (size_t k = 0; !why && materials11 && k < 11 * (size_t)n_triangles; ++k)
    if (!std::isfinite(materials11[k])) why = "material value not finite";
  if (why) {
    fprintf(stderr, "[pt_hip] pt_scene_triangles: %s; the current scene is kept\n", why);
    return -1;
  }
  g_scene = make_from_triangles(vertices, n_vertices, indices, n_triangles, materials11);
  g_scene_init = true;
  g_scene_version++;
  return 0;
}
// ---- scenes made of instances (include/pt_api.h)
static HostScene g_pending;      // the instanced scene being described
static bool g_pending_open = false;
int pt_scene_instanced_begin(void) {
  g_pending = make_from_triangles(nullptr, 0, nullptr, 0, nullptr);  // empty geometry, the reference's light
  g_pending_open = true;
  return 0;
}
int pt_scene_add_mesh(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles, const float* materials11) {
  const char* why = nullptr;
  if (!g_pending_open) why = "no instanced scene has been begun";
  else if (n_vertices <= 0 || n_triangles <= 0) why = "empty mesh";
  else if (!vertices || !indices) why = "null array";
  for (size_t k = 0; !why && k < 3 * (size_t)n_triangles; ++k)
    if (indices[k] >= (uint32_t)n_vertices) why = "vertex index out of range";
  for (size_t k = 0; !why && k < 3 * (size_t)n_vertices; ++k)
    if (!std::isfinite(vertices[k])) why = "vertex coordinate not finite";
  for (size_t k = 0; !why && materials11 && k < 11 * (size_t)n_triangles; ++k)
    if (!std::isfinite(materials11[k])) why = "material value not finite";
  if (why) {
    fprintf(stderr, "[pt_hip] pt_scene_add_mesh: %s\n", why);
    return -1;
  }
  const HostScene one = make_from_triangles(vertices, n_vertices, indices, n_triangles, materials11);
  const uint32_t vbase = (uint32_t)(g_pending.vertices.size() / 3);
  HostScene::MeshRange mr{(uint32_t)g_pending.num_tris(), (uint32_t)n_triangles};
  g_pending.vertices.insert(g_pending.vertices.end(), one.vertices.begin(), one.vertices.end());
  for (uint32_t i : one.indices) g_pending.indices.push_back(vbase + i);
  g_pending.materials.insert(g_pending.materials.end(), one.materials.begin(), one.materials.end());
  g_pending.meshes.push_back(mr);
  return (int)g_pending.meshes.size() - 1;
}
int pt_scene_add_instance(int mesh, const float* xform12, const float* material11) {
  const char* why = nullptr;
  double inv[12];
  if (!g_pending_open) why = "no instanced scene has been begun";
  else if (mesh < 0 || (size_t)mesh >= g_pending.meshes.size()) why = "no such mesh";
  else if (!xform12) why = "null transform";
  for (int k = 0; !why && k < 12; ++k)
    if (!std::isfinite(xform12[k])) why = "transform not finite";
  for (int k = 0; !why && material11 && k < 11; ++k)
    if (!std::isfinite(material11[k])) why = "material value not finite";
  if (!why && !invert_affine(xform12, inv)) why = "transform cannot be inverted";
  if (why) {
    fprintf(stderr, "[pt_hip] pt_scene_add_instance: %s\n", why);
    return -1;
  }
  HostScene::Instance in{};
  in.mesh = (uint32_t)mesh;
  memcpy(in.xform, xform12, sizeof(in.xform));
  if (material11) {
    in.has_material = 1;
    memcpy(&in.material, material11, sizeof(Material));
  }
  g_pending.instances.push_back(in);
  return (int)g_pending.instances.size() - 1;
}
int pt_scene_instanced_commit(void) {
  if (!g_pending_open || g_pending.instances.empty()) {
    fprintf(stderr, "[pt_hip] pt_scene_instanced_commit: no instances; the current scene is kept\n");
    return -1;
  }
  g_pending.compute_normals();
  g_scene = std::move(g_pending);
  g_pending = HostScene();
  g_pending_open = false;
  g_scene_init = true;
  g_scene_version++;
  return (int)g_scene.instances.size();
}
void pt_scene_synthetic_instanced(uint32_t seed, int n_objects, int subdiv) {
  g_scene = make_synthetic_instanced(seed, n_objects, subdiv);
  g_scene_init = true;
  g_scene_version++;
}
// The current scene's instances written out as world-space triangles become the scene (a flat scene stays as it is).
int pt_scene_flatten(void) {
  HostScene& hs = scene();
  if (hs.instanced()) {
    g_scene = hs.flattened();
    g_scene_version++;
  }
  return g_scene.num_tris();
}
int pt_scene_instance_count(void) { return (int)scene().instances.size(); }

void pt_scene_counts(int* n_vertices, int* n_triangles) {
  HostScene& hs = scene();
  if (n_vertices) *n_vertices = (int)(hs.vertices.size() / 3);
  if (n_triangles) *n_triangles = hs.num_tris();
}
int pt_scene_load_obj(const char* path) {
  HostScene loaded;
  std::string err;
  if (!path || !load_obj(path, loaded, err)) {
    fprintf(stderr, "[pt_hip] pt_scene_load_obj: %s\n", path ? err.c_str() : "null path");
    return -1;
  }
  g_scene = std::move(loaded);
  g_scene_init = true;
  g_scene_version++;
  return g_scene.num_tris();
}
int pt_scene_save_obj(const char* path) {
  std::string err;
  // a scene made of instances is written as the world-space triangles it stands for
  if (!path || !save_obj(path, scene().instanced() ? scene().flattened() : scene(), err)) {
    fprintf(stderr, "[pt_hip] pt_scene_save_obj: %s\n", path ? err.c_str() : "null path");
    return -1;
  }
  return 0;
}
void pt_scene_set_light(const float pos[3], const float emit[3], const float normal[3], const float half[2]) {
  HostScene& hs = scene();
  for (int k = 0; k < 3; ++k) {
    if (pos) hs.light_pos[k] = pos[k];
    if (emit) hs.light_emit[k] = emit[k];
    if (normal) hs.light_normal[k] = normal[k];
  }
  if (half) { hs.light_half[0] = half[0]; hs.light_half[1] = half[1]; }
  g_scene_version++;
}
void pt_set_bvh_pairing(int on) {
  bvh_set_pairing(on);
  g_scene_version++;  // the next render rebuilds the tree
}
void pt_scene_info(int* n_triangles, int* n_bvh_nodes, int* bvh_depth, size_t* bvh_bytes) {
  State& s = ensure_handle(g_last_w > 0 ? g_last_w : 16, g_last_h > 0 ? g_last_h : 16);
  if (n_triangles) *n_triangles = (int)s.scene.num_tris;
  if (n_bvh_nodes) *n_bvh_nodes = s.num_nodes;
  if (bvh_depth) *bvh_depth = s.bvh_depth;
  if (bvh_bytes) *bvh_bytes = s.bvh_bytes;
}
void pt_scene_get(float* vertices, uint32_t* indices, float* normals, float* materials11) {
  HostScene& hs = scene();
  if (vertices) memcpy(vertices, hs.vertices.data(), hs.vertices.size() * sizeof(float));
  if (indices) memcpy(indices, hs.indices.data(), hs.indices.size() * sizeof(uint32_t));
  if (normals) memcpy(normals, hs.normals.data(), hs.normals.size() * sizeof(float));
  if (materials11) memcpy(materials11, hs.materials.data(), hs.materials.size() * sizeof(Material));
}

// ---- stage-level entry points

int pt_dbg_pcg(uint64_t seed, int n, uint32_t* out_u32, float* out_f32) {
  State& s = dbg_state();
  DevBuf<uint32_t> u((size_t)n);
  DevBuf<float> f((size_t)n);
  launch_dbg_pcg(seed, n, u.d, f.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  if (out_u32) u.download(out_u32);
  if (out_f32) f.download(out_f32);
  return 0;
}
int pt_dbg_blue_noise(int n, const int* xysf, float* out_uv, float* out_jitter) {
  State& s = dbg_state();
  DevBuf<int> in(xysf, (size_t)n * 4);
  DevBuf<float> uv((size_t)n * 2), jit((size_t)n * 2);
  launch_dbg_blue_noise(n, in.d, uv.d, jit.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  if (out_uv) uv.download(out_uv);
  if (out_jitter) jit.download(out_jitter);
  return 0;
}
int pt_dbg_camera_dirs(int w, int h, int n, const int* xy, const float* jitter, float* out_dirs) {
  State& s = ensure_handle(w, h);
  DevBuf<int> dxy(xy, (size_t)n * 2);
  DevBuf<float> dj(jitter, (size_t)n * 2), dd((size_t)n * 3);
  launch_dbg_camera(s.cam, w, h, n, dxy.d, dj.d, dd.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  dd.download(out_dirs);
  return 0;
}
int pt_trace_rays(int n, const float* origins, const float* dirs, const float* tmin, const float* tmax, int any_hit,
                  int* out_prim, float* out_t, float* out_b1, float* out_b2) {
  return pt_trace_rays_instances(n, origins, dirs, tmin, tmax, any_hit, out_prim, out_t, out_b1, out_b2, nullptr);
}
int pt_trace_rays_instances(int n, const float* origins, const float* dirs, const float* tmin, const float* tmax, int any_hit,
                            int* out_prim, float* out_t, float* out_b1, float* out_b2, int* out_instance) {
  State& s = dbg_state();
  DevBuf<int> inst((size_t)n);
  DevBuf<float> o(origins, (size_t)n * 3), d(dirs, (size_t)n * 3), t0(tmin, (size_t)n), t1(tmax, (size_t)n);
  DevBuf<int> prim((size_t)n);
  DevBuf<float> t((size_t)n), b1((size_t)n), b2((size_t)n);
  HIP_CHECK(hipMemset(t.d, 0, (size_t)std::max(n, 1) * sizeof(float)));
  HIP_CHECK(hipMemset(b1.d, 0, (size_t)std::max(n, 1) * sizeof(float)));
  HIP_CHECK(hipMemset(b2.d, 0, (size_t)std::max(n, 1) * sizeof(float)));
  launch_trace_rays(s.scene, s.bvh_depth, n, o.d, d.d, t0.d, t1.d, any_hit, prim.d, t.d, b1.d, b2.d, inst.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  if (out_instance) inst.download(out_instance);
  if (out_prim) prim.download(out_prim);
  if (out_t) t.download(out_t);
  if (out_b1) b1.download(out_b1);
  if (out_b2) b2.download(out_b2);
  return 0;
}
int pt_dbg_shade(int n, pt_path_rec* recs, const int* prim, const float* b1, const float* b2, int frame, int max_depth,
                 int* out_shadow_traced) {
  State& s = dbg_state();
  if (s.scene.inst_records) return -1;  // single shading steps are offered for flat scenes only
  static_assert(sizeof(pt_path_rec) == sizeof(DbgPathRec), "layout");
  DevBuf<DbgPathRec> r(reinterpret_cast<const DbgPathRec*>(recs), (size_t)n);
  DevBuf<int> p(prim, (size_t)n), sh((size_t)n);
  DevBuf<float> db1(b1, (size_t)n), db2(b2, (size_t)n);
  const int guiding = (g_guiding_enabled && s.guide_valid) ? 1 : 0;
  launch_dbg_shade(s.scene, s.guide, guiding, s.bvh_depth, n, r.d, p.d, db1.d, db2.d, frame,
                   std::max(1, std::min(max_depth, 255)), sh.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  r.download(reinterpret_cast<DbgPathRec*>(recs));
  if (out_shadow_traced) sh.download(out_shadow_traced);
  return 0;
}
int pt_dbg_guiding(int n, const float* positions, const float* u3, const float* wi, int* out_region, int* out_lobe,
                   float* out_dir, float* out_pdf_dir, float* out_pdf_wi) {
  State& s = dbg_state();
  if (!s.guide_valid) return -1;
  DevBuf<float> P(positions, (size_t)n * 3), U(u3, (size_t)n * 3), Wd(wi, (size_t)n * 3);
  DevBuf<int> reg((size_t)n), lobe((size_t)n);
  DevBuf<float> dir((size_t)n * 3), pd((size_t)n), pw((size_t)n);
  launch_dbg_guiding(s.guide, n, P.d, U.d, Wd.d, reg.d, lobe.d, dir.d, pd.d, pw.d, s.stream);
  HIP_CHECK(hipStreamSynchronize(s.stream));
  if (out_region) reg.download(out_region);
  if (out_lobe) lobe.download(out_lobe);
  if (out_dir) dir.download(out_dir);
  if (out_pdf_dir) pd.download(out_pdf_dir);
  if (out_pdf_wi) pw.download(out_pdf_wi);
  return 0;
}

const char* pt_version(void) { return "pt_hip 0.1 (gfx950)"; }

}  // extern "C"
