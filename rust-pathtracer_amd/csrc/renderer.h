// renderer.h -- host runtime behind the C ABI: device state, scene upload, wavefront loop.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/pt_api.h"
#include "bvh_build.h"
#include "pt_kernels.h"
#include "scene.h"

namespace pt {

#define HIP_CHECK(call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "HIP error %s (%d): %s at %s:%d\n", hipGetErrorName(e_), (int)e_,          \
              hipGetErrorString(e_), __FILE__, __LINE__);                                        \
      std::abort();                                                                              \
    }                                                                                            \
  } while (0)

struct KernelStat {
  double total_ms = 0.0;
  uint64_t timed = 0, launches = 0, work = 0;
};

struct TimedLaunch {
  int kernel;
  hipEvent_t start, stop;
};

// Everything that lives on one GPU for one resolution (the reference's State,
// optix_wrapper.cpp:221-367, minus the OptiX pipeline objects).
struct State {
  int device = 0;
  int width = 0, height = 0;
  int num_cus = 256;
  int max_chunks = 1;   // slots allocated per pixel (grows when a render of a large scene wants more)
  int max_chunks_split = 1;  // arrays of a fixed size (borrowed): slots per pixel a plan with a cut last chunk may take
  int base_chunks = 1;  // what a renderer of this resolution starts with: the limit for adaptive renders
  int last_chunks = 1;  // chunks used by the most recent render
  hipStream_t stream = nullptr, copy_stream = nullptr;
  CameraView cam{};
  int bayer_pattern = 0;

  // scene on device
  BvhNode* d_nodes = nullptr;
  q4* d_tris = nullptr;
  ShadeRec* d_shade = nullptr;
  u4* d_wide = nullptr;            // small trees: wide nodes and the primitives' places in them (bvh_build.h)
  uint32_t* d_home_wide = nullptr;
  SceneView scene{};
  int bvh_depth = 1, num_nodes = 0;
  size_t bvh_bytes = 0;
  int num_instances = 0;
  uint64_t scene_version = 0;

  // path state / queues
  q4 *st_o = nullptr, *st_d = nullptr, *st_s = nullptr, *acc = nullptr, *hits = nullptr;
  uint32_t* hit_inst = nullptr;    // per queue slot: the instance of the hit (instanced scenes)
  q4* d_inst = nullptr;            // instanced scenes: 6 q4 per instance (SceneView::inst_records)
  q4 *sh_o = nullptr, *sh_d = nullptr, *sh_c = nullptr;
  uint32_t* sh_count = nullptr;
  u4* st_m = nullptr;
  int* started = nullptr;
  uint32_t* queue_tmp = nullptr;  // staged pipeline: scratch of the stable queue compaction
  uint32_t* tiles = nullptr;
  uint32_t* queue[2] = {nullptr, nullptr};
  uint32_t* counters = nullptr;
  unsigned long long* ray_counters = nullptr;
  uint32_t* h_counters = nullptr;  // pinned, 3 slots x CNT_WORDS (two read-back slots, one staging)
  hipEvent_t ev_slot[2] = {nullptr, nullptr};
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;

  // adaptive sampling (allocated on first use)
  q4* acc2 = nullptr;
  AdaptiveBuffers adaptive{};
  int adaptive_rounds = 0;
  q4 *shared_acc = nullptr, *shared_acc2 = nullptr;  // statistics summed over ranks (shared noise test)
  int* shared_n = nullptr;

  // sample shards rendered side by side in one process (pt_set_device_count): this shard's totals, on shard 0 the
  // landing area for the other shards' totals; a shard other than 0 keeps out of the global statistics
  q4* d_sum4 = nullptr;
  q4* d_peer = nullptr;
  size_t peer_capacity = 0;  // q4 elements
  float* d_shard_stats = nullptr;  // in-library sample shards, adaptive: this shard's (sum, n, sum of squares) per pixel, 8 floats
  float* d_stats_in = nullptr;     // ... on the first shard: a peer's statistics while they are added
  bool counts_from_stats = false;  // the last render's per-pixel sample counts are the totals in d_shard_stats
  bool quiet = false;

  // outputs
  float* d_out_rgb = nullptr;
  float* d_out_bayer = nullptr;

  // guiding
  TrainRec* d_train = nullptr;
  uint32_t* d_train_idx = nullptr;
  uint32_t train_capacity = 0;
  GuideRegionDev* d_regions = nullptr;
  GuideLobeDev* d_lobes = nullptr;
  GuideRegionDev* d_cells = nullptr;
  uint32_t* d_grid = nullptr;
  GuideView guide{};
  bool guide_valid = false;
  uint64_t guide_version = 0;

  std::vector<TimedLaunch> timed;
  double last_render_ms = 0.0;

  // true: every device pointer above points into memory this State does not own (the allocation-free entry
  // point's scratch array), work runs on the null stream, h_counters is ordinary host memory and there are no
  // events; the destructor then releases nothing.
  bool borrowed = false;

  ~State();
};

}  // namespace pt
