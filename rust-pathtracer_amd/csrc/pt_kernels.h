// pt_kernels.h -- launch interface between the host runtime and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "pt_types.h"
#include "pt_guiding.h"

namespace pt {

// Device counters (uint32 words) shared by the wavefront kernels.
// CNT_EXTEND / CNT_OCCLUDE: how much of the queue the waves of k_extend / k_occlude have taken so far (k_generate clears both)
// CNT_GRAB: how much of the queue the waves of the one-launch fused kernel have taken (zero when it starts)
enum { CNT_Q0 = 0, CNT_Q1 = 1, CNT_SHADOW = 2, CNT_EXTEND = 3, CNT_OCCLUDE = 4, CNT_Q2 = 5, CNT_GRAB = 6, CNT_WORDS = 8 };

// Everything a wavefront kernel needs, passed by value (lives in the kernarg segment).
struct RenderParams {
  SceneView scene;
  GuideView guide;
  CameraView cam;
  int width, height;
  int frame, max_depth;
  int sample_begin, sample_count;  // this call renders global samples [begin, begin+count) per pixel
  uint32_t npx;                    // width * height
  uint32_t npx_inv, width_inv;     // ceil(2^32 / npx), ceil(2^32 / width): division by multiplication (decode_slot)
  int chunks, chunk_spp;           // the range is split into `chunks` slots per pixel: the first chunk_long of them
  int chunk_long, chunk_short;     // chunk_spp samples long, the rest chunk_short (decode_slot)
  int round_limit;                 // > 0: no slot goes beyond this many samples yet (adaptive rounds);
  int round_limit_b, round_split;  // ... chunks from round_split on: round_limit_b (the last chunk's smaller share)
  int strict;                      // 1: replay the reference's 4-sample launches literally
  int guiding;
  int stack_depth;                 // traversal stack entries per lane
  uint32_t scene_nodes;            // number of BVH nodes
  int scene_in_lds;                // stage a small scene in LDS (fused kernel)
  int shared_ranges;               // ... their waves take queue ranges from a shared counter (long rays) / hold one fixed share each
  int refill_lanes;                // staged traversal kernels: idle lanes that trigger taking new rays
  int defer_shadow;                // fused kernel: queue shadow rays per wave and trace them 64 at a time
  int fresh_queue;                 // one-launch fused kernel: every queued slot is untouched (nothing to load)
  int last_pass;                   // ... and none will run again (only totals are written back)
  // per-pixel path state, SoA of 16-byte words
  q4* st_o;   // origin.xyz, throughput.x
  q4* st_d;   // direction.xyz, throughput.y
  q4* st_s;   // sample sum.xyz, throughput.z
  u4* st_m;   // seed lo, seed hi, prev_pdf bits, flags
  q4* acc;    // per-pixel sum of finished samples (xyz)
  q4* acc2;   // per-pixel sum of squared sample values (adaptive sampling only, else null)
  int* started;  // samples started per pixel
  // queues
  uint32_t* queue[2];
  uint32_t* queue_tmp;     // staged pipeline: survivors of a generate pass, compacted per tile of 1024 entries
  uint32_t* tiles;         // ... their counts per tile, then the tiles' places in the next queue
  uint32_t* counters;      // CNT_WORDS words
  unsigned long long* ray_counters;  // [0] radiance rays, [1] shadow rays
  q4* hits;                // per queue slot: prim bits, t, b1, b2
  uint32_t* hit_inst;      // per queue slot: the instance the hit primitive belongs to (instanced scenes)
  q4* sh_o;                // shadow queue: origin.xyz, tmax
  q4* sh_d;                //               dir.xyz, pixel bits
  q4* sh_c;                //               contrib.xyz
  uint32_t* sh_count;      // shadow rays of every 256 queue entries, stored at the front of those 256 records
  // guiding training ring (optix_device.cu:10-12)
  TrainRec* train;
  uint32_t* train_write_idx;
  uint32_t train_capacity;
};

// is_delta value of a training-ring slot that no vertex has been written to (the ring is cleared to 0xff bytes)
constexpr uint32_t kTrainEmpty = 0xffffffffu;

struct LaunchCfg {
  int blocks, threads;
  hipStream_t stream;
};

// One wavefront iteration, staged: qin -> generate -> qout; extend / shade / occlude over qout
// (occlude also clears qin's counter for the next generate pass).
void launch_generate(const RenderParams& p, int qin, const LaunchCfg& c);
void launch_bin_train(const TrainRec* ring, uint32_t capacity, const BinField& field, const float* d_axes, long long* d_bins,
                      hipStream_t stream);
void launch_extend(const RenderParams& p, int q, const LaunchCfg& c);
void launch_shade(const RenderParams& p, int q, const LaunchCfg& c);
void launch_occlude(const RenderParams& p, int q, const LaunchCfg& c);
// Fused: up to `bounces` bounces per path in registers, regenerating finished samples in place.
// compact: append survivors to queue qin^1 (the caller swaps), else process queue qin in place.
// The length of queue qin is read from counters[cnt.in]; a compacting launch counts its survivors in counters[cnt.out],
// which must be zero when it starts; every launch zeroes counters[cnt.clear] (a counter it neither reads nor writes).
// The fused launches of a render rotate through CNT_Q0, CNT_Q1, CNT_Q2, so that every launch can compact without a
// memset in between: the one after it appends to the counter this one cleared.
struct QueueCounters {
  int in, out, clear;
};
inline int next_queue_counter(int c) { return c == CNT_Q0 ? CNT_Q1 : (c == CNT_Q1 ? CNT_Q2 : CNT_Q0); }
void launch_bounce(const RenderParams& p, int qin, int bounces, int compact, QueueCounters cnt, const LaunchCfg& c);
// Fused, the whole queue in one launch: every slot of queue qin (length in counters[cnt_in]) runs to its end; lanes take
// queue entries from counters[CNT_GRAB], which must be zero at the start.  Launch with about as many blocks as stay resident.
bool bounce_persist_usable(const RenderParams& p);
bool persist_in_place_stages_scene(const RenderParams& p, int threads);
void launch_bounce_persist(const RenderParams& p, int qin, int cnt_in, const LaunchCfg& c);

// Adaptive sampling: active-pixel lists (ping-pong), the 3x3-dilated noise map and each pixel's own noise.
struct AdaptiveBuffers {
  uint32_t* active[2];
  uint32_t* count;      // one word: length of the list k_noise_select writes (cleared by the caller)
  uint32_t* noise_map;  // float bits, compared as unsigned (values are >= 0)
  float* own_noise;
  int* total;           // samples of the pixel over all its slots, as of the last test
};
// One evaluation at a round boundary: reads active[which] (n_active pixels), writes the pixels that stay
// active to active[which ^ 1] and their number to *count.
// acc / acc2 / started hold `chunks` slots per pixel (slot = chunk * w*h + pixel), added up in chunk order.
void launch_noise_round(const AdaptiveBuffers& a, int which, uint32_t n_active, const q4* acc, const q4* acc2,
                        const int* started, int chunks, int w, int h, int pattern, int min_samples, int spp,
                        float threshold, hipStream_t s);
// queue[i * chunks + c] = active[i] + c * npx: the slots of the active pixels
void launch_expand_active(const uint32_t* active, uint32_t n_active, int chunks, uint32_t npx, uint32_t* queue, hipStream_t s);

// per-pixel statistics as 8 floats (sum rgb, samples, sum of squares rgb, 0) and back; count = round(samples)
void launch_pack_stats(const q4* acc, const q4* acc2, const int* started, int chunks, float* stats, uint32_t npx, hipStream_t s);
void launch_unpack_stats(const float* stats, q4* acc, q4* acc2, int* count, uint32_t npx, hipStream_t s);

// packed statistics (8 floats per pixel): own += peer over n floats; mean image = sum / samples taken
void launch_add_stats(float* own, const float* peer, size_t n, hipStream_t s);
void launch_stats_mean(const float* stats, float* out_rgb, uint32_t npx, hipStream_t s);
// resident blocks per CU of the fused kernel launch_bounce picks for p (unguided, no training ring), and its LDS bytes
int bounce_occupancy(const RenderParams& p, int threads, size_t* lds_out, bool one_launch);
// profiling build (-DPT_PROFILE) only: section counters, (waves, lanes) pairs; returns the number of sections
int read_profile_counters(unsigned long long* out64);
void launch_iota(uint32_t* dst, uint32_t n, hipStream_t s);
// per pixel, over its `chunks` slots: mode 0 writes the mean (sum / samples), mode 1 adds the raw sums
void launch_finalize(const q4* acc, const int* started, float* out_rgb, uint32_t npx, int chunks, int mode, hipStream_t s);
// one shard's totals per pixel {sum r, g, b, samples} over its chunks; and the mean over shard `own` and npeers more
// (peers: npeers arrays of npx q4 one after the other), added in that order
void launch_shard_sums(const q4* acc, const int* started, q4* sum4, uint32_t npx, int chunks, hipStream_t s);
void launch_reduce_shards(const q4* own, const q4* peers, int npeers, float* out_rgb, uint32_t npx, hipStream_t s);
// count[p] += samples the pixel took (over its chunks)
void launch_add_counts(const int* started, float* count, uint32_t npx, int chunks, hipStream_t s);
// mosaic: raw[p] = max(rgb[3p + ch(p)] / divisor, 0)
void launch_mosaic(const float* rgb, float* raw, int w, int h, int pattern, float divisor, hipStream_t s);

// demosaic passes (pt_demosaic.hip)
// d_tmp: 3*w*h floats of scratch for the pass-by-pass form of builds with -DPT_DEMOSAIC_PASSES, else unused (may be null)
void launch_demosaic(const float* d_raw, float* d_out, float* d_tmp, int w, int h, int pattern,
                     int algo, hipStream_t s);

// stage-level debug kernels
void launch_dbg_pcg(unsigned long long seed, int n, uint32_t* out_u, float* out_f, hipStream_t s);
void launch_dbg_blue_noise(int n, const int* xysf, float* uv, float* jitter, hipStream_t s);
void launch_dbg_camera(CameraView cam, int w, int h, int n, const int* xy, const float* jitter, float* dirs, hipStream_t s);
void launch_trace_rays(SceneView sc, int stack_depth, int n, const float* o, const float* d, const float* tmin,
                       const float* tmax, int any_hit, int* prim, float* t, float* b1, float* b2, int* inst, hipStream_t s);
struct DbgPathRec {  // == pt_path_rec in include/pt_api.h
  float radiance[3]; float throughput[3]; float origin[3]; float direction[3];
  unsigned long long seed; int depth; int done; float prev_pdf_bsdf; int prev_pdf_valid;
  int px, py, sample, rng_dim;
};
void launch_dbg_shade(SceneView sc, GuideView gv, int guiding, int stack_depth, int n, DbgPathRec* recs, const int* prim,
                      const float* b1, const float* b2, int frame, int max_depth, int* shadow_traced, hipStream_t s);
void launch_dbg_guiding(GuideView gv, int n, const float* pos, const float* u3, const float* wi, int* region, int* lobe,
                        float* dir, float* pdf_dir, float* pdf_wi, hipStream_t s);

}  // namespace pt
