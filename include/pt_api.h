/*
 * pt_api.h -- C ABI of libpt_hip.so, the MI355X (gfx950) path-tracing backend.
 *
 * Part 1 is the drop-in boundary: the exact symbols the reference's Rust host
 * binds (src/main.rs:9-32, src/guiding.rs:31-68) and that its wrapper DLL
 * exports (optix/optix_wrapper.cpp, optix/openpgl_bridge.cpp).  Signatures,
 * argument meaning and error behaviour are the reference's.
 *
 * Part 2 (pt_*) are additions with no counterpart in the reference: render-mode
 * selection, device-resident / sample-sharded rendering for multi-GPU hosts,
 * the GPU demosaic and EXR output the reference does on the CPU in Rust, ray
 * counters and per-kernel timing for measurement, and stage-level entry points
 * used by the parity tests.
 *
 * Conventions: plain pointers and sizes only.  Unless a name says "device",
 * pointers are host pointers.  All failures of the HIP runtime print to stderr
 * and abort(), like CU_CHECK/OTK_CHECK in optix_wrapper.cpp:48-69; the int
 * return value is 0, or -1 for a NULL output pointer (optix_wrapper.cpp:980).
 * The library is not thread-safe (global singleton renderer, like the
 * reference: optix_wrapper.cpp:923).
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_EXPORT __attribute__((visibility("default")))

/* ====================================================================== *
 * Part 1 -- the reference's exported surface                              *
 * ====================================================================== */

/* optix/optix_wrapper.cpp:1102-1107.  Blocking.  Writes w*h*3 floats (row-major,
 * top row first, interleaved RGB, linear per-pixel radiance means).  Lazily
 * (re)creates the renderer when (w,h) changes (1087-1100) and increments the
 * global frame counter (983), so consecutive calls use different seeds.
 * noise_threshold == 0 disables adaptive sampling.  noise_threshold > 0 runs the noise
 * test of optix_device.cu:1011-1060 in a scheduling-independent form: see
 * pt_set_adaptive_round below.
 * max_depth is clamped to [1, 255] (the path state keeps the depth in 8 bits; the reference
 * takes any int, its own host passes 32: src/main.rs:481). */
PT_EXPORT int optix_render_rgb(int w, int h, int spp, int max_depth, float noise_threshold,
                               int min_adaptive_samples, float* out_rgb);

/* optix/optix_wrapper.cpp:1109-1115.  Same render, writes the w*h Bayer mosaic
 * (value of the CFA channel at each pixel, clamped >= 0; optix_device.cu:1069).
 * pattern: 0 RGGB, 1 BGGR, 2 GRBG, 3 GBRG (optix_device.cu:902-912). */
PT_EXPORT int optix_render_bayer_f32(int w, int h, int spp, int max_depth, float noise_threshold,
                                     int min_adaptive_samples, int pattern, float* out_raw);

/* optix/optix_wrapper.cpp:1132-1164: pinned host memory. */
PT_EXPORT void* optix_alloc_host(size_t bytes);
PT_EXPORT void  optix_free_host(void* ptr);
/* optix/optix_wrapper.cpp:1116-1124: drain the renderer's streams; no-op without a renderer. */
PT_EXPORT void  optix_synchronize(void);

/* optix/optix_wrapper.cpp:927-976: explicit context API (exported, unused by the Rust host). */
PT_EXPORT void* optix_ctx_create(int width, int height);
PT_EXPORT void  optix_ctx_destroy(void* handle);
PT_EXPORT void  optix_ctx_set_bayer(void* handle, int pattern);
PT_EXPORT void  optix_ctx_set_camera(void* handle, float eye_x, float eye_y, float eye_z,
                                     float look_x, float look_y, float look_z,
                                     float up_x, float up_y, float up_z, float vfov_deg);

/* optix/openpgl_bridge.h:28-29 and optix/guiding_gpu.h:24-29 == src/guiding.rs:5-28 */
struct pgl_region { float bmin[3]; float bmax[3]; uint32_t lobe_ofs; uint32_t lobe_num; };
struct pgl_lobe   { float mu[3]; float kappa; float weight; float rgb[3]; };
typedef struct TrainSample { float position[3]; float dir_in[3]; float contrib[3]; uint32_t is_delta; } TrainSample;

/* optix/optix_wrapper.cpp:1168-1299 (declared optix/guiding_gpu.h:35-41). */
PT_EXPORT void guiding_upload_snapshot(const struct pgl_region* regions, uint32_t n_regions,
                                       const struct pgl_lobe* lobes, uint32_t n_lobes);
PT_EXPORT void guiding_set_enabled(int enabled);
PT_EXPORT void guiding_set_train_buffer_size(uint32_t n);
/* Returns a pointer into library-owned storage valid until the next call.  Unlike
 * optix_wrapper.cpp:1279-1285 the count is clamped to the ring capacity, and the records are the
 * ring slots that were written: vertices are stored at hashed slots instead of being appended
 * through one atomic counter, so the ring holds a random subset of the render's non-delta vertices. */
PT_EXPORT void guiding_map_train_samples(const TrainSample** samples, uint32_t* count);
PT_EXPORT void guiding_reset_train_write_idx(void);
PT_EXPORT void guiding_set_grid_res(int x, int y, int z);

/* optix/openpgl_bridge.h:7-33.  OpenPGL is not available to this build; these are a
 * stand-in trainer with the same ABI (per-region vMF moment fit), NOT OpenPGL. */
typedef void* pgl_dev_t;
typedef void* pgl_field_t;
typedef void* pgl_samples_t;
PT_EXPORT pgl_dev_t     pgl_dev_create(int num_threads);
PT_EXPORT void          pgl_dev_destroy(pgl_dev_t);
PT_EXPORT pgl_field_t   pgl_field_create(pgl_dev_t dev, const float world_min[3], const float world_max[3]);
PT_EXPORT void          pgl_field_destroy(pgl_field_t);
PT_EXPORT pgl_samples_t pgl_samples_create(void);
PT_EXPORT void          pgl_samples_destroy(pgl_samples_t);
PT_EXPORT void pgl_samples_add_surface(pgl_samples_t s, const float pos[3], const float dir_in[3],
                                       const float weight_rgb[3], int is_delta);
PT_EXPORT void pgl_field_update(pgl_field_t, pgl_samples_t);
PT_EXPORT void pgl_field_snapshot(pgl_field_t, const struct pgl_region** regions, uint32_t* region_count,
                                  const struct pgl_lobe** lobes, uint32_t* lobe_count);

/* ====================================================================== *
 * Part 2 -- additions (no counterpart in the reference)                   *
 * ====================================================================== */

/* Render mode.  PT_MODE_STRICT replays the reference literally: each 4-sample launch
 * re-derives the seed and restarts the sample index (optix_wrapper.cpp:983-1021,
 * optix_device.cu:945-953), so N spp are N/4 repeats of the same 4 paths.
 * PT_MODE_CONVERGE (default) feeds the global sample index into the jitter hash and
 * continues the PCG stream, so N spp are N distinct samples of the same estimator.
 * Also settable with the environment variable PT_MODE=strict|converge. */
enum { PT_MODE_STRICT = 0, PT_MODE_CONVERGE = 1 };
PT_EXPORT void pt_set_mode(int mode);
PT_EXPORT int  pt_get_mode(void);

/* Integrator pipeline.  PT_PIPELINE_STAGED runs generate / extend / shade / occlude as
 * separate kernels over SoA queues in HBM; PT_PIPELINE_FUSED keeps a path in registers
 * for several bounces per launch.  Same results, bit for bit.  Env: PT_PIPELINE=staged|fused.
 * PT_PIPELINE_AUTO (the default when neither the call nor the variable chooses): scenes whose BVH is
 * at most 256 KiB run fused; larger ones run staged until few paths are left, then finish fused. */
enum { PT_PIPELINE_AUTO = -1, PT_PIPELINE_STAGED = 0, PT_PIPELINE_FUSED = 1 };
PT_EXPORT void pt_set_pipeline(int pipeline);
PT_EXPORT int  pt_get_pipeline(void);

/* Adaptive sampling (noise_threshold > 0).  The reference tests a pixel's noise (relative
 * standard deviation of luminance and of the CFA channel, maximum over the 3x3 neighbourhood)
 * after every 4-sample launch once it has min_adaptive_samples, racing with its neighbours
 * (optix_device.cu:1011-1060, optix_wrapper.cpp:1002-1021).  Here all active pixels run to the
 * next round boundary, then one pass computes every pixel's noise, a second pass reads the 3x3
 * maxima and drops the pixels below the threshold.  samples > 0: rounds of that many samples;
 * samples < 0 (default -32, env PT_ADAPTIVE_ROUND): the first round is -samples and every later
 * round doubles the count (32, 64, 128, ...); strict mode always uses rounds of 4.  No test runs
 * before min_adaptive_samples.  The variance covers all samples taken so far.
 * pt_get_sample_counts copies the last render's per-pixel sample counts (w*h ints, may be
 * NULL) and the number of rounds it ran (may be NULL; 0 for a non-adaptive render). */
PT_EXPORT void pt_set_adaptive_round(int samples);
PT_EXPORT int  pt_get_adaptive_round(void);
/* Adaptive renders split a pixel's samples into several chunks rendered side by side (own random stream
 * each, like the ranks of a sharded render) where that keeps enough paths in flight; every chunk takes the
 * same number of samples per round (scenes that run fused: the last chunk about a quarter of the others' part of a
 * long round, so that a round ends on short slots; the shared-statistics entry point follows the same schedule).  0 (default): as many as 16 M slots allow (8 at 1920x1080), a power of two that
 * divides the round and leaves every chunk at least 4 samples of it (scenes that run staged by default: up to 64,
 * at least 1 sample; see pt_last_render_chunks); 1: one stream per pixel, the sample sequence of the reference's
 * launch loop.  Env: PT_ADAPTIVE_CHUNKS. */
PT_EXPORT void pt_set_adaptive_chunks(int chunks);
PT_EXPORT int  pt_get_adaptive_chunks(void);
PT_EXPORT int  pt_get_sample_counts(int* out_counts, int* out_rounds);

/* Fused pipeline only.  0 (env PT_DEFER_SHADOW=0): shadow rays are traced where they are made, a slot's samples are
 * added in sample order, and the fused and the staged pipeline agree bit for bit.  1: each wave queues its shadow
 * rays in LDS and traces them 64 at a time, at full occupancy of the wave; a slot's total is the same set of terms
 * added in a different order, so images agree with the staged pipeline to rounding (same paths).  -1 (default):
 * chosen per scene -- queued for trees of more than one node and up to 1024 triangles, where it measures 10 %
 * faster, unless the scene is small enough (about 90 triangles) to sit in LDS beside the in-place kernel's data;
 * in place otherwise (the reference's scene: 298 ms in place, 312 ms queued at 1080p, 1024 spp).
 * Adaptive renders always trace in place (the noise test needs per-sample sums). */
PT_EXPORT void pt_set_defer_shadow(int on);
PT_EXPORT int  pt_get_defer_shadow(void);
/* Fused pipeline.  on (default, env PT_PERSIST=1): a fixed-count render, or a round of an adaptive one, is one
 * kernel launch; its waves take slots from a shared counter and a lane that has finished a slot takes the next
 * one, so no path state goes through memory in a fixed-count render (4 % faster at 1024 samples a pixel than launch
 * after launch) and a round's stored slot state is fetched 32 slots at a time into LDS (adaptive renders 5 % faster).
 * off: launch after launch, 32 bounces of every slot each.  The same paths and, with shadow rays traced in place,
 * the same bits either way. */
PT_EXPORT void pt_set_one_launch(int on);
PT_EXPORT int  pt_get_one_launch(void);

/* Select the HIP device the next renderer is created on (default 0, or env PT_DEVICE). */
PT_EXPORT void pt_set_device(int device);

/* Several GPUs behind the reference's own entry points (no counterpart in the reference, which drives one device:
 * optix_wrapper.cpp:788,1138).  With n > 1 (or the environment variable PT_GPUS=n), optix_render_rgb and
 * optix_render_bayer_f32 split every converge-mode render by sample index into n shards -- shard d renders global
 * samples [d*spp/n, (d+1)*spp/n) of every pixel (pt_shard_range), exactly what rank d of a multi-process render
 * renders through pt_render_accumulate_device -- on GPUs first, first + 1, ... (modulo the number of GPUs: more
 * shards than GPUs is allowed and changes only the speed), one host thread per shard; the shards' per-pixel totals
 * are copied to the first GPU (peer copies over xGMI), added there in shard order and divided by the samples
 * taken, so a caller of optix_render_rgb sees one image and nothing else changes.  Adaptive renders are sharded
 * too, with ONE noise test: at every round boundary the shards meet, the first GPU adds the peers' per-pixel
 * statistics (sum, samples, sum of squares: 32 bytes per pixel, peer copies) to its own in shard order and hands the
 * totals back, so noise_threshold and min_adaptive_samples mean what they mean on one GPU and every shard keeps the
 * same pixels active (a shard takes its proportional part of every round, rounded up; pt_get_sample_counts then
 * reports the totals of all shards).  Scene, camera, Bayer pattern and guiding snapshot are replicated;
 * guiding_map_train_samples returns the records of all shards' rings.  Strict mode is never sharded. */
PT_EXPORT void pt_set_device_count(int n);
PT_EXPORT int  pt_get_device_count(void);
PT_EXPORT void pt_shard_range(int spp, int n_shards, int shard, int* begin, int* count);

/* Frame counter: the reference increments Params.frame once per render call
 * (optix_wrapper.cpp:983); tests pin it.  The next render uses frame = value + 1. */
PT_EXPORT void pt_set_frame(int frame);
PT_EXPORT int  pt_get_frame(void);

/* Sample-sharded, device-resident render for multi-GPU hosts.  Renders samples
 * [sample_begin, sample_begin + sample_count) of every pixel in converge mode and ADDS
 * the per-pixel sums into d_sum_rgb (device pointer, w*h*3 floats, caller-zeroed); the
 * caller all-reduces the sums (RCCL) and divides by the total count.  frame is explicit
 * so every rank uses the same one.  A chunk starting at sample_begin > 0 re-seeds the
 * per-pixel PCG stream, so G chunks on one GPU equal G ranks with one chunk each. */
PT_EXPORT int pt_render_accumulate_device(int w, int h, int sample_begin, int sample_count, int max_depth,
                                          int frame, float* d_sum_rgb);
/* pt_render_accumulate_device for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Not a call that can be recorded into a HIP graph: it runs
 * on the null stream and synchronises the device several times per render (a blocking read-back of the live-slot
 * counter after every batch of launches, and once more at the end), and its first call on a device copies the scene.  Same arguments, same
 * result bit for bit (same scene, default camera, fused pipeline), but every byte of device memory besides
 * d_sum_rgb -- path state, slot queues, counters, the scene -- lives in one static array of the library's code
 * object (PT_NOALLOC_MAX_SLOTS * 92 + PT_NOALLOC_MAX_SCENE_BYTES + 512 bytes = 24.2 MB, held by every GPU the
 * library is loaded on until the process ends), and the call makes no hipMalloc / hipFree / hipHostMalloc /
 * stream or event creation: work runs on the null stream, the live-slot counter is read back with a blocking
 * copy after every batch of launches.  Limits: w * h * chunks <= PT_NOALLOC_MAX_SLOTS, where chunks is what
 * pt_render_accumulate_device would use (min(8, 33554432 / (w*h)), lowered until every chunk has 32 samples, the last one cut into up to four, so
 * 1 for sample_count < 64); the scene's BVH nodes, leaf items and shading records together at most
 * PT_NOALLOC_MAX_SCENE_BYTES (the built-in scene: 3.6 KB); guiding off.  Returns 0, -1 for a NULL output, -2
 * with the reason on stderr when a limit is exceeded (nothing is rendered then).  Every call on a GPU uses
 * that GPU's one array: calls that overlap on the same GPU (two host threads, two processes do not share it)
 * race; the array keeps the previous call's contents and the call clears what it needs. */
#define PT_NOALLOC_MAX_SLOTS (1u << 18)
#define PT_NOALLOC_MAX_SCENE_BYTES (64u << 10)
PT_EXPORT int pt_render_accumulate_device_noalloc(int w, int h, int sample_begin, int sample_count, int max_depth,
                                                  int frame, float* d_sum_rgb);
/* The same with adaptive sampling: this share's samples stop per pixel by the noise test run on the
 * share's own statistics (noise_threshold <= 0: fixed count).  ADDS the sums into d_sum_rgb (w*h*3) and
 * the number of samples each pixel took into d_count (w*h floats), both device pointers, caller-zeroed;
 * after the all-reduce of both, mean = sum / count. */
PT_EXPORT int pt_render_adaptive_accumulate_device(int w, int h, int sample_begin, int sample_count, int max_depth,
                                                   int frame, float noise_threshold, int min_adaptive_samples,
                                                   float* d_sum_rgb, float* d_count);
/* Adaptive share with ONE noise test for all ranks (the progressive reduce the per-launch test of
 * optix_device.cu:1011-1060 needs once the samples of a pixel live on several GPUs).  Every rank calls this
 * with its own [sample_begin, sample_begin + sample_count) and the same total_spp (samples per pixel of all
 * ranks together), frame, threshold and minimum.  Rounds follow pt_set_adaptive_round counted in samples of
 * all ranks; a rank renders its proportional part of each round.  At every round boundary, and once more
 * when the render is finished, the library writes 8 floats per pixel into d_stats (device, w*h*8: sum r g b,
 * samples taken, sum of squares r g b, 0), waits for its stream and calls reduce(user), which must replace
 * d_stats by its sum over all ranks (an in-place all-reduce) and return 0 once the result is visible to the
 * device; a NULL reduce means one rank.  The noise test then runs on the totals, so all ranks keep the same
 * pixels active and make the same number of reduce calls.  On return d_stats holds the totals of the
 * finished render: mean = sum / samples.  Returns 0, -1 for bad arguments (NULL d_stats, threshold <= 0,
 * sample_count <= 0 or > total_spp), or the first non-zero value reduce returned. */
typedef int (*pt_reduce_fn)(void* user);
PT_EXPORT int pt_render_adaptive_shared_device(int w, int h, int sample_begin, int sample_count, int total_spp,
                                               int max_depth, int frame, float noise_threshold,
                                               int min_adaptive_samples, float* d_stats, pt_reduce_fn reduce,
                                               void* user);
/* d_rgb (w*h*3, device) -> mosaic d_raw (w*h, device); value = max(channel / divisor, 0). */
PT_EXPORT int pt_mosaic_device(const float* d_rgb, int w, int h, int pattern, float divisor, float* d_raw);

/* Demosaic (src/main.rs:190-415) on the GPU.  algo 0 = gradient method ("AMaZE" in the
 * reference), 1 = bilinear.  Host-pointer and device-pointer forms; the result is the reference's, bit for bit, at
 * every image size and pattern (a pattern outside 0..3 is clamped, as the reference's cfa_at clamps it).
 * Returns 0, or -1 for a NULL buffer or an empty image. */
PT_EXPORT int pt_demosaic_f32(const float* raw, int w, int h, int pattern, int algo, float* out_rgb);
PT_EXPORT int pt_demosaic_device(const float* d_raw, int w, int h, int pattern, int algo, float* d_out_rgb);

/* src/main.rs:420-435: write w*h RGB f32 (times exposure) as a scanline OpenEXR file. */
PT_EXPORT int pt_write_exr(const char* path, int w, int h, const float* rgb, float exposure);
/* Display copy of a render as a PNG: rgb * exposure clamped to [0, 1], sRGB transfer function, rounded to the
 * nearest of 2^bits - 1 codes; bits = 8 or 16.  The reference ships such a file (pt.png) and still lists an image
 * encoder in Cargo.toml:8-13, but its writer is gone from src/main.rs; arguments follow save_exr_linear
 * (src/main.rs:420-435).  Returns 0, -1 for bad arguments, -2 for an image beyond 2 GiB of samples, -3 / -4 when
 * compression or the file fails. */
PT_EXPORT int pt_write_png(const char* path, int w, int h, const float* rgb, float exposure, int bits);
/* Minimal reader for files written by pt_write_exr (tests). Returns 0 on success. */
PT_EXPORT int pt_read_exr(const char* path, int* w, int* h, float* rgb_or_null, size_t capacity_floats);

/* Measurement.  Ray = one BVH query (radiance or shadow).  Counters accumulate over
 * render calls until reset. */
PT_EXPORT void pt_get_ray_counts(uint64_t* radiance_rays, uint64_t* shadow_rays);
PT_EXPORT void pt_reset_ray_counts(void);
/* Per-kernel GPU time measured with HIP events on the render stream.  Index: 0 generate,
 * 1 extend, 2 shade, 3 occlude, 4 fused bounce, 5 demosaic.  Timing is sampled on every
 * launch when enabled (pt_set_kernel_timing(1)); off by default. */
enum { PT_K_GENERATE = 0, PT_K_EXTEND = 1, PT_K_SHADE = 2, PT_K_OCCLUDE = 3, PT_K_BOUNCE = 4, PT_K_DEMOSAIC = 5, PT_K_COUNT = 6 };
PT_EXPORT void pt_set_kernel_timing(int every_nth_launch);
PT_EXPORT void pt_get_kernel_stats(int kernel, double* total_ms, uint64_t* timed_launches, uint64_t* launches,
                                   uint64_t* work_items);
PT_EXPORT void pt_reset_kernel_stats(void);
/* Section counters of a library built with -DPT_PROFILE (make prof): out64[2k] = times a wave ran
 * section k of the fused kernel, out64[2k+1] = active lanes summed over those; cleared by the call.
 * out64 holds 64 words.  Returns the number of sections, 0 from the shipped build. */
PT_EXPORT int pt_dbg_profile_counters(uint64_t* out64);
/* Blocks of 256 threads the runtime keeps resident per compute unit for the fused kernel of the current renderer and
 * scene (adaptive != 0: the variant that adaptive renders use), and the LDS bytes one block takes.  -1 without a renderer. */
PT_EXPORT int pt_dbg_fused_occupancy(int adaptive, unsigned long long* lds_bytes);
/* Wall time of the device work of the last render call (HIP events around the whole loop). */
PT_EXPORT double pt_last_render_ms(void);
/* Chunks the last render split every pixel's sample range into (each chunk is a path slot with a random stream of
 * its own, all in flight together).  Fixed-count renders of scenes that run fused: min(8, 33554432 / (w*h)), lowered
 * until every chunk has 32 samples, the last of them then cut into up to four shorter ones (so 11 for 8: the render
 * ends on short slots; env PT_TAIL_SPLIT=0 keeps them equal).  Scenes that run staged by default (trees beyond 256 KiB, instances) keep more
 * in flight: min(64, 150994944 / (w*h)) chunks of as little as one sample, path state of 164 bytes per slot
 * growing on demand up to a quarter of the free device memory.  strict_parity: 1. */
PT_EXPORT int pt_last_render_chunks(void);
/* The rule itself, without a GPU (tests): chunks a fixed-count converge-mode render of w x h pixels and sample_count
 * samples would take on a scene that runs staged by default (staged_scene != 0) or fused, in a renderer that holds
 * have_chunks slots per pixel while free_bytes of device memory are free.  -1 for arguments out of range. */
PT_EXPORT int pt_dbg_chunk_plan(int w, int h, int sample_count, int staged_scene, int have_chunks,
                                unsigned long long free_bytes);

/* Scene.  The built-in scene is the reference's Cornell box (optix_wrapper.cpp:379-489).
 * pt_scene_synthetic replaces it by the box plus a seeded grid of triangulated spheres
 * (about n_objects * 20 * 4^subdiv triangles) to stress BVH traversal out of HBM;
 * pt_scene_triangles installs caller geometry (materials: 11 floats per triangle in
 * HitData order, optix_wrapper.cpp:162-170; NULL = white diffuse) and returns 0, or -1 with the reason on
 * stderr and the scene unchanged when a count is negative, an array NULL, an index >= n_vertices or a value not
 * finite.  pt_scene_cornell
 * restores the default.  All take effect at the next render. */
PT_EXPORT void pt_scene_cornell(void);
PT_EXPORT void pt_scene_synthetic(uint32_t seed, int n_objects, int subdiv);
PT_EXPORT int  pt_scene_triangles(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                                  const float* materials11);
/* Scenes made of instances: meshes in their own object space, each placed any number of times by an affine
 * transform, under a two-level tree (a top level over the instances' world boxes, one tree per mesh).  The reference's
 * pipeline allows single-level instancing (optix_wrapper.cpp:565-566) and never uses it; its one acceleration
 * structure is built over flat triangles (optix_wrapper.cpp:492-544).
 *   pt_scene_instanced_begin()                        start describing a scene (the reference's light set-up)
 *   pt_scene_add_mesh(...) -> mesh id                 arrays as for pt_scene_triangles, object space
 *   pt_scene_add_instance(mesh, xform12, material11)  xform12: object -> world, three rows of (x y z | translation);
 *                                                     material11 (may be NULL): one material for the whole instance
 *                                                     in place of the mesh's per-triangle ones; -> instance id
 *   pt_scene_instanced_commit() -> instances          the described scene becomes the current one
 * -1 on bad input (nothing changes).  Such scenes render through the staged kernels; hits carry the instance, the
 * ray is taken to object space with its direction as it is (distances stay the world ray's), normals come back through
 * the inverse transpose.  pt_scene_flatten() replaces the current scene by its instances written out as world-space
 * triangles (returns the triangle count); pt_scene_synthetic_instanced is pt_scene_synthetic with the room as one
 * instance and every sphere an instance of one unit icosphere.  pt_trace_rays_instances is pt_trace_rays that also
 * returns the instance of each closest hit (-1 for flat scenes or a miss). */
PT_EXPORT int  pt_scene_instanced_begin(void);
PT_EXPORT int  pt_scene_add_mesh(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                                 const float* materials11);
PT_EXPORT int  pt_scene_add_instance(int mesh, const float* xform12, const float* material11);
PT_EXPORT int  pt_scene_instanced_commit(void);
PT_EXPORT int  pt_scene_flatten(void);
PT_EXPORT int  pt_scene_instance_count(void);
PT_EXPORT void pt_scene_synthetic_instanced(uint32_t seed, int n_objects, int subdiv);
PT_EXPORT int  pt_trace_rays_instances(int n, const float* origins, const float* dirs, const float* tmin, const float* tmax,
                                       int any_hit, int* out_prim, float* out_t, float* out_b1, float* out_b2,
                                       int* out_instance);
/* pgl_samples_add_surface for `count` records in the layout guiding_map_train_samples returns (the host loop
 * of src/guiding.rs:196-219 in one call); used by multi-GPU hosts that gather every rank's ring on one rank. */
PT_EXPORT void pt_pgl_samples_add_batch(pgl_samples_t samples, const TrainSample* records, uint32_t count);

/* Training from statistics binned on the GPU.  The reference copies the whole training ring to the host (10.5 MB)
 * and adds its records one call at a time (src/guiding.rs:194-220); the trainer here only ever needs, per region and
 * lobe of its field, sums over the samples assigned to it.  pt_guiding_bin_train_samples_device takes those sums on
 * the device: every valid record of the current renderer's ring is assigned as pgl_field_update would assign it
 * (region by position, lobe by the largest dot(dir_in, axis)) and added to a table of 64-bit fixed-point integers,
 *   d_bins[(region * PT_PGL_BIN_LOBES + lobe) * PT_PGL_BIN_WORDS + k],
 *   k = 0..2 sum of weight * dir_in, 3 sum of weight, 4..6 sum of contrib (x 2^PT_PGL_BIN_SHIFT), 7 number of samples,
 * in device memory (the caller zeroes it; regions = product of the field's resolution, pt_pgl_field_layout).
 * Integer sums do not depend on the order of addition: the tables of several GPUs can be added (RCCL all-reduce of
 * int64) and every rank that applies the total to its own copy of the field gets the same field, bit for bit.
 * pt_pgl_field_update_binned applies a table (host memory) to the field; pt_pgl_field_update_from_ring does
 * bin + copy + update + ring reset for the current renderer (and its sample shards) in one call.
 * Return 0, or -1 without a renderer, ring or field. */
#define PT_PGL_BIN_LOBES 6
#define PT_PGL_BIN_WORDS 8
#define PT_PGL_BIN_SHIFT 24
/* What the fixed-point table represents exactly, and therefore what both forms of the trainer accept: a training
 * record is used when its weight (the mean of contrib) lies in [PT_PGL_MIN_WEIGHT, PT_PGL_MAX_WEIGHT] and no
 * component of contrib exceeds PT_PGL_MAX_WEIGHT in magnitude; any other record (a firefly of 1e12, a weight that
 * would round to zero steps of 2^-24, a NaN) is dropped by pgl_field_update and by the binning kernel alike, so the
 * two never diverge.  A bin then holds 2^19 records of the largest weight before its 64-bit sum could wrap; the
 * training ring holds 2^18. */
#define PT_PGL_MAX_WEIGHT 1048576.0f          /* 2^20 */
#define PT_PGL_MIN_WEIGHT 2.98023223876953125e-8f /* 2^-25: half a step of the table */
PT_EXPORT int  pt_pgl_field_layout(pgl_field_t field, float bmin[3], float bmax[3], int res[3], float* lobe_axes,
                                   uint32_t* regions);
PT_EXPORT void pt_pgl_field_update_binned(pgl_field_t field, const long long* bins);
PT_EXPORT int  pt_guiding_bin_train_samples_device(pgl_field_t field, long long* d_bins);
PT_EXPORT int  pt_pgl_field_update_from_ring(pgl_field_t field);

/* Scene input beyond the built-in box (the reference has none: its quads are hard-coded,
 * optix_wrapper.cpp:379-489).  pt_scene_load_obj reads a Wavefront OBJ file and the material library it names:
 * v, f (polygons are fanned; v, v/vt, v//vn, v/vt/vn; negative indices), usemtl; Kd = base colour, Ke =
 * emission, Ni = index of refraction, Pr = roughness (else from Ns), d / Tr = opacity (d < 1: dielectric with
 * transmission 1 - d).  Returns the number of triangles, or -1 with the reason on stderr (the current scene is
 * then kept).  pt_scene_save_obj writes the current scene as OBJ + MTL (same path, extension .mtl) with
 * enough digits to read back the same bits; 0 or -1.  Next-event estimation samples one rectangle light in a
 * plane of constant z, as in the reference (optix_wrapper.cpp:842-845, optix_device.cu:495-548):
 * pt_scene_set_light replaces its centre, emission, normal and half extents along x and y (NULL keeps a
 * field); scene setters reset it to the reference's. */
PT_EXPORT int  pt_scene_load_obj(const char* path);
/* Sizes of the current scene on the host (the arrays pt_scene_get fills); needs no GPU. */
PT_EXPORT void pt_scene_counts(int* n_vertices, int* n_triangles);
PT_EXPORT int  pt_scene_save_obj(const char* path);
PT_EXPORT void pt_scene_set_light(const float pos[3], const float emit[3], const float normal[3], const float half[2]);
/* Leaf items of the BVH: 1 (default) lets two consecutive triangles that form a parallelogram -- a quad
 * split along a diagonal -- share one traversal record and one plane intersection, 0 keeps every triangle
 * on its own, < 0 returns to the default (environment PT_BVH_PAIRS).  Results are the same primitives and
 * distances either way; the tree is rebuilt at the next render.  No counterpart in the reference (the
 * driver's GAS build is opaque, optix_wrapper.cpp:492-544). */
PT_EXPORT void pt_set_bvh_pairing(int on);
PT_EXPORT void pt_scene_info(int* n_triangles, int* n_bvh_nodes, int* bvh_depth, size_t* bvh_bytes);
/* Copy out the active scene (tests compare it with the oracle's). Any pointer may be NULL. */
PT_EXPORT void pt_scene_get(float* vertices, uint32_t* indices, float* normals, float* materials11);

/* Stage-level entry points (host pointers) used by the parity tests. */
/* a3/a4: first n outputs of the RNG pieces. */
PT_EXPORT int pt_dbg_pcg(uint64_t seed, int n, uint32_t* out_u32, float* out_f32);
PT_EXPORT int pt_dbg_blue_noise(int n, const int* xysf /* n*4: x,y,s,frame */, float* out_uv /* n*2 */, float* out_jitter /* n*2 */);
/* a5: camera directions for n pixels with given jitter (n*2). */
PT_EXPORT int pt_dbg_camera_dirs(int w, int h, int n, const int* xy, const float* jitter, float* out_dirs);
/* a7: trace n rays through the BVH kernel.  any_hit = 0: closest hit -> prim,t,b1,b2
 * (prim = -1 on miss); any_hit = 1: prim = 1 if occluded else 0. */
PT_EXPORT int pt_trace_rays(int n, const float* origins, const float* dirs, const float* tmin, const float* tmax,
                            int any_hit, int* out_prim, float* out_t, float* out_b1, float* out_b2);
/* a8: one shading step per record.  Layout mirrors PRD (optix_device.cu:227-242). */
typedef struct pt_path_rec {
  float radiance[3]; float throughput[3]; float origin[3]; float direction[3];
  uint64_t seed; int depth; int done; float prev_pdf_bsdf; int prev_pdf_valid;
  int px, py, sample, rng_dim;
} pt_path_rec;
PT_EXPORT int pt_dbg_shade(int n, pt_path_rec* recs, const int* prim, const float* b1, const float* b2,
                           int frame, int max_depth, int* out_shadow_traced);
/* a13: guiding math on the device for n queries against the uploaded snapshot. */
PT_EXPORT int pt_dbg_guiding(int n, const float* positions, const float* u3 /* n*3 */, const float* wi /* n*3 */,
                             int* out_region, int* out_lobe, float* out_dir /* n*3 */, float* out_pdf_dir,
                             float* out_pdf_wi);

PT_EXPORT const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif
