/*
 * pt_oracle.h -- CPU restatement of the reference path tracer's live code path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing the product ships (the HIP library, the
 * C++ driver) includes, links or calls this.  It exists so that tests can check
 * the HIP path and so that bench.py can time a CPU baseline next to the GPU.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  Arithmetic is scalar fp32, compiled without FMA
 * contraction and without fast-math.
 *
 * Parity pin status: the reference ships exactly four executable known-answer
 * tests (src/main.rs:566-601, cfa_at) -- they are restated in
 * tests/test_oracle_golden.py and pass against orc_cfa_at.  The committed
 * renders pt.png/pt_bayer.png pin scene layout qualitatively.  Everything else
 * (ray casting inside the proprietary OptiX runtime, OpenPGL training) has no
 * reference-supplied vectors: "parity unpinned" for those.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_f3;

/* optix/optix_device.cu:62-71 (HitgroupData) == optix/optix_wrapper.cpp:162-170 */
typedef struct {
  float base[3];
  float metallic;
  float roughness;
  float ior;
  float alpha;
  float transmission;
  float emission[3];
} orc_material;

/* optix/guiding_gpu.cuh:8-17, optix/openpgl_bridge.h:28-29 */
typedef struct { float bmin[3]; float bmax[3]; uint32_t lobe_ofs; uint32_t lobe_num; } orc_region;
typedef struct { float mu[3]; float kappa; float weight; float rgb[3]; } orc_lobe;
/* optix/guiding_gpu.cuh:24-29 */
typedef struct { float position[3]; float dir_in[3]; float contrib[3]; uint32_t is_delta; } orc_train_sample;

/* Host-side picture of optix/guiding_gpu.cuh:18-22 (GuideGPU). */
typedef struct {
  const orc_region* regions; uint32_t region_count;
  const orc_lobe*   lobes;   uint32_t lobe_count;
  const uint32_t*   grid;
  int   grid_res[3];
  float grid_min[3], grid_max[3], cell_size[3];
} orc_guide;

/* Per-ray data, optix/optix_device.cu:227-242 (PRD). */
typedef struct {
  float radiance[3];
  float throughput[3];
  float origin[3];
  float direction[3];
  uint64_t seed;
  int depth;
  int done;
  float prev_pdf_bsdf;
  int prev_pdf_valid;
  int px, py, sample, rng_dim;
} orc_prd;

typedef struct {
  int   prim;      /* -1 = miss */
  float t, b1, b2; /* b1,b2 weight v1,v2 (OptiX triangle barycentrics) */
} orc_hit;

/* Render modes.
 * ORC_STRICT   replays the reference literally: every 4-sample launch re-derives
 *              the seed from (frame,pixel), restarts the in-launch sample index
 *              at 0 and resets the Welford m2 (optix_wrapper.cpp:983-1021,
 *              optix_device.cu:936-953,1052).
 * ORC_CONVERGE feeds the global sample index into the jitter/blue-noise hash and
 *              lets the PCG stream and m2 continue across launches, so that spp
 *              samples are spp distinct samples of the same estimator. */
enum { ORC_STRICT = 0, ORC_CONVERGE = 1 };

typedef struct {
  int   width, height;
  int   spp;                 /* total samples per pixel (max_spp) */
  int   max_depth;
  int   frame;               /* value of Params.frame during this render (first call = 1) */
  int   bayer_pattern;
  float noise_threshold;     /* 0 disables adaptive sampling */
  int   min_adaptive_samples;
  int   mode;                /* ORC_STRICT / ORC_CONVERGE */
  int   sample_begin;        /* converge mode: global index of this chunk's first sample */
  int   adaptive_deterministic; /* 1: two-pass 3x3 dilation per launch instead of in-launch atomics */
  int   nthreads;            /* <=0: one thread */
  int   adaptive_round;      /* samples per launch (optix_wrapper.cpp:988-989 uses 4); 0: 4; <0: first launch
                              * -adaptive_round samples, each later launch doubles the pixel's count */
  const orc_guide* guide;    /* NULL: no guiding */
  orc_train_sample* train;   /* optional capture ring (single-threaded renders only) */
  uint32_t  train_capacity;
  uint32_t* train_write_idx;
} orc_render_args;

/* ---- RNG / sampling (optix_device.cu:148-222) ---- */
uint32_t orc_pcg(uint64_t* state);
float    orc_rnd(uint64_t* state);
uint32_t orc_hash_u32(uint32_t x);
void     orc_blue_noise(int x, int y, int s, int frame, float out_uv[2]);
float    orc_sample_tri(float u);
void     orc_olpf_jitter(int x, int y, int s, int frame, float out[2]);
uint64_t orc_pixel_seed(int frame, uint32_t dst);
float    orc_next_rand(orc_prd* prd, int frame);
void     orc_make_onb(const float n[3], float t[3], float b[3]);
void     orc_sample_ggx_vndf(const float u[2], float alpha, const float N[3], const float V[3],
                             float L[3], float H[3], float* pdf);

/* ---- camera (optix_wrapper.cpp:813-833, optix_device.cu:890-900) ---- */
void orc_camera_default(int width, int height, float eye[3], float u[3], float v[3], float w[3]);
void orc_camera_dir(int x, int y, const float jitter[2], int width, int height,
                    const float u[3], const float v[3], const float w[3], float out_dir[3]);
int  orc_bayer_channel_for(int x, int y, int pattern);

/* ---- scene (optix_wrapper.cpp:370-489, 842-845) ---- */
int  orc_scene_num_triangles(void);
int  orc_scene_num_vertices(void);
const float*    orc_scene_vertices(void);  /* 3 floats per vertex */
const uint32_t* orc_scene_indices(void);   /* 3 per triangle */
const float*    orc_scene_normals(void);   /* 3 per triangle */
const orc_material* orc_scene_materials(void);
/* Replace the built-in scene by caller geometry (used to cross-check BVH traversal
 * on random soups).  Pass ntri = 0 to restore the Cornell box. */
void orc_scene_set(const float* vertices, int nvert, const uint32_t* indices, int ntri,
                   const orc_material* materials);

/* ---- ray casting: brute force, watertight, two-sided (stands in for optixTrace) ---- */
orc_hit orc_trace_closest(const float o[3], const float d[3], float tmin, float tmax);
int     orc_trace_any(const float o[3], const float d[3], float tmin, float tmax);

/* ---- shading step: optix_device.cu:400-673 + miss 370-375 ----
 * Applies one closest-hit (or miss) to prd.  shadow_* report the NEE ray that the
 * step traced, if any. */
void orc_closesthit(orc_prd* prd, const orc_hit* hit, int frame, int max_depth,
                    const orc_guide* guide, uint64_t* shadow_rays);

/* ---- guiding (guiding_gpu.cuh:35-118, optix_wrapper.cpp:1168-1253) ---- */
int   orc_guiding_region_id(const orc_guide* G, const float P[3]);
void  orc_vmf_sample(const float mu[3], float kappa, const float u[2], float out[3]);
float orc_vmf_pdf(const float mu[3], float kappa, const float wi[3]);
int   orc_guiding_choose_lobe(const orc_region* R, const orc_lobe* L, float u, int ch);
float orc_guiding_mixture_pdf(const orc_region* R, const orc_lobe* L, const float wi[3]);
void  orc_guiding_build_grid(const orc_region* regions, uint32_t n_regions, const int res_in[3],
                             uint32_t* grid_out, int res_out[3], float gmin[3], float gmax[3], float cell[3]);

/* ---- full render (optix_device.cu:914-1071 driven by optix_wrapper.cpp:978-1027) ----
 * out_rgb: w*h*3, out_bayer: w*h, sample_counts: w*h (may be NULL),
 * ray_counts[2]: radiance rays, shadow rays (may be NULL). */
/* Test instrumentation (no counterpart in the reference): while set, orc_render marks per_pixel[y * width + x] = 1
 * for every pixel one of whose radiance rays starts within 1e-5 of the plane of some triangle (the reference's
 * 1e-3 origin offset puts a path that leaves the glass box through its bottom face exactly on the floor). */
void orc_set_on_plane_flags(unsigned char* per_pixel);
/* Test instrumentation: 0 lets every path pass the roulette of optix_device.cu:660-672 with its throughput unchanged
 * (the random number is still drawn); 1 (default) is the reference's behaviour. */
void orc_set_roulette(int on);
int orc_render(const orc_render_args* a, float* out_rgb, float* out_bayer,
               int* sample_counts, uint64_t ray_counts[2]);

/* ---- CFA / demosaic (src/main.rs:157-173, 190-342, 344-401) ---- */
int  orc_cfa_at(int x, int y, int pattern);  /* 0=R 1=G 2=B */
void orc_demosaic_amaze(const float* raw, int w, int h, int pattern, float* out_rgb);
void orc_demosaic_bilinear(const float* raw, int w, int h, int pattern, float* out_rgb);

#ifdef __cplusplus
}
#endif
#endif
