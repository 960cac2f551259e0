/*
 * pt_oracle.c -- scalar CPU restatement of the reference integrator.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  Not linked into the product.
 *
 * Restates, in plain C99 / fp32:
 *   optix/optix_device.cu      15-17, 81-222, 227-242, 261-332, 370-375, 384-396,
 *                              400-673, 890-1071           (live integrator)
 *   optix/guiding_gpu.cuh      35-118                      (guiding lookup + vMF)
 *   optix/optix_wrapper.cpp    370-489 (scene), 813-845 (camera, light),
 *                              978-1027 (batch loop), 1168-1253 (guide grid)
 *   optix/cuda_kernels.cu      3-25                        (compaction / noise clear)
 * Ray casting (optixTrace into the driver's BVH, no source) is restated from the
 * semantics at the call sites: closest hit, two-sided, open interval (tmin,tmax),
 * barycentrics (b1,b2) for v1,v2 -- implemented as a brute-force loop over all
 * triangles with the watertight test of Woop, Benthin, Wald (JCGT 2013).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see Makefile).
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846f /* CUDART_PI_F */

/* optix_device.cu:15-17 */
#define INV_UINT16 (1.0f / 65536.0f)
#define INV_PLASTIC (1.0f / 1.32471795724474602596f)
#define INV_PLASTIC_SQR (1.0f / (1.32471795724474602596f * 1.32471795724474602596f))

/* ------------------------------------------------------------------ float3 */
/* optix_device.cu:81-146 */
typedef orc_f3 f3;
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
static inline void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* operator/ multiplies by the reciprocal (optix_device.cu:104-107) */
static inline f3 div3(f3 a, float s) { float inv = 1.0f / s; return mk3(a.x * inv, a.y * inv, a.z * inv); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
/* normalize: optix_device.cu:122-125 ; normalize3: 137-139 */
static inline f3 normalize_chk(f3 v) {
  float len = length3(v);
  return (len > 0.f) ? div3(v, len) : mk3(0.f, 0.f, 0.f);
}
static inline f3 normalize3(f3 a) { return div3(a, fmaxf(length3(a), 1e-20f)); }
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* --------------------------------------------------------------------- RNG */
/* optix_device.cu:148-154 */
uint32_t orc_pcg(uint64_t* state) {
  uint64_t oldstate = *state;
  *state = oldstate * 6364136223846793005ULL + 1442695040888963407ULL;
  uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  uint32_t rot = (uint32_t)(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
/* optix_device.cu:156-158 */
float orc_rnd(uint64_t* state) {
  return (float)(orc_pcg(state) >> 8) * (1.0f / (float)0x01000000u);
}
/* optix_device.cu:164-174 */
uint32_t orc_hash_u32(uint32_t x) {
  x ^= x >> 17; x *= 0xed5ad4bbU;
  x ^= x >> 11; x *= 0xac4c1b51U;
  x ^= x >> 15; x *= 0x31848babU;
  x ^= x >> 14;
  return x;
}
/* optix_device.cu:176-185 */
void orc_blue_noise(int x, int y, int s, int frame, float out_uv[2]) {
  uint32_t h = orc_hash_u32((uint32_t)x * 1973u + (uint32_t)y * 9277u + (uint32_t)frame * 26699u);
  float rx = (float)(h & 0xffff) * INV_UINT16;
  float ry = (float)((h >> 16) & 0xffff) * INV_UINT16;
  float fs = (float)s + 0.5f;
  out_uv[0] = fmodf(rx + fs * INV_PLASTIC, 1.0f);
  out_uv[1] = fmodf(ry + fs * INV_PLASTIC_SQR, 1.0f);
}
/* optix_device.cu:187-190 */
float orc_sample_tri(float u) {
  return (u < 0.5f) ? sqrtf(2.0f * u) - 1.0f : 1.0f - sqrtf(2.0f * (1.0f - u));
}
/* optix_device.cu:192-196 */
void orc_olpf_jitter(int x, int y, int s, int frame, float out[2]) {
  float u[2];
  orc_blue_noise(x, y, s, frame, u);
  out[0] = orc_sample_tri(u[0]);
  out[1] = orc_sample_tri(u[1]);
}
/* optix_device.cu:945-948 */
uint64_t orc_pixel_seed(int frame, uint32_t dst) {
  return ((uint64_t)(int64_t)frame * 9781ULL) ^ ((uint64_t)(int64_t)(int)dst * 6271ULL) ^ 0x853c49e6748fea9bULL;
}
/* optix_device.cu:198-209 */
float orc_next_rand(orc_prd* prd, int frame) {
  if (prd->rng_dim < 2) {
    float u[2];
    orc_blue_noise(prd->px, prd->py, prd->sample, frame, u);
    float r = prd->rng_dim == 0 ? u[0] : u[1];
    prd->rng_dim++;
    return r;
  }
  prd->rng_dim++;
  return orc_rnd(&prd->seed);
}
/* optix_device.cu:211-222 */
static void make_onb(f3 n, f3* t, f3* b) {
  if (n.z < -0.9999999f) {
    *t = mk3(0.0f, -1.0f, 0.0f);
    *b = mk3(-1.0f, 0.0f, 0.0f);
  } else {
    const float a = 1.0f / (1.0f + n.z);
    const float naxy = -n.x * n.y * a;
    *t = mk3(1.0f - n.x * n.x * a, naxy, -n.x);
    *b = mk3(naxy, 1.0f - n.y * n.y * a, -n.y);
  }
}
void orc_make_onb(const float n[3], float t[3], float b[3]) {
  f3 T, B;
  make_onb(ld3(n), &T, &B);
  st3(t, T); st3(b, B);
}

/* optix_device.cu:261-278 */
static inline f3 reflect3(f3 i, f3 n) { return sub3(i, scale3(n, 2.0f * dot3(i, n))); }
static int refract3(f3 i, f3 n, float eta, f3* t) {
  float cosi = dot3(neg3(n), i);
  float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
  if (k < 0.0f) return 0;
  *t = add3(scale3(i, eta), scale3(n, eta * cosi - sqrtf(k)));
  return 1;
}
static inline f3 fresnel_schlick(float cosTheta, f3 F0) {
  float p = powf(fmaxf(0.0f, 1.0f - cosTheta), 5.0f);
  return add3(F0, scale3(sub3(mk3(1.0f, 1.0f, 1.0f), F0), p));
}

/* optix_device.cu:285-332 */
static void sample_ggx_vndf(float ux, float uy, float alpha, f3 N, f3 V, f3* L, f3* H, float* pdf) {
  f3 T, B;
  make_onb(N, &T, &B);
  f3 Vlocal = mk3(dot3(V, T), dot3(V, B), dot3(V, N));
  f3 Vh = normalize_chk(mk3(alpha * Vlocal.x, alpha * Vlocal.y, Vlocal.z));
  float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  f3 T1 = lensq > 0.0f ? div3(mk3(-Vh.y, Vh.x, 0.0f), sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
  f3 T2 = cross3(Vh, T1);
  float r = sqrtf(ux);
  float phi = 2.0f * ORC_PI * uy;
  float t1 = r * cosf(phi);
  float t2 = r * sinf(phi);
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
  f3 Nh = add3(add3(scale3(T1, t1), scale3(T2, t2)),
               scale3(Vh, sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2))));
  f3 Hlocal = normalize_chk(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
  *H = normalize_chk(add3(add3(scale3(T, Hlocal.x), scale3(B, Hlocal.y)), scale3(N, Hlocal.z)));
  *L = normalize_chk(reflect3(neg3(V), *H));
  float NoH = fmaxf(dot3(N, *H), 0.0f);
  float NoV = fmaxf(dot3(N, V), 0.0f);
  float VoH = fmaxf(dot3(V, *H), 0.0f);
  float alpha2 = alpha * alpha;
  float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
  float D = alpha2 / (ORC_PI * denom * denom);
  float G1V = 2.0f * NoV / (NoV + sqrtf(alpha2 + (1.0f - alpha2) * NoV * NoV));
  *pdf = (D * NoH * G1V) / (4.0f * fmaxf(VoH, 1e-6f));
}
void orc_sample_ggx_vndf(const float u[2], float alpha, const float N[3], const float V[3],
                         float L[3], float H[3], float* pdf) {
  f3 l, h;
  sample_ggx_vndf(u[0], u[1], alpha, ld3(N), ld3(V), &l, &h, pdf);
  st3(L, l); st3(H, h);
}

/* ------------------------------------------------------------------ camera */
/* host normalize3 of optix_wrapper.cpp:132-138 */
static f3 host_normalize3(f3 v) {
  float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  if (len <= 0.f) return mk3(0.f, 0.f, 0.f);
  float inv = 1.0f / len;
  return mk3(v.x * inv, v.y * inv, v.z * inv);
}
/* optix_wrapper.cpp:813-833 (identical to 949-976 with the ensure_handle arguments 1094-1098) */
void orc_camera_default(int width, int height, float eye_o[3], float u_o[3], float v_o[3], float w_o[3]) {
  f3 eye = mk3(0.0f, 3.0f, 1.0f), lookat = mk3(0.0f, 0.0f, 1.0f), up = mk3(0.0f, 0.0f, 1.0f);
  float fovY = 45.0f * (float)M_PI / 180.0f;
  f3 Wv = host_normalize3(sub3(lookat, eye));
  f3 Uv = host_normalize3(cross3(Wv, up));
  f3 Vv = cross3(Uv, Wv);
  float aspect = (height == 0) ? 1.f : (float)width / (float)height;
  float th = tanf(fovY * 0.5f);
  st3(eye_o, eye);
  st3(w_o, Wv);
  st3(u_o, mk3(th * aspect * Uv.x, th * aspect * Uv.y, th * aspect * Uv.z));
  st3(v_o, mk3(th * Vv.x, th * Vv.y, th * Vv.z));
}
/* optix_device.cu:890-900 */
void orc_camera_dir(int x, int y, const float jitter[2], int width, int height,
                    const float u[3], const float v[3], const float w[3], float out_dir[3]) {
  float fx = ((float)x + 0.5f + jitter[0]) / (float)width;
  float fy = ((float)y + 0.5f + jitter[1]) / (float)height;
  fx = clamp01(fx);
  fy = clamp01(fy);
  float dx = 2.0f * fx - 1.0f, dy = 1.0f - 2.0f * fy;
  f3 d = add3(add3(ld3(w), scale3(ld3(u), dx)), scale3(ld3(v), dy));
  st3(out_dir, normalize3(d));
}
/* optix_device.cu:902-912 */
int orc_bayer_channel_for(int x, int y, int pattern) {
  const int mx = x & 1, my = y & 1;
  switch (pattern) {
    case 0: return (my == 0) ? (mx == 0 ? 0 : 1) : (mx == 0 ? 1 : 2);
    case 1: return (my == 0) ? (mx == 0 ? 2 : 1) : (mx == 0 ? 1 : 0);
    case 2: return (my == 0) ? (mx == 0 ? 1 : 0) : (mx == 0 ? 2 : 1);
    default: return (my == 0) ? (mx == 0 ? 1 : 2) : (mx == 0 ? 0 : 1);
  }
}

/* ------------------------------------------------------------------- scene */
static float*        g_vertices;  /* 3 per vertex */
static uint32_t*     g_indices;   /* 3 per tri */
static float*        g_normals;   /* 3 per tri */
static orc_material* g_materials; /* per tri */
static int g_nvert, g_ntri;
static int g_scene_is_builtin;
/* optix_wrapper.cpp:842-845 */
static const f3 g_light_pos = {0.f, 0.f, 1.99f};
static const f3 g_light_emit = {15.f, 15.f, 15.f};
static const f3 g_light_normal = {0.f, 0.f, -1.f};
static const float g_light_half[2] = {0.3f, 0.3f};

static void scene_free(void) {
  free(g_vertices); free(g_indices); free(g_normals); free(g_materials);
  g_vertices = NULL; g_indices = NULL; g_normals = NULL; g_materials = NULL;
  g_nvert = g_ntri = 0;
}
static void scene_compute_normals(void) {
  /* optix_wrapper.cpp:466-476 */
  for (int i = 0; i < g_ntri; ++i) {
    f3 a = ld3(g_vertices + 3 * g_indices[3 * i + 0]);
    f3 b = ld3(g_vertices + 3 * g_indices[3 * i + 1]);
    f3 c = ld3(g_vertices + 3 * g_indices[3 * i + 2]);
    st3(g_normals + 3 * i, host_normalize3(cross3(sub3(b, a), sub3(c, a))));
  }
}
/* optix_wrapper.cpp:370-377 */
static void add_quad(int* nv, int* nt, f3 a, f3 b, f3 c, f3 d) {
  uint32_t base = (uint32_t)*nv;
  st3(g_vertices + 3 * (*nv)++, a); st3(g_vertices + 3 * (*nv)++, b);
  st3(g_vertices + 3 * (*nv)++, c); st3(g_vertices + 3 * (*nv)++, d);
  uint32_t* i0 = g_indices + 3 * (*nt)++;
  i0[0] = base; i0[1] = base + 1; i0[2] = base + 2;
  uint32_t* i1 = g_indices + 3 * (*nt)++;
  i1[0] = base; i1[1] = base + 2; i1[2] = base + 3;
}
static void push_material(int* nm, f3 base, float rough, float transmission, f3 emit) {
  /* optix_wrapper.cpp:392-400: one entry per triangle of the quad */
  for (int k = 0; k < 2; ++k) {
    orc_material* m = &g_materials[(*nm)++];
    st3(m->base, base); m->metallic = 0.f; m->roughness = rough; m->ior = 1.5f;
    m->alpha = 1.f; m->transmission = transmission; st3(m->emission, emit);
  }
}
/* optix_wrapper.cpp:379-489 */
static void scene_build_cornell(void) {
  scene_free();
  g_vertices = (float*)calloc(48 * 3, sizeof(float));
  g_indices = (uint32_t*)calloc(24 * 3, sizeof(uint32_t));
  g_normals = (float*)calloc(24 * 3, sizeof(float));
  g_materials = (orc_material*)calloc(24, sizeof(orc_material));
  int nv = 0, nt = 0, nm = 0;
  const f3 red = mk3(0.49f, 0.056f, 0.016f), green = mk3(0.272f, 0.733f, 0.088f);
  const f3 white = mk3(0.789f, 0.801f, 0.800f), black = mk3(0, 0, 0), emit = mk3(15.f, 15.f, 15.f);
  const f3 A = mk3(-1, -1, 0), B = mk3(1, -1, 0), C = mk3(1, 1, 0), D = mk3(-1, 1, 0);
  const f3 Au = mk3(-1, -1, 2), Bu = mk3(1, -1, 2), Cu = mk3(1, 1, 2), Du = mk3(-1, 1, 2);
  add_quad(&nv, &nt, A, B, C, D);     push_material(&nm, white, 0.25f, 0.f, black); /* floor */
  add_quad(&nv, &nt, Du, Cu, Bu, Au); push_material(&nm, white, 0.25f, 0.f, black); /* ceiling */
  add_quad(&nv, &nt, A, Au, Bu, B);   push_material(&nm, white, 0.25f, 0.f, black); /* back */
  add_quad(&nv, &nt, D, Du, Au, A);   push_material(&nm, green, 0.25f, 0.f, black); /* x=-1 */
  add_quad(&nv, &nt, B, Bu, Cu, C);   push_material(&nm, red, 0.25f, 0.f, black);   /* x=+1 */
  const f3 L0 = mk3(-0.3f, -0.3f, 1.999f), L1 = mk3(0.3f, -0.3f, 1.999f);
  const f3 L2 = mk3(0.3f, 0.3f, 1.999f), L3 = mk3(-0.3f, 0.3f, 1.999f);
  add_quad(&nv, &nt, L3, L2, L1, L0); push_material(&nm, white, 0.25f, 0.f, emit);
  {
    float z0 = 0.001f, z1 = 0.6f;
    f3 p0 = mk3(0.076f, -0.5f, z0), p1 = mk3(0.5f, -0.924f, z0), p2 = mk3(0.924f, -0.5f, z0), p3 = mk3(0.5f, -0.076f, z0);
    f3 q0 = mk3(0.076f, -0.5f, z1), q1 = mk3(0.5f, -0.924f, z1), q2 = mk3(0.924f, -0.5f, z1), q3 = mk3(0.5f, -0.076f, z1);
    add_quad(&nv, &nt, p3, p2, p1, p0); push_material(&nm, white, 0.f, 1.f, black);
    add_quad(&nv, &nt, q0, q1, q2, q3); push_material(&nm, white, 0.f, 1.f, black);
    add_quad(&nv, &nt, p0, p1, q1, q0); push_material(&nm, white, 0.f, 1.f, black);
    add_quad(&nv, &nt, p1, p2, q2, q1); push_material(&nm, white, 0.f, 1.f, black);
    add_quad(&nv, &nt, p2, p3, q3, q2); push_material(&nm, white, 0.f, 1.f, black);
    add_quad(&nv, &nt, p3, p0, q0, q3); push_material(&nm, white, 0.f, 1.f, black);
  }
  g_nvert = nv; g_ntri = nt;
  scene_compute_normals();
  g_scene_is_builtin = 1;
}
static pthread_once_t g_scene_once = PTHREAD_ONCE_INIT;
static void scene_init_once(void) { scene_build_cornell(); }
static void scene_ensure(void) { pthread_once(&g_scene_once, scene_init_once); }

int orc_scene_num_triangles(void) { scene_ensure(); return g_ntri; }
int orc_scene_num_vertices(void) { scene_ensure(); return g_nvert; }
const float* orc_scene_vertices(void) { scene_ensure(); return g_vertices; }
const uint32_t* orc_scene_indices(void) { scene_ensure(); return g_indices; }
const float* orc_scene_normals(void) { scene_ensure(); return g_normals; }
const orc_material* orc_scene_materials(void) { scene_ensure(); return g_materials; }

void orc_scene_set(const float* vertices, int nvert, const uint32_t* indices, int ntri,
                   const orc_material* materials) {
  scene_ensure();
  if (ntri <= 0) { scene_build_cornell(); return; }
  scene_free();
  g_vertices = (float*)malloc(sizeof(float) * 3 * (size_t)nvert);
  g_indices = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)ntri);
  g_normals = (float*)malloc(sizeof(float) * 3 * (size_t)ntri);
  g_materials = (orc_material*)calloc((size_t)ntri, sizeof(orc_material));
  memcpy(g_vertices, vertices, sizeof(float) * 3 * (size_t)nvert);
  memcpy(g_indices, indices, sizeof(uint32_t) * 3 * (size_t)ntri);
  if (materials) memcpy(g_materials, materials, sizeof(orc_material) * (size_t)ntri);
  g_nvert = nvert; g_ntri = ntri;
  scene_compute_normals();
  g_scene_is_builtin = 0;
}

/* -------------------------------------------------------------- ray casting */
/* Watertight ray/triangle test (Woop, Benthin, Wald 2013), two-sided, with the
 * double-precision fallback when an edge function is exactly zero.  Stands in for
 * the driver's triangle test behind optixTrace (optix_device.cu:508-520, 976-986). */
typedef struct { int kx, ky, kz; float Sx, Sy, Sz; } ray_shear;

static ray_shear shear_of(const float d[3]) {
  ray_shear r;
  float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
  r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  r.kx = r.kz + 1; if (r.kx == 3) r.kx = 0;
  r.ky = r.kx + 1; if (r.ky == 3) r.ky = 0;
  if (d[r.kz] < 0.0f) { int t = r.kx; r.kx = r.ky; r.ky = t; }
  r.Sx = d[r.kx] / d[r.kz];
  r.Sy = d[r.ky] / d[r.kz];
  r.Sz = 1.0f / d[r.kz];
  return r;
}

/* returns 1 when the triangle is hit at (tmin, tlimit) exclusive */
static int tri_hit(const ray_shear* rs, const float o[3], const float* v0, const float* v1, const float* v2,
                   float tmin, float tlimit, float* t_out, float* b1_out, float* b2_out) {
  float A[3] = {v0[0] - o[0], v0[1] - o[1], v0[2] - o[2]};
  float B[3] = {v1[0] - o[0], v1[1] - o[1], v1[2] - o[2]};
  float C[3] = {v2[0] - o[0], v2[1] - o[1], v2[2] - o[2]};
  const float Ax = A[rs->kx] - rs->Sx * A[rs->kz], Ay = A[rs->ky] - rs->Sy * A[rs->kz];
  const float Bx = B[rs->kx] - rs->Sx * B[rs->kz], By = B[rs->ky] - rs->Sy * B[rs->kz];
  const float Cx = C[rs->kx] - rs->Sx * C[rs->kz], Cy = C[rs->ky] - rs->Sy * C[rs->kz];
  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
    U = (float)(CxBy - CyBx);
    double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
    V = (float)(AxCy - AyCx);
    double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
    W = (float)(BxAy - ByAx);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
  float det = U + V + W;
  if (det == 0.0f) return 0;
  const float Az = rs->Sz * A[rs->kz], Bz = rs->Sz * B[rs->kz], Cz = rs->Sz * C[rs->kz];
  float T = U * Az + V * Bz + W * Cz;
  if (det < 0.0f) { T = -T; det = -det; }
  if (!(T > tmin * det) || !(T < tlimit * det)) return 0;
  const float rcp = 1.0f / det;
  *t_out = T * rcp;
  /* U,V,W carry the sign of the original det; |.|/|det| gives the weights */
  *b1_out = fabsf(V) * rcp;
  *b2_out = fabsf(W) * rcp;
  return 1;
}

orc_hit orc_trace_closest(const float o[3], const float d[3], float tmin, float tmax) {
  scene_ensure();
  orc_hit h = {-1, tmax, 0.f, 0.f};
  ray_shear rs = shear_of(d);
  for (int i = 0; i < g_ntri; ++i) {
    const uint32_t* tri = g_indices + 3 * i;
    float t, b1, b2;
    if (tri_hit(&rs, o, g_vertices + 3 * tri[0], g_vertices + 3 * tri[1], g_vertices + 3 * tri[2],
                tmin, h.t, &t, &b1, &b2)) {
      h.prim = i; h.t = t; h.b1 = b1; h.b2 = b2;
    }
  }
  return h;
}
int orc_trace_any(const float o[3], const float d[3], float tmin, float tmax) {
  scene_ensure();
  ray_shear rs = shear_of(d);
  for (int i = 0; i < g_ntri; ++i) {
    const uint32_t* tri = g_indices + 3 * i;
    float t, b1, b2;
    if (tri_hit(&rs, o, g_vertices + 3 * tri[0], g_vertices + 3 * tri[1], g_vertices + 3 * tri[2],
                tmin, tmax, &t, &b1, &b2))
      return 1;
  }
  return 0;
}

/* ----------------------------------------------------------------- guiding */
/* guiding_gpu.cuh:35-45 */
int orc_guiding_region_id(const orc_guide* G, const float P[3]) {
  float rx = (P[0] - G->grid_min[0]) / G->cell_size[0];
  float ry = (P[1] - G->grid_min[1]) / G->cell_size[1];
  float rz = (P[2] - G->grid_min[2]) / G->cell_size[2];
  int cx = (int)rx, cy = (int)ry, cz = (int)rz;
  if (cx > G->grid_res[0] - 1) cx = G->grid_res[0] - 1;
  if (cx < 0) cx = 0;
  if (cy > G->grid_res[1] - 1) cy = G->grid_res[1] - 1;
  if (cy < 0) cy = 0;
  if (cz > G->grid_res[2] - 1) cz = G->grid_res[2] - 1;
  if (cz < 0) cz = 0;
  uint32_t id = G->grid[(cz * G->grid_res[1] + cy) * G->grid_res[0] + cx];
  return id == 0xFFFFFFFFu ? -1 : (int)id;
}
/* guiding_gpu.cuh:55-58 */
static f3 g_normalize3(f3 v) {
  float len = sqrtf(dot3(v, v));
  return len > 0.f ? scale3(v, 1.f / len) : mk3(0.f, 0.f, 0.f);
}
/* guiding_gpu.cuh:60-78 */
static f3 vmf_sample(f3 mu, float kappa, float ux, float uy) {
  float cos_theta;
  if (kappa <= 1e-4f) {
    cos_theta = 1.f - 2.f * ux;
  } else {
    float term = expf(-2.f * kappa);
    cos_theta = 1.f + logf(term + (1.f - term) * ux) / kappa;
  }
  float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
  float phi = 2.f * ORC_PI * uy;
  f3 local = mk3(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
  f3 w = g_normalize3(mu);
  f3 up = fabsf(w.z) < 0.9999999f ? mk3(0.f, 0.f, 1.f) : mk3(1.f, 0.f, 0.f);
  f3 u_axis = g_normalize3(cross3(up, w));
  f3 v_axis = cross3(w, u_axis);
  return mk3(u_axis.x * local.x + v_axis.x * local.y + w.x * local.z,
             u_axis.y * local.x + v_axis.y * local.y + w.y * local.z,
             u_axis.z * local.x + v_axis.z * local.y + w.z * local.z);
}
void orc_vmf_sample(const float mu[3], float kappa, const float u[2], float out[3]) {
  st3(out, vmf_sample(ld3(mu), kappa, u[0], u[1]));
}
/* guiding_gpu.cuh:80-84 */
static float vmf_pdf(f3 mu, float kappa, f3 wi) {
  if (kappa <= 1e-4f) return 1.f / (4.f * ORC_PI);
  float norm = kappa / (4.f * ORC_PI * sinhf(kappa));
  return norm * expf(kappa * dot3(mu, wi));
}
float orc_vmf_pdf(const float mu[3], float kappa, const float wi[3]) { return vmf_pdf(ld3(mu), kappa, ld3(wi)); }
/* guiding_gpu.cuh:86-106 */
int orc_guiding_choose_lobe(const orc_region* R, const orc_lobe* L, float u, int ch) {
  float sum = 0.f;
  for (uint32_t i = 0; i < R->lobe_num; ++i) {
    float w = L[i].weight;
    if (ch == 0) w *= L[i].rgb[0]; else if (ch == 1) w *= L[i].rgb[1]; else if (ch == 2) w *= L[i].rgb[2];
    sum += w;
  }
  if (sum <= 0.f) return -1;
  float r = u * sum, acc = 0.f;
  for (uint32_t i = 0; i < R->lobe_num; ++i) {
    float w = L[i].weight;
    if (ch == 0) w *= L[i].rgb[0]; else if (ch == 1) w *= L[i].rgb[1]; else if (ch == 2) w *= L[i].rgb[2];
    acc += w;
    if (r <= acc) return (int)i;
  }
  return (int)(R->lobe_num - 1);
}
/* guiding_gpu.cuh:112-118 */
float orc_guiding_mixture_pdf(const orc_region* R, const orc_lobe* L, const float wi[3]) {
  float sum = 0.f, pdf = 0.f;
  f3 w3 = ld3(wi);
  for (uint32_t i = 0; i < R->lobe_num; ++i) {
    float w = L[i].weight;
    sum += w;
    pdf += w * vmf_pdf(ld3(L[i].mu), L[i].kappa, w3);
  }
  return sum > 0.f ? pdf / sum : 0.f;
}
/* optix_wrapper.cpp:1191-1242 */
void orc_guiding_build_grid(const orc_region* regions, uint32_t n_regions, const int res_in[3],
                            uint32_t* grid_out, int res_out[3], float gmin[3], float gmax[3], float cell[3]) {
  float bmin[3] = {1e30f, 1e30f, 1e30f}, bmax[3] = {-1e30f, -1e30f, -1e30f};
  for (uint32_t i = 0; i < n_regions; ++i)
    for (int k = 0; k < 3; ++k) {
      bmin[k] = fminf(bmin[k], regions[i].bmin[k]);
      bmax[k] = fmaxf(bmax[k], regions[i].bmax[k]);
    }
  int res[3];
  for (int k = 0; k < 3; ++k) res[k] = res_in[k] <= 0 ? 1 : res_in[k];
  for (int k = 0; k < 3; ++k) cell[k] = (bmax[k] - bmin[k]) / (float)res[k];
  size_t n = (size_t)res[0] * (size_t)res[1] * (size_t)res[2];
  for (size_t i = 0; i < n; ++i) grid_out[i] = 0xFFFFFFFFu;
  for (uint32_t r = 0; r < n_regions; ++r) {
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      lo[k] = (int)floorf((regions[r].bmin[k] - bmin[k]) / cell[k]);
      hi[k] = (int)floorf((regions[r].bmax[k] - bmin[k]) / cell[k]);
      if (lo[k] > res[k] - 1) lo[k] = res[k] - 1;
      if (lo[k] < 0) lo[k] = 0;
      if (hi[k] > res[k] - 1) hi[k] = res[k] - 1;
      if (hi[k] < 0) hi[k] = 0;
    }
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int x = lo[0]; x <= hi[0]; ++x) {
          size_t idx = ((size_t)z * (size_t)res[1] + (size_t)y) * (size_t)res[0] + (size_t)x;
          if (grid_out[idx] == 0xFFFFFFFFu) grid_out[idx] = r;
        }
  }
  for (int k = 0; k < 3; ++k) { res_out[k] = res[k]; gmin[k] = bmin[k]; gmax[k] = bmax[k]; }
}

/* --------------------------------------------------------------- closest hit */
typedef struct {
  const orc_guide* guide;
  orc_train_sample* train;
  uint32_t train_capacity;
  uint32_t* train_write_idx;
  int frame, max_depth;
  uint64_t* shadow_rays;
} ch_ctx;

static int g_no_roulette = 0; /* orc_set_roulette(0): instrumentation for the unbiasedness test, never the reference's behaviour */
/* optix_device.cu:660-672 (and the identical block 468-480): depth / Russian roulette.
 * Returns 1 when the caller must return immediately. */
static void depth_and_rr(orc_prd* prd, const ch_ctx* c) {
  prd->depth++;
  if (prd->depth >= c->max_depth) { prd->done = 1; return; }
  if (prd->depth >= c->max_depth - 2) {
    float p = fmaxf(prd->throughput[0], fmaxf(prd->throughput[1], prd->throughput[2]));
    if (g_no_roulette) {  /* test instrumentation: the draw is taken, the path always goes on, nothing is divided */
      (void)orc_next_rand(prd, c->frame);
      return;
    }
    if (orc_next_rand(prd, c->frame) > p) { prd->done = 1; return; }
    st3(prd->throughput, div3(ld3(prd->throughput), fmaxf(p, 1e-3f)));
  }
}

/* optix_device.cu:400-673 */
static void closesthit(orc_prd* prd, const orc_hit* hit, const ch_ctx* c) {
  const int prim = hit->prim;
  const uint32_t* tri = g_indices + 3 * prim;
  const f3 v0 = ld3(g_vertices + 3 * tri[0]), v1 = ld3(g_vertices + 3 * tri[1]), v2 = ld3(g_vertices + 3 * tri[2]);
  const float b1 = hit->b1, b2 = hit->b2, b0 = 1.0f - b1 - b2;
  const f3 P = add3(add3(scale3(v0, b0), scale3(v1, b1)), scale3(v2, b2));
  const f3 Ng = normalize3(ld3(g_normals + 3 * prim));
  const orc_material* m = &g_materials[prim];
  const f3 Base_colour = ld3(m->base);
  const float Roughness = m->roughness, IOR = m->ior, Transmission = m->transmission;
  const f3 Emission = ld3(m->emission);
  const f3 dir = ld3(prd->direction);

  int is_delta = (Transmission > 0.0f) || (Roughness <= 0.0f);
  if (!is_delta && c->train && c->train_write_idx) {
    /* optix_device.cu:432-441 */
    uint32_t idx = (*c->train_write_idx)++;
    if (c->train_capacity) idx %= c->train_capacity;
    orc_train_sample* s = &c->train[idx];
    st3(s->position, P); st3(s->dir_in, neg3(dir)); st3(s->contrib, ld3(prd->throughput)); s->is_delta = 0u;
  }

  if (Transmission > 0.0f) {
    /* optix_device.cu:443-482 */
    f3 N = dot3(dir, Ng) < 0.0f ? Ng : neg3(Ng);
    float eta = dot3(dir, Ng) < 0.0f ? 1.0f / IOR : IOR;
    f3 I = dir;
    float cosi = fmaxf(dot3(neg3(N), I), 0.0f);
    float f0 = (IOR - 1.0f) / (IOR + 1.0f);
    f0 = f0 * f0;
    float Fr = f0 + (1.0f - f0) * powf(1.0f - cosi, 5.0f);
    if (orc_next_rand(prd, c->frame) < Fr) {
      st3(prd->origin, add3(P, scale3(N, 1e-3f)));
      st3(prd->direction, reflect3(I, N));
    } else {
      f3 T;
      if (!refract3(I, N, eta, &T)) {
        st3(prd->origin, add3(P, scale3(N, 1e-3f)));
        st3(prd->direction, reflect3(I, N));
      } else {
        st3(prd->origin, sub3(P, scale3(N, 1e-3f)));
        st3(prd->direction, normalize3(T));
        st3(prd->throughput, mul3(ld3(prd->throughput), Base_colour));
      }
    }
    st3(prd->radiance, add3(ld3(prd->radiance), Emission));
    prd->prev_pdf_valid = 0;
    prd->prev_pdf_bsdf = 0.0f;
    depth_and_rr(prd, c);
    return;
  }

  /* optix_device.cu:484-492 */
  float f0 = (IOR - 1.0f) / (IOR + 1.0f);
  f0 = f0 * f0;
  f3 F0 = mk3(f0, f0, f0);
  f3 V = neg3(dir);
  float NoV = fmaxf(dot3(Ng, V), 0.0f);
  f3 F = fresnel_schlick(NoV, F0);
  float spec_prob = fminf(0.99f, fmaxf(0.01f, (F.x + F.y + F.z) * (1.0f / 3.0f)));
  float diff_prob = 1.0f - spec_prob;
  f3 kd = mul3(Base_colour, sub3(mk3(1.0f, 1.0f, 1.0f), F));

  /* optix_device.cu:495-548: next-event estimation on the rectangle light */
  f3 Lo = mk3(0.0f, 0.0f, 0.0f);
  {
    float u = (orc_next_rand(prd, c->frame) * 2.0f - 1.0f) * g_light_half[0];
    float v = (orc_next_rand(prd, c->frame) * 2.0f - 1.0f) * g_light_half[1];
    f3 lp = mk3(g_light_pos.x + u, g_light_pos.y + v, g_light_pos.z);
    f3 L = sub3(lp, P);
    float dist2 = fmaxf(dot3(L, L), 1e-6f);
    float dist = sqrtf(dist2);
    f3 wi = div3(L, dist);
    float cosS = fmaxf(0.0f, dot3(Ng, wi));
    float cosL = fmaxf(0.0f, dot3(g_light_normal, neg3(wi)));
    if (cosS > 0.0f && cosL > 0.0f) {
      f3 so = add3(P, scale3(Ng, 1e-3f));
      float so_a[3], wi_a[3];
      st3(so_a, so); st3(wi_a, wi);
      if (c->shadow_rays) (*c->shadow_rays)++;
      int occluded = orc_trace_any(so_a, wi_a, 0.0f, dist - 1e-3f);
      if (!occluded) {
        f3 H = normalize3(add3(V, wi));
        float VoH = fmaxf(dot3(V, H), 0.0f);
        float NoH = fmaxf(dot3(Ng, H), 0.0f);
        float alpha = fmaxf(Roughness * Roughness, 1e-4f);
        float alpha2 = alpha * alpha;
        float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
        float D = alpha2 / (ORC_PI * denom * denom);
        float G1V = 2.0f * NoV / (NoV + sqrtf(alpha2 + (1.0f - alpha2) * NoV * NoV));
        float G1L = 2.0f * cosS / (cosS + sqrtf(alpha2 + (1.0f - alpha2) * cosS * cosS));
        f3 Fh = fresnel_schlick(VoH, F0);
        f3 f_spec = scale3(Fh, D * G1V * G1L / fmaxf(4.0f * NoV * cosS, 1e-6f));
        float pdf_spec = D * NoH * G1V / fmaxf(4.0f * VoH, 1e-6f) * spec_prob;
        float pdf_diff = cosS * (1.0f / ORC_PI) * diff_prob;
        float pdf_bsdf = pdf_spec + pdf_diff;
        float area = 4.0f * g_light_half[0] * g_light_half[1];
        float pdf_area = 1.0f / area;
        float pdf_light = pdf_area * dist2 / cosL;
        float w = pdf_light / (pdf_light + pdf_bsdf);
        f3 f = add3(scale3(kd, 1.0f / ORC_PI), f_spec);
        f3 contrib = div3(scale3(mul3(g_light_emit, f), cosS * cosL / dist2), pdf_area);
        Lo = add3(Lo, scale3(contrib, w));
      }
    }
  }

  /* optix_device.cu:551-561: emission with MIS against the previous bounce's pdf */
  f3 emission = Emission;
  if (prd->prev_pdf_valid && (Emission.x > 0.0f || Emission.y > 0.0f || Emission.z > 0.0f)) {
    f3 L = sub3(P, ld3(prd->origin));
    float dist2 = fmaxf(dot3(L, L), 1e-6f);
    float area = 4.0f * g_light_half[0] * g_light_half[1];
    float cosL = fmaxf(0.0f, dot3(g_light_normal, neg3(dir)));
    float pdf_light = dist2 / (cosL * area);
    float w_bsdf = prd->prev_pdf_bsdf / (prd->prev_pdf_bsdf + pdf_light);
    emission = scale3(emission, w_bsdf);
  }
  st3(prd->radiance, add3(ld3(prd->radiance), add3(emission, Lo)));

  float choose = orc_next_rand(prd, c->frame);
  if (choose < spec_prob) {
    /* optix_device.cu:564-589; make_float2 arguments evaluated left to right */
    float ux = orc_next_rand(prd, c->frame);
    float uy = orc_next_rand(prd, c->frame);
    f3 newDir, H;
    float pdf;
    float alpha = fmaxf(Roughness * Roughness, 1e-4f);
    sample_ggx_vndf(ux, uy, alpha, Ng, V, &newDir, &H, &pdf);
    float NoL = fmaxf(dot3(Ng, newDir), 0.0f);
    float NoV2 = fmaxf(dot3(Ng, V), 0.0f);
    float VoH = fmaxf(dot3(V, H), 0.0f);
    float NoH = fmaxf(dot3(Ng, H), 0.0f);
    float alpha2 = alpha * alpha;
    float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
    float D = alpha2 / (ORC_PI * denom * denom);
    float G1V = 2.0f * NoV2 / (NoV2 + sqrtf(alpha2 + (1.0f - alpha2) * NoV2 * NoV2));
    float G1L = 2.0f * NoL / (NoL + sqrtf(alpha2 + (1.0f - alpha2) * NoL * NoL));
    f3 Fh = fresnel_schlick(VoH, F0);
    f3 spec_brdf = scale3(Fh, D * G1V * G1L / fmaxf(4.0f * NoV2 * NoL, 1e-6f));
    st3(prd->origin, add3(P, scale3(Ng, 1e-3f)));
    st3(prd->direction, newDir);
    st3(prd->throughput, mul3(ld3(prd->throughput), div3(scale3(spec_brdf, NoL), fmaxf(pdf * spec_prob, 1e-6f))));
    prd->prev_pdf_bsdf = pdf * spec_prob;
    prd->prev_pdf_valid = 1;
  } else {
    /* optix_device.cu:590-658 */
    f3 newDir = mk3(0, 0, 0);
    float cosTheta = 0.f, p_guided = 0.f, p_bsdf = 0.f;
    f3 lobe_rgb = mk3(1.0f, 1.0f, 1.0f);
    const orc_guide* G = c->guide;
    const orc_lobe* lobes = NULL;
    orc_region R;
    memset(&R, 0, sizeof(R));
    int have_guiding = 0;
    if (G && G->regions && G->lobes && G->grid) {
      float Pa[3];
      st3(Pa, P);
      int region_id = orc_guiding_region_id(G, Pa);
      if (region_id >= 0) {
        R = G->regions[region_id];
        if (R.lobe_num > 0) { lobes = G->lobes + R.lobe_ofs; have_guiding = 1; }
      }
    }
    int used_guiding = 0;
    if (have_guiding) {
      int lobe_idx = orc_guiding_choose_lobe(&R, lobes, orc_next_rand(prd, c->frame), -1);
      if (lobe_idx >= 0) {
        float ux = orc_next_rand(prd, c->frame);
        float uy = orc_next_rand(prd, c->frame);
        f3 gdir = vmf_sample(ld3(lobes[lobe_idx].mu), lobes[lobe_idx].kappa, ux, uy);
        float cosG = fmaxf(dot3(gdir, Ng), 0.0f);
        float ga[3];
        st3(ga, gdir);
        float p_g = orc_guiding_mixture_pdf(&R, lobes, ga);
        float p_b = cosG * (1.0f / ORC_PI);
        float ch2 = orc_next_rand(prd, c->frame) * (p_g + p_b);
        if (ch2 < p_g) {
          used_guiding = 1;
          newDir = gdir; cosTheta = cosG; p_guided = p_g; p_bsdf = p_b;
          lobe_rgb = ld3(lobes[lobe_idx].rgb);
        }
      }
    }
    if (!used_guiding) {
      float r1 = orc_next_rand(prd, c->frame);
      float r2 = orc_next_rand(prd, c->frame);
      const float phi = 2.0f * ORC_PI * r1;
      cosTheta = sqrtf(1.0f - r2);
      const float sinTheta = sqrtf(r2);
      f3 localDir = mk3(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta);
      f3 tangent, bitangent;
      make_onb(Ng, &tangent, &bitangent);
      newDir = normalize_chk(add3(add3(scale3(tangent, localDir.x), scale3(bitangent, localDir.y)), scale3(Ng, localDir.z)));
      p_bsdf = cosTheta * (1.0f / ORC_PI);
      if (have_guiding) {
        float na[3];
        st3(na, newDir);
        p_guided = orc_guiding_mixture_pdf(&R, lobes, na);
      }
    }
    st3(prd->origin, add3(P, scale3(Ng, 1e-3f)));
    st3(prd->direction, newDir);
    float final_pdf = used_guiding ? p_guided : p_bsdf;
    f3 T = mul3(ld3(prd->throughput), kd);
    if (used_guiding) T = mul3(T, lobe_rgb);
    T = scale3(T, cosTheta / (ORC_PI * final_pdf * diff_prob));
    st3(prd->throughput, T);
    prd->prev_pdf_bsdf = final_pdf * diff_prob;
    prd->prev_pdf_valid = 1;
  }
  depth_and_rr(prd, c);
}

void orc_closesthit(orc_prd* prd, const orc_hit* hit, int frame, int max_depth,
                    const orc_guide* guide, uint64_t* shadow_rays) {
  scene_ensure();
  if (hit->prim < 0) {
    /* optix_device.cu:370-375 */
    prd->radiance[0] = prd->radiance[1] = prd->radiance[2] = 0.0f;
    prd->done = 1;
    return;
  }
  ch_ctx c = {guide, NULL, 0, NULL, frame, max_depth, shadow_rays};
  closesthit(prd, hit, &c);
}

/* Test instrumentation, no counterpart in the reference: marks the pixels one of whose rays starts (to
 * within 1e-5) IN the plane of some triangle.  The reference pushes every new origin 1e-3 off the surface it
 * leaves (optix_device.cu:455-466, 577, 640), which puts a path that leaves the glass box through its bottom
 * face (z = 0.001) exactly on the floor; whether such a ray sees that plane is rounding noise in any
 * implementation, and tests use the marks to tell these pixels from real disagreements. */
static unsigned char* g_on_plane_flags = NULL;
void orc_set_on_plane_flags(unsigned char* per_pixel) { g_on_plane_flags = per_pixel; }
void orc_set_roulette(int on) { g_no_roulette = on ? 0 : 1; }
static int origin_on_a_plane(const float o[3]) {
  for (int t = 0; t < g_ntri; ++t) {
    const f3 n = normalize3(ld3(g_normals + 3 * t));
    const f3 v0 = ld3(g_vertices + 3 * g_indices[3 * t]);
    const f3 rel = sub3(ld3(o), v0);
    if (fabsf(dot3(n, rel)) >= 1e-5f) continue;
    /* in the plane: also (nearly) inside the triangle?  barycentrics from the dual basis of the edges */
    const f3 e1 = sub3(ld3(g_vertices + 3 * g_indices[3 * t + 1]), v0), e2 = sub3(ld3(g_vertices + 3 * g_indices[3 * t + 2]), v0);
    const float d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), r1 = dot3(rel, e1), r2 = dot3(rel, e2);
    const float det = d11 * d22 - d12 * d12;
    if (!(det > 0.0f)) continue;
    const float b1 = (d22 * r1 - d12 * r2) / det, b2 = (d11 * r2 - d12 * r1) / det;
    if (b1 >= -0.01f && b2 >= -0.01f && b1 + b2 <= 1.01f) return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ raygen */
typedef struct {
  const orc_render_args* a;
  float cam_eye[3], cam_u[3], cam_v[3], cam_w[3];
  float* out_rgb;   /* running mean, float3 per pixel */
  float* out_bayer; /* running mean */
  float* m2_rgb;    /* carried only in converge mode */
  float* m2_bayer;
  uint64_t* seeds;  /* carried only in converge mode */
  int* sample_counts;
  float* noise_map;
  float* own_noise; /* deterministic adaptive: per-pixel noise of this launch, <0 = not evaluated */
  uint32_t* flags;
  const uint32_t* active;
  uint32_t n_active;
  volatile uint32_t next_chunk;
  uint64_t rays[2];
  pthread_mutex_t lock;
} launch_ctx;

static void atomic_maxf_seq(float* addr, float val) { if (!(*addr >= val)) *addr = val; }

static float pixel_noise(const float mean_rgb[3], const float m2_rgb[3], float mean_bayer, float m2_bayer, int samples) {
  /* optix_device.cu:1013-1025 */
  f3 var_rgb = div3(ld3(m2_rgb), (float)(samples - 1));
  f3 w = mk3(0.2126f, 0.7152f, 0.0722f);
  float lum = dot3(ld3(mean_rgb), w);
  float lum_var = dot3(var_rgb, w);
  float stddev_rgb = sqrtf(fmaxf(lum_var, 0.f));
  float noise = stddev_rgb / fmaxf(lum, 1e-6f);
  float var_bayer = m2_bayer / (float)(samples - 1);
  float stddev_bayer = sqrtf(fmaxf(var_bayer, 0.f));
  float noise_bayer = stddev_bayer / fmaxf(mean_bayer, 1e-6f);
  return fmaxf(noise, noise_bayer);
}

/* optix_device.cu:914-1071 for one launch index */
static void raygen(launch_ctx* L, uint32_t dst, uint64_t rays[2]) {
  const orc_render_args* a = L->a;
  const int W = a->width, H = a->height;
  const int x = (int)dst % W, y = (int)dst / W;
  const int converge = (a->mode == ORC_CONVERGE);
  /* optix_wrapper.cpp:988-989 launches 4 samples; adaptive_round > 0 changes the launch size, < 0 makes
   * the first launch -adaptive_round samples and every later one double the pixel's count */
  const int batch = a->adaptive_round > 0 ? a->adaptive_round
                    : (a->adaptive_round < 0 ? (L->sample_counts[dst] == 0 ? -a->adaptive_round : L->sample_counts[dst]) : 4);

  float mean_rgb[3] = {L->out_rgb[3 * dst], L->out_rgb[3 * dst + 1], L->out_rgb[3 * dst + 2]};
  float m2_rgb[3] = {0.f, 0.f, 0.f};
  float mean_bayer = L->out_bayer[dst];
  float m2_bayer = 0.f;
  const int ch = orc_bayer_channel_for(x, y, a->bayer_pattern);
  uint64_t seed = orc_pixel_seed(a->frame, dst);
  int samples = L->sample_counts[dst];
  if (converge) {
    if (samples == 0) {
      if (a->sample_begin != 0) { /* a later sample range: a stream of its own, all 64 bits of its start mixed */
        seed ^= (uint64_t)(int64_t)a->sample_begin * 0x9E3779B97F4A7C15ULL;
        seed ^= seed >> 30; seed *= 0xBF58476D1CE4E5B9ULL;
        seed ^= seed >> 27; seed *= 0x94D049BB133111EBULL;
        seed ^= seed >> 31;
      }
    } else {
      seed = L->seeds[dst];
    }
    m2_rgb[0] = L->m2_rgb[3 * dst]; m2_rgb[1] = L->m2_rgb[3 * dst + 1]; m2_rgb[2] = L->m2_rgb[3 * dst + 2];
    m2_bayer = L->m2_bayer[dst];
  }
  float max_noise = a->noise_threshold;
  ch_ctx c = {a->guide, a->train, a->train_capacity, a->train_write_idx, a->frame, a->max_depth, &rays[1]};

  for (int s = 0; s < batch && samples < a->spp; ++s) {
    const int s_hash = converge ? (a->sample_begin + samples) : s;
    samples++;
    orc_prd prd;
    memset(&prd, 0, sizeof(prd));
    prd.throughput[0] = prd.throughput[1] = prd.throughput[2] = 1.0f;
    memcpy(prd.origin, L->cam_eye, sizeof(float) * 3);
    float jitter[2];
    orc_olpf_jitter(x, y, s_hash, a->frame, jitter);
    orc_camera_dir(x, y, jitter, W, H, L->cam_u, L->cam_v, L->cam_w, prd.direction);
    prd.seed = seed;
    prd.px = x; prd.py = y; prd.sample = s_hash; prd.rng_dim = 0;
    float sample_rgb[3] = {0.f, 0.f, 0.f};
    float sample_bayer = 0.f;

    while (!prd.done) {
      prd.radiance[0] = prd.radiance[1] = prd.radiance[2] = 0.0f;
      rays[0]++;
      if (g_on_plane_flags && origin_on_a_plane(prd.origin)) g_on_plane_flags[dst] = 1;
      orc_hit hit = orc_trace_closest(prd.origin, prd.direction, 0.0f, 1e16f);
      if (hit.prim < 0) {
        prd.radiance[0] = prd.radiance[1] = prd.radiance[2] = 0.0f;
        prd.done = 1;
      } else {
        closesthit(&prd, &hit, &c);
      }
      /* optix_device.cu:988-997: contribution uses the post-bounce throughput */
      for (int k = 0; k < 3; ++k) sample_rgb[k] += prd.radiance[k] * prd.throughput[k];
      sample_bayer += prd.radiance[ch] * prd.throughput[ch];
    }

    /* optix_device.cu:1000-1009 */
    for (int k = 0; k < 3; ++k) {
      float delta = sample_rgb[k] - mean_rgb[k];
      mean_rgb[k] += delta * (1.0f / (float)samples); /* float3 / float multiplies by the reciprocal */
      float delta2 = sample_rgb[k] - mean_rgb[k];
      m2_rgb[k] += delta * delta2;
    }
    {
      float delta = sample_bayer - mean_bayer;
      mean_bayer += delta / (float)samples;
      float delta2 = sample_bayer - mean_bayer;
      m2_bayer += delta * delta2;
    }

    /* optix_device.cu:1011-1050 */
    if (!a->adaptive_deterministic && a->noise_threshold > 0.f && samples >= a->min_adaptive_samples) {
      float noise = pixel_noise(mean_rgb, m2_rgb, mean_bayer, m2_bayer, samples);
      float* nm = L->noise_map;
      atomic_maxf_seq(&nm[dst], noise);
      if (x > 0) atomic_maxf_seq(&nm[dst - 1], noise);
      if (x + 1 < W) atomic_maxf_seq(&nm[dst + 1], noise);
      if (y > 0) atomic_maxf_seq(&nm[dst - W], noise);
      if (y + 1 < H) atomic_maxf_seq(&nm[dst + W], noise);
      if (x > 0 && y > 0) atomic_maxf_seq(&nm[dst - W - 1], noise);
      if (x + 1 < W && y > 0) atomic_maxf_seq(&nm[dst - W + 1], noise);
      if (x > 0 && y + 1 < H) atomic_maxf_seq(&nm[dst + W - 1], noise);
      if (x + 1 < W && y + 1 < H) atomic_maxf_seq(&nm[dst + W + 1], noise);
      max_noise = fmaxf(noise, nm[dst]);
      if (x > 0) max_noise = fmaxf(max_noise, nm[dst - 1]);
      if (x + 1 < W) max_noise = fmaxf(max_noise, nm[dst + 1]);
      if (y > 0) max_noise = fmaxf(max_noise, nm[dst - W]);
      if (y + 1 < H) max_noise = fmaxf(max_noise, nm[dst + W]);
      if (x > 0 && y > 0) max_noise = fmaxf(max_noise, nm[dst - W - 1]);
      if (x + 1 < W && y > 0) max_noise = fmaxf(max_noise, nm[dst - W + 1]);
      if (x > 0 && y + 1 < H) max_noise = fmaxf(max_noise, nm[dst + W - 1]);
      if (x + 1 < W && y + 1 < H) max_noise = fmaxf(max_noise, nm[dst + W + 1]);
      if (max_noise < a->noise_threshold) { seed = prd.seed; break; }
    }
    seed = prd.seed; /* optix_device.cu:1052 */
  }
  L->sample_counts[dst] = samples;

  /* optix_device.cu:1056-1060 */
  uint32_t active = (samples < a->spp) ? 1u : 0u;
  if (!a->adaptive_deterministic && a->noise_threshold > 0.f && samples >= a->min_adaptive_samples)
    active = (max_noise >= a->noise_threshold) && (samples < a->spp);
  L->flags[dst] = active;
  if (a->adaptive_deterministic) {
    L->own_noise[dst] = (a->noise_threshold > 0.f && samples >= a->min_adaptive_samples)
                            ? pixel_noise(mean_rgb, m2_rgb, mean_bayer, m2_bayer, samples) : -1.0f;
  }

  for (int k = 0; k < 3; ++k) L->out_rgb[3 * dst + k] = mean_rgb[k];
  L->out_bayer[dst] = fmaxf(mean_bayer, 0.f); /* optix_device.cu:1069 */
  if (converge) {
    L->seeds[dst] = seed;
    for (int k = 0; k < 3; ++k) L->m2_rgb[3 * dst + k] = m2_rgb[k];
    L->m2_bayer[dst] = m2_bayer;
  }
}

#define ORC_CHUNK 256u
static void* launch_worker(void* p) {
  launch_ctx* L = (launch_ctx*)p;
  uint64_t rays[2] = {0, 0};
  for (;;) {
    uint32_t c = __sync_fetch_and_add(&L->next_chunk, 1u);
    uint32_t lo = c * ORC_CHUNK;
    if (lo >= L->n_active) break;
    uint32_t hi = lo + ORC_CHUNK < L->n_active ? lo + ORC_CHUNK : L->n_active;
    for (uint32_t i = lo; i < hi; ++i) raygen(L, L->active[i], rays);
  }
  pthread_mutex_lock(&L->lock);
  L->rays[0] += rays[0];
  L->rays[1] += rays[1];
  pthread_mutex_unlock(&L->lock);
  return NULL;
}

/* optix_wrapper.cpp:978-1027 (== 1029-1080: the Bayer render launches the same programs) */
int orc_render(const orc_render_args* a, float* out_rgb, float* out_bayer, int* sample_counts_out,
               uint64_t ray_counts[2]) {
  if (!out_rgb || !out_bayer) return -1;
  scene_ensure();
  const size_t px = (size_t)a->width * (size_t)a->height;
  launch_ctx L;
  memset(&L, 0, sizeof(L));
  L.a = a;
  orc_camera_default(a->width, a->height, L.cam_eye, L.cam_u, L.cam_v, L.cam_w);
  L.out_rgb = out_rgb;
  L.out_bayer = out_bayer;
  memset(out_rgb, 0, px * 3 * sizeof(float));
  memset(out_bayer, 0, px * sizeof(float));
  L.m2_rgb = (float*)calloc(px * 3, sizeof(float));
  L.m2_bayer = (float*)calloc(px, sizeof(float));
  L.seeds = (uint64_t*)calloc(px, sizeof(uint64_t));
  L.sample_counts = (int*)calloc(px, sizeof(int));
  L.noise_map = (float*)calloc(px, sizeof(float));
  L.own_noise = (float*)calloc(px, sizeof(float));
  L.flags = (uint32_t*)calloc(px, sizeof(uint32_t));
  uint32_t* active_in = (uint32_t*)malloc(px * sizeof(uint32_t));
  uint32_t* active_out = (uint32_t*)malloc(px * sizeof(uint32_t));
  float* dil = (float*)calloc(px, sizeof(float));
  for (size_t i = 0; i < px; ++i) active_in[i] = (uint32_t)i;
  pthread_mutex_init(&L.lock, NULL);
  int nthreads = a->nthreads > 0 ? a->nthreads : 1;
  if (nthreads > 256) nthreads = 256;
  if (a->train) nthreads = 1;
  const int W = a->width, H = a->height;

  uint32_t active = (uint32_t)px;
  while (active > 0) {
    /* clear_noise: cuda_kernels.cu:18-25 */
    for (uint32_t i = 0; i < active; ++i) L.noise_map[active_in[i]] = 0.f;
    L.active = active_in;
    L.n_active = active;
    L.next_chunk = 0;
    if (nthreads == 1) {
      launch_worker(&L);
    } else {
      pthread_t th[256];
      for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, launch_worker, &L);
      for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    if (a->adaptive_deterministic && a->noise_threshold > 0.f) {
      /* Scheduling-independent form of optix_device.cu:1027-1049: every evaluated pixel
       * raises its 3x3 neighbourhood in the noise map (stale values at inactive pixels
       * persist, as they do in the reference where only active pixels are cleared),
       * then reads the 3x3 maximum. */
      for (uint32_t i = 0; i < active; ++i) {
        uint32_t p = active_in[i];
        float n = L.own_noise[p];
        if (n < 0.f) continue;
        int x = (int)p % W, y = (int)p / W;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            int xx = x + dx, yy = y + dy;
            if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
            atomic_maxf_seq(&L.noise_map[yy * W + xx], n);
          }
      }
      for (uint32_t i = 0; i < active; ++i) {
        uint32_t p = active_in[i];
        float n = L.own_noise[p];
        if (n < 0.f) continue;
        int x = (int)p % W, y = (int)p / W;
        float mx = n;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            int xx = x + dx, yy = y + dy;
            if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
            mx = fmaxf(mx, L.noise_map[yy * W + xx]);
          }
        L.flags[p] = (mx >= a->noise_threshold) && (L.sample_counts[p] < a->spp);
      }
    }
    /* compact_active: cuda_kernels.cu:3-16 (order kept here; the reference's is unordered) */
    uint32_t n_out = 0;
    for (uint32_t i = 0; i < active; ++i)
      if (L.flags[active_in[i]]) active_out[n_out++] = active_in[i];
    active = n_out;
    uint32_t* t = active_in; active_in = active_out; active_out = t;
  }

  if (sample_counts_out) memcpy(sample_counts_out, L.sample_counts, px * sizeof(int));
  if (ray_counts) { ray_counts[0] = L.rays[0]; ray_counts[1] = L.rays[1]; }
  pthread_mutex_destroy(&L.lock);
  free(L.m2_rgb); free(L.m2_bayer); free(L.seeds); free(L.sample_counts); free(L.noise_map);
  free(L.own_noise); free(L.flags); free(active_in); free(active_out); free(dil);
  return 0;
}
