// pt_demosaic.hip -- Bayer demosaic on the GPU (a16), bit-compatible with the reference's
// sequential CPU loops (src/main.rs:190-342 gradient method, 344-401 bilinear).
//
// The reference runs six in-place passes over three colour planes.  In every pass a site
// writes only itself and reads only sites that the pass does not write (or itself, through
// border clamping, before its own write), so each pass is a data-parallel gather.  The shipped
// kernel, k_gradient_tiles, runs them all on one tile in LDS.  The pass-by-pass form it fuses is kept
// for comparison in builds with -DPT_DEMOSAIC_PASSES (the profiling library) only:
//   k_green   raw            -> g     (passes 0+1: plane split folded into the reads)
//   k_rb_at_g raw, g         -> r, b  (pass 2; red/blue sites carry their raw value / 0)
//   k_rb_diag r, g, b        -> r, b  (pass 3, in place)
//   k_combine r, g, b        -> rgb   (passes 4-6: chroma planes formed on the fly)
// Arithmetic is plain fp32 in the reference's operation order; this file is compiled with
// FMA contraction off and correctly rounded division so results match bit for bit.
#include "pt_kernels.h"
#include "pt_demosaic_tile.h"

#include <cstdio>
#include <cstdlib>

namespace pt {

enum { CFA_R = 0, CFA_G = 1, CFA_B = 2 };

// src/main.rs:157-173
__device__ __forceinline__ int cfa_at(int x, int y, int pattern) {
  const int p = pattern < 0 ? 0 : (pattern > 3 ? 3 : pattern);
  // the four 2x2 tables, 2 bits per entry (R=0 G=1 B=2), one byte per pattern:
  // p0 RGGB: [R G; G B]  p1 BGGR: [B G; G R]  p2 GRBG: [G R; B G]  p3 GBRG: [G B; R G]
  constexpr unsigned t0 = 0x0u | (1u << 2) | (1u << 4) | (2u << 6), t1 = 2u | (1u << 2) | (1u << 4) | (0u << 6);
  constexpr unsigned t2 = 1u | (0u << 2) | (2u << 4) | (1u << 6), t3 = 1u | (2u << 2) | (0u << 4) | (1u << 6);
  constexpr unsigned all = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);
  return (int)((all >> (8 * p + 2 * ((y & 1) * 2 + (x & 1)))) & 3u);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// src/main.rs:175-179
__device__ __forceinline__ float getf(const float* img, int w, int h, int x, int y) {
  return img[(size_t)clampi(y, 0, h - 1) * (size_t)w + (size_t)clampi(x, 0, w - 1)];
}
// plane value of colour c at a (clamped) site straight from the mosaic: raw where the CFA
// has colour c, 0 elsewhere (src/main.rs:192-204)
__device__ __forceinline__ float plane_raw(const float* raw, int w, int h, int x, int y, int pattern, int c) {
  const int xx = clampi(x, 0, w - 1), yy = clampi(y, 0, h - 1);
  return cfa_at(xx, yy, pattern) == c ? raw[(size_t)yy * w + xx] : 0.0f;
}

#ifdef PT_DEMOSAIC_PASSES
// src/main.rs:205-237
__global__ void k_green(const float* raw, float* g, int w, int h, int pattern) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int c = cfa_at(x, y, pattern);
  const size_t i = (size_t)y * w + x;
  if (c == CFA_G) { g[i] = raw[i]; return; }
  const float c0 = plane_raw(raw, w, h, x, y, pattern, c);
  const float c_l2 = plane_raw(raw, w, h, x - 2, y, pattern, c), c_l1 = plane_raw(raw, w, h, x - 1, y, pattern, c);
  const float c_r1 = plane_raw(raw, w, h, x + 1, y, pattern, c), c_r2 = plane_raw(raw, w, h, x + 2, y, pattern, c);
  const float c_u2 = plane_raw(raw, w, h, x, y - 2, pattern, c), c_u1 = plane_raw(raw, w, h, x, y - 1, pattern, c);
  const float c_d1 = plane_raw(raw, w, h, x, y + 1, pattern, c), c_d2 = plane_raw(raw, w, h, x, y + 2, pattern, c);
  const float g_l1 = plane_raw(raw, w, h, x - 1, y, pattern, CFA_G), g_r1 = plane_raw(raw, w, h, x + 1, y, pattern, CFA_G);
  const float g_u1 = plane_raw(raw, w, h, x, y - 1, pattern, CFA_G), g_d1 = plane_raw(raw, w, h, x, y + 1, pattern, CFA_G);
  const float grad_h = fabsf(g_l1 - g_r1) + fabsf(2.0f * c0 - c_l2 - c_r2) * 0.5f + fabsf(c_l1 - c_r1);
  const float grad_v = fabsf(g_u1 - g_d1) + fabsf(2.0f * c0 - c_u2 - c_d2) * 0.5f + fabsf(c_u1 - c_d1);
  const float gh = 0.5f * (g_l1 + g_r1) + 0.25f * (2.0f * c0 - c_l2 - c_r2);
  const float gv = 0.5f * (g_u1 + g_d1) + 0.25f * (2.0f * c0 - c_u2 - c_d2);
  const float wh = 1.0f / (1e-6f + grad_h * grad_h);
  const float wv = 1.0f / (1e-6f + grad_v * grad_v);
  g[i] = (wh * gh + wv * gv) / (wh + wv);
}

// src/main.rs:238-276
__global__ void k_rb_at_g(const float* raw, const float* g, float* r, float* b, int w, int h, int pattern) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  const int c = cfa_at(x, y, pattern);
  if (c == CFA_R) { r[i] = raw[i]; b[i] = 0.0f; return; }
  if (c == CFA_B) { b[i] = raw[i]; r[i] = 0.0f; return; }
  const float g0 = g[i];
  const bool has_red_h = cfa_at(x - 1, y, pattern) == CFA_R || cfa_at(x + 1, y, pattern) == CFA_R;
  const bool has_red_v = cfa_at(x, y - 1, pattern) == CFA_R || cfa_at(x, y + 1, pattern) == CFA_R;
  const bool has_blue_h = cfa_at(x - 1, y, pattern) == CFA_B || cfa_at(x + 1, y, pattern) == CFA_B;
  const bool has_blue_v = cfa_at(x, y - 1, pattern) == CFA_B || cfa_at(x, y + 1, pattern) == CFA_B;
  float r_est, b_est;
  if (has_red_h)
    r_est = 0.5f * (plane_raw(raw, w, h, x - 1, y, pattern, CFA_R) + plane_raw(raw, w, h, x + 1, y, pattern, CFA_R)) -
            0.5f * (getf(g, w, h, x - 1, y) + getf(g, w, h, x + 1, y)) + g0;
  else if (has_red_v)
    r_est = 0.5f * (plane_raw(raw, w, h, x, y - 1, pattern, CFA_R) + plane_raw(raw, w, h, x, y + 1, pattern, CFA_R)) -
            0.5f * (getf(g, w, h, x, y - 1) + getf(g, w, h, x, y + 1)) + g0;
  else
    r_est = g0;
  if (has_blue_h)
    b_est = 0.5f * (plane_raw(raw, w, h, x - 1, y, pattern, CFA_B) + plane_raw(raw, w, h, x + 1, y, pattern, CFA_B)) -
            0.5f * (getf(g, w, h, x - 1, y) + getf(g, w, h, x + 1, y)) + g0;
  else if (has_blue_v)
    b_est = 0.5f * (plane_raw(raw, w, h, x, y - 1, pattern, CFA_B) + plane_raw(raw, w, h, x, y + 1, pattern, CFA_B)) -
            0.5f * (getf(g, w, h, x, y - 1) + getf(g, w, h, x, y + 1)) + g0;
  else
    b_est = g0;
  r[i] = r_est;
  b[i] = b_est;
}

// src/main.rs:277-303
__global__ void k_rb_diag(const float* g, float* r, float* b, int w, int h, int pattern) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int c = cfa_at(x, y, pattern);
  if (c == CFA_G) return;
  float* dst = (c == CFA_R) ? b : r;  // the colour missing at this site
  const float g0 = getf(g, w, h, x, y);
  const float est = 0.25f * ((getf(dst, w, h, x - 1, y - 1) - getf(g, w, h, x - 1, y - 1)) +
                             (getf(dst, w, h, x + 1, y - 1) - getf(g, w, h, x + 1, y - 1)) +
                             (getf(dst, w, h, x - 1, y + 1) - getf(g, w, h, x - 1, y + 1)) +
                             (getf(dst, w, h, x + 1, y + 1) - getf(g, w, h, x + 1, y + 1))) + g0;
  dst[(size_t)y * w + x] = est;
}

__device__ __forceinline__ float diff_at(const float* a, const float* g, int w, int h, int x, int y) {
  const int xx = clampi(x, 0, w - 1), yy = clampi(y, 0, h - 1);
  const size_t i = (size_t)yy * w + xx;
  return a[i] - g[i];  // src/main.rs:306-309
}
// src/main.rs:304-341
__global__ void k_combine(const float* r, const float* g, const float* b, float* out, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const float gh = fabsf(getf(g, w, h, x - 1, y) - getf(g, w, h, x + 1, y));
  const float gv = fabsf(getf(g, w, h, x, y - 1) - getf(g, w, h, x, y + 1));
  const float w_h = 1.0f / (1e-6f + gh);
  const float w_v = 1.0f / (1e-6f + gv);
  const float rgh = 0.5f * (diff_at(r, g, w, h, x - 1, y) + diff_at(r, g, w, h, x + 1, y));
  const float rgv = 0.5f * (diff_at(r, g, w, h, x, y - 1) + diff_at(r, g, w, h, x, y + 1));
  const float bgh = 0.5f * (diff_at(b, g, w, h, x - 1, y) + diff_at(b, g, w, h, x + 1, y));
  const float bgv = 0.5f * (diff_at(b, g, w, h, x, y - 1) + diff_at(b, g, w, h, x, y + 1));
  const float rg2 = (w_h * rgh + w_v * rgv) / (w_h + w_v);
  const float bg2 = (w_h * bgh + w_v * bgv) / (w_h + w_v);
  const size_t i = (size_t)y * w + x;
  const float g0 = g[i];
  out[3 * i + 0] = fmaxf(g0 + rg2, 0.0f);
  out[3 * i + 1] = fmaxf(g0, 0.0f);
  out[3 * i + 2] = fmaxf(g0 + bg2, 0.0f);
}

#endif  // PT_DEMOSAIC_PASSES

// ------------------------------------------------------------------ fused gradient method
// The four passes above move about 70 bytes per pixel through HBM (three intermediate planes written
// and read back).  k_gradient_tiles runs them on one tile per workgroup with the planes in LDS: raw is
// read once with an apron, rgb written once (about 20 bytes per pixel).  Every plane value is computed by
// the same expressions in the same order as in the pass kernels, so the result is the same bits.
// Two forms of the tile routine:
//   border_tile    any tile: coordinates are clamped before every LDS lookup, exactly as getf() clamps them, and
//                  the colour of a clamped neighbour is looked up where it was clamped to;
//   interior_tile  a tile whose apron lies inside the image (pt_demosaic_tile.h): nothing is clamped, colours are
//                  compile-time constants of the pattern, a thread works on a 2x2 cell (no lane idles because its
//                  site has the wrong colour), neighbours are read as aligned pairs.
// The grid lists the border tiles first, so that the slower routine runs beside the interior tiles, not after them.
constexpr int kTW = dm::kTileW, kTH = dm::kTileH;  // output tile
constexpr int kHR = 5, kHG = 3, kHS = 2;           // aprons of border_tile: raw, green, red/blue planes
constexpr int kRW = kTW + 2 * kHR, kRH = kTH + 2 * kHR;
constexpr int kGW = kTW + 2 * kHG, kGH = kTH + 2 * kHG;
constexpr int kSW = kTW + 2 * kHS, kSH = kTH + 2 * kHS;
constexpr int kBorderFloats = kRW * kRH + kGW * kGH + 2 * kSW * kSH;
constexpr int kTileFloats = 4 * dm::kPlane > kBorderFloats ? 4 * dm::kPlane : kBorderFloats;

struct Tile {
  int x0, y0, w, h, pattern;
  const float* raw;  // LDS, kRW x kRH, origin (x0 - kHR, y0 - kHR)
  float* g;          // LDS, kGW x kGH, origin (x0 - kHG, y0 - kHG)
  float* r;          // LDS, kSW x kSH, origin (x0 - kHS, y0 - kHS)
  float* b;
  __device__ __forceinline__ float raw_at(int x, int y) const {
    return raw[(clampi(y, 0, h - 1) - (y0 - kHR)) * kRW + (clampi(x, 0, w - 1) - (x0 - kHR))];
  }
  // value of colour plane c taken straight from the mosaic (src/main.rs:192-204)
  __device__ __forceinline__ float plane_at(int x, int y, int c) const {
    const int xx = clampi(x, 0, w - 1), yy = clampi(y, 0, h - 1);
    return cfa_at(xx, yy, pattern) == c ? raw[(yy - (y0 - kHR)) * kRW + (xx - (x0 - kHR))] : 0.0f;
  }
  __device__ __forceinline__ int gi(int x, int y) const {
    return (clampi(y, 0, h - 1) - (y0 - kHG)) * kGW + (clampi(x, 0, w - 1) - (x0 - kHG));
  }
  __device__ __forceinline__ int si(int x, int y) const {
    return (clampi(y, 0, h - 1) - (y0 - kHS)) * kSW + (clampi(x, 0, w - 1) - (x0 - kHS));
  }
};

__device__ __forceinline__ void border_tile(const float* raw, float* out, int w, int h, int pattern, int x0, int y0, float* lds) {
  float* s_raw = lds;
  float* s_g = s_raw + kRW * kRH;
  float* s_r = s_g + kGW * kGH;
  float* s_b = s_r + kSW * kSH;
  Tile t;
  t.x0 = x0; t.y0 = y0; t.w = w; t.h = h; t.pattern = pattern;
  t.raw = s_raw; t.g = s_g; t.r = s_r; t.b = s_b;
  const int tid = threadIdx.x;
  // mosaic with apron (only in-image cells are ever read)
  for (int k = tid; k < kRW * kRH; k += 256) {
    const int x = t.x0 - kHR + k % kRW, y = t.y0 - kHR + k / kRW;
    if (x >= 0 && y >= 0 && x < w && y < h) s_raw[k] = raw[(size_t)y * w + x];
  }
  __syncthreads();
  // green plane (k_green)
  for (int k = tid; k < kGW * kGH; k += 256) {
    const int x = t.x0 - kHG + k % kGW, y = t.y0 - kHG + k / kGW;
    if (x < 0 || y < 0 || x >= w || y >= h) continue;
    const int c = cfa_at(x, y, pattern);
    if (c == CFA_G) { s_g[k] = t.raw_at(x, y); continue; }
    const float c0 = t.plane_at(x, y, c);
    const float c_l2 = t.plane_at(x - 2, y, c), c_l1 = t.plane_at(x - 1, y, c);
    const float c_r1 = t.plane_at(x + 1, y, c), c_r2 = t.plane_at(x + 2, y, c);
    const float c_u2 = t.plane_at(x, y - 2, c), c_u1 = t.plane_at(x, y - 1, c);
    const float c_d1 = t.plane_at(x, y + 1, c), c_d2 = t.plane_at(x, y + 2, c);
    const float g_l1 = t.plane_at(x - 1, y, CFA_G), g_r1 = t.plane_at(x + 1, y, CFA_G);
    const float g_u1 = t.plane_at(x, y - 1, CFA_G), g_d1 = t.plane_at(x, y + 1, CFA_G);
    const float grad_h = fabsf(g_l1 - g_r1) + fabsf(2.0f * c0 - c_l2 - c_r2) * 0.5f + fabsf(c_l1 - c_r1);
    const float grad_v = fabsf(g_u1 - g_d1) + fabsf(2.0f * c0 - c_u2 - c_d2) * 0.5f + fabsf(c_u1 - c_d1);
    const float gh = 0.5f * (g_l1 + g_r1) + 0.25f * (2.0f * c0 - c_l2 - c_r2);
    const float gv = 0.5f * (g_u1 + g_d1) + 0.25f * (2.0f * c0 - c_u2 - c_d2);
    const float wh = 1.0f / (1e-6f + grad_h * grad_h);
    const float wv = 1.0f / (1e-6f + grad_v * grad_v);
    s_g[k] = (wh * gh + wv * gv) / (wh + wv);
  }
  __syncthreads();
  // red and blue at green sites, raw / 0 at red and blue sites (k_rb_at_g)
  for (int k = tid; k < kSW * kSH; k += 256) {
    const int x = t.x0 - kHS + k % kSW, y = t.y0 - kHS + k / kSW;
    if (x < 0 || y < 0 || x >= w || y >= h) continue;
    const int c = cfa_at(x, y, pattern);
    if (c == CFA_R) { s_r[k] = t.raw_at(x, y); s_b[k] = 0.0f; continue; }
    if (c == CFA_B) { s_b[k] = t.raw_at(x, y); s_r[k] = 0.0f; continue; }
    const float g0 = s_g[t.gi(x, y)];
    const bool has_red_h = cfa_at(x - 1, y, pattern) == CFA_R || cfa_at(x + 1, y, pattern) == CFA_R;
    const bool has_red_v = cfa_at(x, y - 1, pattern) == CFA_R || cfa_at(x, y + 1, pattern) == CFA_R;
    const bool has_blue_h = cfa_at(x - 1, y, pattern) == CFA_B || cfa_at(x + 1, y, pattern) == CFA_B;
    const bool has_blue_v = cfa_at(x, y - 1, pattern) == CFA_B || cfa_at(x, y + 1, pattern) == CFA_B;
    float r_est, b_est;
    if (has_red_h)
      r_est = 0.5f * (t.plane_at(x - 1, y, CFA_R) + t.plane_at(x + 1, y, CFA_R)) -
              0.5f * (s_g[t.gi(x - 1, y)] + s_g[t.gi(x + 1, y)]) + g0;
    else if (has_red_v)
      r_est = 0.5f * (t.plane_at(x, y - 1, CFA_R) + t.plane_at(x, y + 1, CFA_R)) -
              0.5f * (s_g[t.gi(x, y - 1)] + s_g[t.gi(x, y + 1)]) + g0;
    else
      r_est = g0;
    if (has_blue_h)
      b_est = 0.5f * (t.plane_at(x - 1, y, CFA_B) + t.plane_at(x + 1, y, CFA_B)) -
              0.5f * (s_g[t.gi(x - 1, y)] + s_g[t.gi(x + 1, y)]) + g0;
    else if (has_blue_v)
      b_est = 0.5f * (t.plane_at(x, y - 1, CFA_B) + t.plane_at(x, y + 1, CFA_B)) -
              0.5f * (s_g[t.gi(x, y - 1)] + s_g[t.gi(x, y + 1)]) + g0;
    else
      b_est = g0;
    s_r[k] = r_est;
    s_b[k] = b_est;
  }
  __syncthreads();
  // the missing colour at red and blue sites from the four diagonal neighbours (k_rb_diag), on the tile
  // and one ring around it.  A site reads only sites this pass does not write (or itself, through
  // clamping, before its own write), so doing it in place is safe without a second buffer.
  for (int k = tid; k < (kTW + 2) * (kTH + 2); k += 256) {
    const int x = t.x0 - 1 + k % (kTW + 2), y = t.y0 - 1 + k / (kTW + 2);
    if (x < 0 || y < 0 || x >= w || y >= h) continue;
    const int c = cfa_at(x, y, pattern);
    if (c == CFA_G) continue;
    float* dst = (c == CFA_R) ? s_b : s_r;
    const float g0 = s_g[t.gi(x, y)];
    const float est = 0.25f * ((dst[t.si(x - 1, y - 1)] - s_g[t.gi(x - 1, y - 1)]) +
                               (dst[t.si(x + 1, y - 1)] - s_g[t.gi(x + 1, y - 1)]) +
                               (dst[t.si(x - 1, y + 1)] - s_g[t.gi(x - 1, y + 1)]) +
                               (dst[t.si(x + 1, y + 1)] - s_g[t.gi(x + 1, y + 1)])) + g0;
    dst[t.si(x, y)] = est;
  }
  __syncthreads();
  // chroma refinement and output (k_combine)
  for (int k = tid; k < kTW * kTH; k += 256) {
    const int x = t.x0 + k % kTW, y = t.y0 + k / kTW;
    if (x >= w || y >= h) continue;
    const float gh = fabsf(s_g[t.gi(x - 1, y)] - s_g[t.gi(x + 1, y)]);
    const float gv = fabsf(s_g[t.gi(x, y - 1)] - s_g[t.gi(x, y + 1)]);
    const float w_h = 1.0f / (1e-6f + gh);
    const float w_v = 1.0f / (1e-6f + gv);
    const float rgh = 0.5f * ((s_r[t.si(x - 1, y)] - s_g[t.gi(x - 1, y)]) + (s_r[t.si(x + 1, y)] - s_g[t.gi(x + 1, y)]));
    const float rgv = 0.5f * ((s_r[t.si(x, y - 1)] - s_g[t.gi(x, y - 1)]) + (s_r[t.si(x, y + 1)] - s_g[t.gi(x, y + 1)]));
    const float bgh = 0.5f * ((s_b[t.si(x - 1, y)] - s_g[t.gi(x - 1, y)]) + (s_b[t.si(x + 1, y)] - s_g[t.gi(x + 1, y)]));
    const float bgv = 0.5f * ((s_b[t.si(x, y - 1)] - s_g[t.gi(x, y - 1)]) + (s_b[t.si(x, y + 1)] - s_g[t.gi(x, y + 1)]));
    const float rg2 = (w_h * rgh + w_v * rgv) / (w_h + w_v);
    const float bg2 = (w_h * bgh + w_v * bgv) / (w_h + w_v);
    const float g0 = s_g[t.gi(x, y)];
    const size_t i = (size_t)y * w + x;
    out[3 * i + 0] = fmaxf(g0 + rg2, 0.0f);
    out[3 * i + 1] = fmaxf(g0, 0.0f);
    out[3 * i + 2] = fmaxf(g0 + bg2, 0.0f);
  }
}

template <int P>
__device__ __forceinline__ void interior_tile(const float* raw, float* out, int w, int x0, int y0, float* lds) {
  float* s_raw = lds;
  float* s_g = lds + dm::kPlane;
  float* s_rg = lds + 2 * dm::kPlane;
  float* s_bg = lds + 3 * dm::kPlane;
  const int tid = threadIdx.x;
  const bool pairs = (w & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;  // x0, y0 and the cells are even
  const float* src = raw + (size_t)(y0 - dm::kApron) * (size_t)w + (size_t)(x0 - dm::kApron);
  {  // a wave takes every fourth row: 64 columns by lane, the last four by its first lanes; all loads issued
     // before the first is waited for
    static_assert(dm::kStride == 68 && dm::kRows % 4 == 0, "row split below");
    constexpr int kPer = dm::kRows / 4;
    const int lane = tid & 63, first = tid >> 6;
    const float* row = src + (size_t)first * (size_t)w + lane;
    float v[kPer], e[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = row[(size_t)(4 * i) * (size_t)w];
    if (lane < 4) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) e[i] = row[(size_t)(4 * i) * (size_t)w + 64];
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) s_raw[(first + 4 * i) * dm::kStride + lane] = v[i];
    if (lane < 4) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) s_raw[(first + 4 * i) * dm::kStride + 64 + lane] = e[i];
    }
  }
  __syncthreads();
  static_assert(dm::kCellsW == 32, "cell index split below");
  for (int q = tid; q < dm::kCellsW * dm::kCellsH; q += 256)
    dm::green_cell<P>(s_raw, s_g, (2 + 2 * (q >> 5)) * dm::kStride + 2 + 2 * (q & 31));
  __syncthreads();
  for (int q = tid; q < dm::kCellsW * dm::kCellsH; q += 256)
    dm::chroma_cell<P>(s_raw, s_g, s_rg, s_bg, (2 + 2 * (q >> 5)) * dm::kStride + 2 + 2 * (q & 31));
  __syncthreads();
  for (int q = tid; q < dm::kOutCellsW * dm::kOutCellsH; q += 256) {
    const int qy = q / dm::kOutCellsW, qx = q - qy * dm::kOutCellsW;
    float row0[6], row1[6];
    dm::combine_cell(s_g, s_rg, s_bg, (dm::kApron + 2 * qy) * dm::kStride + dm::kApron + 2 * qx, row0, row1);
    float* o = out + 3 * ((size_t)(y0 + 2 * qy) * (size_t)w + (size_t)(x0 + 2 * qx));
    if (pairs) {  // both rows start on 8-byte boundaries
#pragma unroll
      for (int j = 0; j < 6; j += 2) dm::st2(o + j, row0[j], row0[j + 1]);
      o += 3 * (size_t)w;
#pragma unroll
      for (int j = 0; j < 6; j += 2) dm::st2(o + j, row1[j], row1[j + 1]);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = row0[j];
      o += 3 * (size_t)w;
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = row1[j];
    }
  }
}

template <int P>
__global__ void __launch_bounds__(256) k_gradient_tiles(const float* raw, float* out, int w, int h, dm::TileGrid tg) {
  __shared__ __attribute__((aligned(16))) float lds[kTileFloats];
  int b = (int)blockIdx.x;
  if (b >= tg.n_edge) {
    b -= tg.n_edge;
    const int iw = tg.ix1 - 1, by = b / iw, bx = b - by * iw;
    interior_tile<P>(raw, out, w, (1 + bx) * kTW, (1 + by) * kTH, lds);
    return;
  }
  // the frame of border tiles: the first row of tiles, the rows from iy1 on, and in between the first column and
  // the columns from ix1 on
  int bx, by;
  const int top = tg.gw, bottom = (tg.gh - tg.iy1) * tg.gw;
  if (b < top) {
    bx = b; by = 0;
  } else if (b < top + bottom) {
    b -= top;
    by = tg.iy1 + b / tg.gw; bx = b % tg.gw;
  } else {
    b -= top + bottom;
    const int mw = 1 + tg.gw - tg.ix1, c = b % mw;
    by = 1 + b / mw; bx = c == 0 ? 0 : tg.ix1 + c - 1;
  }
  border_tile(raw, out, w, h, P, bx * kTW, by * kTH, lds);
}

// src/main.rs:344-401
__global__ void k_bilinear(const float* raw, float* out, int w, int h, int pattern) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  const float v = raw[i];
  const int c = cfa_at(x, y, pattern);
  if (c == CFA_G) {
    const float hor = (getf(raw, w, h, x - 1, y) + getf(raw, w, h, x + 1, y)) * 0.5f;
    const float ver = (getf(raw, w, h, x, y - 1) + getf(raw, w, h, x, y + 1)) * 0.5f;
    const bool red_h = cfa_at(x - 1, y, pattern) == CFA_R || cfa_at(x + 1, y, pattern) == CFA_R;
    out[3 * i + 0] = red_h ? hor : ver;
    out[3 * i + 1] = v;
    out[3 * i + 2] = red_h ? ver : hor;
  } else {
    const float cross4 = (getf(raw, w, h, x - 1, y) + getf(raw, w, h, x + 1, y) + getf(raw, w, h, x, y - 1) +
                          getf(raw, w, h, x, y + 1)) * 0.25f;
    const float diag4 = (getf(raw, w, h, x - 1, y - 1) + getf(raw, w, h, x + 1, y - 1) + getf(raw, w, h, x - 1, y + 1) +
                         getf(raw, w, h, x + 1, y + 1)) * 0.25f;
    out[3 * i + 0] = c == CFA_R ? v : diag4;
    out[3 * i + 1] = cross4;
    out[3 * i + 2] = c == CFA_R ? diag4 : v;
  }
}

#define PT_KCHECK2()                                                                     \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "HIP launch error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      abort();                                                                           \
    }                                                                                    \
  } while (0)

void launch_demosaic(const float* d_raw, float* d_out, float* d_tmp, int w, int h, int pattern, int algo, hipStream_t s) {
  const dim3 block(64, 4);
  const dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4));
  if (algo == 1) {
    k_bilinear<<<grid, block, 0, s>>>(d_raw, d_out, w, h, pattern);
    PT_KCHECK2();
    return;
  }
#ifdef PT_DEMOSAIC_PASSES
  if (getenv("PT_DEMOSAIC_PASSES") != nullptr && d_tmp) {  // the four-pass form, for comparison
    const size_t n = (size_t)w * (size_t)h;
    float* g = d_tmp;
    float* r = d_tmp + n;
    float* b = d_tmp + 2 * n;
    k_green<<<grid, block, 0, s>>>(d_raw, g, w, h, pattern);
    PT_KCHECK2();
    k_rb_at_g<<<grid, block, 0, s>>>(d_raw, g, r, b, w, h, pattern);
    PT_KCHECK2();
    k_rb_diag<<<grid, block, 0, s>>>(g, r, b, w, h, pattern);
    PT_KCHECK2();
    k_combine<<<grid, block, 0, s>>>(r, g, b, d_out, w, h);
    PT_KCHECK2();
    return;
  }
#endif
  (void)d_tmp;
  const dm::TileGrid tg = dm::tile_grid(w, h);
  const unsigned tiles = (unsigned)(tg.gw * tg.gh);
  switch (pattern < 0 ? 0 : (pattern > 3 ? 3 : pattern)) {  // cfa_at clamps the pattern the same way
    case 0: k_gradient_tiles<0><<<tiles, 256, 0, s>>>(d_raw, d_out, w, h, tg); break;
    case 1: k_gradient_tiles<1><<<tiles, 256, 0, s>>>(d_raw, d_out, w, h, tg); break;
    case 2: k_gradient_tiles<2><<<tiles, 256, 0, s>>>(d_raw, d_out, w, h, tg); break;
    default: k_gradient_tiles<3><<<tiles, 256, 0, s>>>(d_raw, d_out, w, h, tg); break;
  }
  PT_KCHECK2();
}

}  // namespace pt
