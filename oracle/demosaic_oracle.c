/*
 * demosaic_oracle.c -- CPU restatement of the reference's CFA lookup and the two
 * demosaic routines, sequential and in place exactly like the Rust loops.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 *   src/main.rs:157-173  cfa_at
 *   src/main.rs:175-185  getf / setf (edge-clamped access)
 *   src/main.rs:190-342  demosaic_amaze
 *   src/main.rs:344-401  demosaic_bilinear
 *
 * Rust f32 arithmetic never contracts to FMA; build with -ffp-contract=off.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>

enum { CFA_R = 0, CFA_G = 1, CFA_B = 2 };

/* src/main.rs:157-173 */
int orc_cfa_at(int x, int y, int pattern) {
  static const int P[4][2][2] = {
      {{CFA_R, CFA_G}, {CFA_G, CFA_B}}, /* 0: RGGB */
      {{CFA_B, CFA_G}, {CFA_G, CFA_R}}, /* 1: BGGR */
      {{CFA_G, CFA_R}, {CFA_B, CFA_G}}, /* 2: GRBG */
      {{CFA_G, CFA_B}, {CFA_R, CFA_G}}, /* 3: GBRG */
  };
  int p = pattern < 0 ? 0 : (pattern > 3 ? 3 : pattern);
  return P[p][y & 1][x & 1];
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* src/main.rs:175-185 */
static inline float getf(const float* img, int w, int h, int x, int y) {
  return img[(size_t)clampi(y, 0, h - 1) * (size_t)w + (size_t)clampi(x, 0, w - 1)];
}
static inline void setf(float* img, int w, int h, int x, int y, float v) {
  img[(size_t)clampi(y, 0, h - 1) * (size_t)w + (size_t)clampi(x, 0, w - 1)] = v;
}
static inline float maxf0(float v) { return v > 0.0f ? v : 0.0f; } /* f32::max(0.0) */

/* src/main.rs:190-342 */
void orc_demosaic_amaze(const float* raw, int w, int h, int pattern, float* out) {
  const size_t n = (size_t)w * (size_t)h;
  float* r = (float*)calloc(n, sizeof(float));
  float* g = (float*)calloc(n, sizeof(float));
  float* b = (float*)calloc(n, sizeof(float));
  /* 195-204 */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float v = raw[(size_t)y * w + x];
      switch (orc_cfa_at(x, y, pattern)) {
        case CFA_R: r[(size_t)y * w + x] = v; break;
        case CFA_G: g[(size_t)y * w + x] = v; break;
        default:    b[(size_t)y * w + x] = v; break;
      }
    }
  /* 205-237: green at red/blue sites */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int c = orc_cfa_at(x, y, pattern);
      if (c == CFA_R || c == CFA_B) {
        const float* cx = (c == CFA_R) ? r : b;
        float c0 = getf(cx, w, h, x, y);
        float c_l2 = getf(cx, w, h, x - 2, y), c_l1 = getf(cx, w, h, x - 1, y);
        float c_r1 = getf(cx, w, h, x + 1, y), c_r2 = getf(cx, w, h, x + 2, y);
        float c_u2 = getf(cx, w, h, x, y - 2), c_u1 = getf(cx, w, h, x, y - 1);
        float c_d1 = getf(cx, w, h, x, y + 1), c_d2 = getf(cx, w, h, x, y + 2);
        float g_l1 = getf(g, w, h, x - 1, y), g_r1 = getf(g, w, h, x + 1, y);
        float g_u1 = getf(g, w, h, x, y - 1), g_d1 = getf(g, w, h, x, y + 1);
        float grad_h = fabsf(g_l1 - g_r1) + fabsf(2.0f * c0 - c_l2 - c_r2) * 0.5f + fabsf(c_l1 - c_r1);
        float grad_v = fabsf(g_u1 - g_d1) + fabsf(2.0f * c0 - c_u2 - c_d2) * 0.5f + fabsf(c_u1 - c_d1);
        float gh = 0.5f * (g_l1 + g_r1) + 0.25f * (2.0f * c0 - c_l2 - c_r2);
        float gv = 0.5f * (g_u1 + g_d1) + 0.25f * (2.0f * c0 - c_u2 - c_d2);
        float wh = 1.0f / (1e-6f + grad_h * grad_h);
        float wv = 1.0f / (1e-6f + grad_v * grad_v);
        float g_est = (wh * gh + wv * gv) / (wh + wv);
        setf(g, w, h, x, y, g_est);
      }
    }
  /* 238-276: red/blue at green sites */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if (orc_cfa_at(x, y, pattern) == CFA_G) {
        float g0 = getf(g, w, h, x, y);
        int has_red_h = orc_cfa_at(x - 1, y, pattern) == CFA_R || orc_cfa_at(x + 1, y, pattern) == CFA_R;
        int has_red_v = orc_cfa_at(x, y - 1, pattern) == CFA_R || orc_cfa_at(x, y + 1, pattern) == CFA_R;
        int has_blue_h = orc_cfa_at(x - 1, y, pattern) == CFA_B || orc_cfa_at(x + 1, y, pattern) == CFA_B;
        int has_blue_v = orc_cfa_at(x, y - 1, pattern) == CFA_B || orc_cfa_at(x, y + 1, pattern) == CFA_B;
        float r_est, b_est;
        if (has_red_h)
          r_est = 0.5f * (getf(r, w, h, x - 1, y) + getf(r, w, h, x + 1, y)) -
                  0.5f * (getf(g, w, h, x - 1, y) + getf(g, w, h, x + 1, y)) + g0;
        else if (has_red_v)
          r_est = 0.5f * (getf(r, w, h, x, y - 1) + getf(r, w, h, x, y + 1)) -
                  0.5f * (getf(g, w, h, x, y - 1) + getf(g, w, h, x, y + 1)) + g0;
        else
          r_est = g0;
        if (has_blue_h)
          b_est = 0.5f * (getf(b, w, h, x - 1, y) + getf(b, w, h, x + 1, y)) -
                  0.5f * (getf(g, w, h, x - 1, y) + getf(g, w, h, x + 1, y)) + g0;
        else if (has_blue_v)
          b_est = 0.5f * (getf(b, w, h, x, y - 1) + getf(b, w, h, x, y + 1)) -
                  0.5f * (getf(g, w, h, x, y - 1) + getf(g, w, h, x, y + 1)) + g0;
        else
          b_est = g0;
        setf(r, w, h, x, y, r_est);
        setf(b, w, h, x, y, b_est);
      }
    }
  /* 277-303: blue at red sites, red at blue sites (diagonals of colour minus green) */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int c = orc_cfa_at(x, y, pattern);
      if (c == CFA_R) {
        float g0 = getf(g, w, h, x, y);
        float b_est = 0.25f * ((getf(b, w, h, x - 1, y - 1) - getf(g, w, h, x - 1, y - 1)) +
                               (getf(b, w, h, x + 1, y - 1) - getf(g, w, h, x + 1, y - 1)) +
                               (getf(b, w, h, x - 1, y + 1) - getf(g, w, h, x - 1, y + 1)) +
                               (getf(b, w, h, x + 1, y + 1) - getf(g, w, h, x + 1, y + 1))) + g0;
        setf(b, w, h, x, y, b_est);
      } else if (c == CFA_B) {
        float g0 = getf(g, w, h, x, y);
        float r_est = 0.25f * ((getf(r, w, h, x - 1, y - 1) - getf(g, w, h, x - 1, y - 1)) +
                               (getf(r, w, h, x + 1, y - 1) - getf(g, w, h, x + 1, y - 1)) +
                               (getf(r, w, h, x - 1, y + 1) - getf(g, w, h, x - 1, y + 1)) +
                               (getf(r, w, h, x + 1, y + 1) - getf(g, w, h, x + 1, y + 1))) + g0;
        setf(r, w, h, x, y, r_est);
      }
    }
  /* 304-309 */
  float* rg = (float*)calloc(n, sizeof(float));
  float* bg = (float*)calloc(n, sizeof(float));
  for (size_t i = 0; i < n; ++i) { rg[i] = r[i] - g[i]; bg[i] = b[i] - g[i]; }
  /* 311-326 */
  float* rg2 = (float*)calloc(n, sizeof(float));
  float* bg2 = (float*)calloc(n, sizeof(float));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float gh = fabsf(getf(g, w, h, x - 1, y) - getf(g, w, h, x + 1, y));
      float gv = fabsf(getf(g, w, h, x, y - 1) - getf(g, w, h, x, y + 1));
      float w_h = 1.0f / (1e-6f + gh);
      float w_v = 1.0f / (1e-6f + gv);
      float rgh = 0.5f * (getf(rg, w, h, x - 1, y) + getf(rg, w, h, x + 1, y));
      float rgv = 0.5f * (getf(rg, w, h, x, y - 1) + getf(rg, w, h, x, y + 1));
      float bgh = 0.5f * (getf(bg, w, h, x - 1, y) + getf(bg, w, h, x + 1, y));
      float bgv = 0.5f * (getf(bg, w, h, x, y - 1) + getf(bg, w, h, x, y + 1));
      setf(rg2, w, h, x, y, (w_h * rgh + w_v * rgv) / (w_h + w_v));
      setf(bg2, w, h, x, y, (w_h * bgh + w_v * bgv) / (w_h + w_v));
    }
  /* 327-341 */
  for (size_t i = 0; i < n; ++i) {
    r[i] = maxf0(g[i] + rg2[i]);
    b[i] = maxf0(g[i] + bg2[i]);
  }
  for (size_t i = 0; i < n; ++i) {
    out[i * 3 + 0] = maxf0(r[i]);
    out[i * 3 + 1] = maxf0(g[i]);
    out[i * 3 + 2] = maxf0(b[i]);
  }
  free(r); free(g); free(b); free(rg); free(bg); free(rg2); free(bg2);
}

/* src/main.rs:344-401 */
void orc_demosaic_bilinear(const float* raw, int w, int h, int pattern, float* out) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      size_t i = (size_t)y * (size_t)w + (size_t)x;
      float v = raw[i];
      float cross4 = (getf(raw, w, h, x - 1, y) + getf(raw, w, h, x + 1, y) + getf(raw, w, h, x, y - 1) +
                      getf(raw, w, h, x, y + 1)) * 0.25f;
      float diag4 = (getf(raw, w, h, x - 1, y - 1) + getf(raw, w, h, x + 1, y - 1) + getf(raw, w, h, x - 1, y + 1) +
                     getf(raw, w, h, x + 1, y + 1)) * 0.25f;
      switch (orc_cfa_at(x, y, pattern)) {
        case CFA_R: out[i * 3] = v; out[i * 3 + 1] = cross4; out[i * 3 + 2] = diag4; break;
        case CFA_B: out[i * 3] = diag4; out[i * 3 + 1] = cross4; out[i * 3 + 2] = v; break;
        default: {
          float hor = (getf(raw, w, h, x - 1, y) + getf(raw, w, h, x + 1, y)) * 0.5f;
          float ver = (getf(raw, w, h, x, y - 1) + getf(raw, w, h, x, y + 1)) * 0.5f;
          int red_h = orc_cfa_at(x - 1, y, pattern) == CFA_R || orc_cfa_at(x + 1, y, pattern) == CFA_R;
          out[i * 3] = red_h ? hor : ver;
          out[i * 3 + 1] = v;
          out[i * 3 + 2] = red_h ? ver : hor;
        } break;
      }
    }
}
