// pt_rng.h -- random numbers, pixel jitter and camera rays (a3, a4, a5, a12).
#pragma once
#include "pt_types.h"

namespace pt {

// optix_device.cu:148-154: PCG-XSH-RR 64 -> 32
PT_HD uint32_t pcg(uint64_t& state) {
  uint64_t old = state;
  state = old * 6364136223846793005ULL + 1442695040888963407ULL;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
// optix_device.cu:156-158: 24-bit mantissa in [0,1)
PT_HD float rnd(uint64_t& state) { return (float)(pcg(state) >> 8) * (1.0f / 16777216.0f); }

// optix_device.cu:164-174
PT_HD uint32_t hash_u32(uint32_t x) {
  x ^= x >> 17; x *= 0xed5ad4bbU;
  x ^= x >> 11; x *= 0xac4c1b51U;
  x ^= x >> 15; x *= 0x31848babU;
  x ^= x >> 14;
  return x;
}

// optix_device.cu:15-17, 176-185: R2 sequence with a per-pixel Cranley-Patterson shift.  The shift depends
// on the pixel and the frame only, so a kernel that works on one pixel for a while computes it once
// (pixel_shift) and derives each sample's pair from it (blue_noise_shifted); blue_noise is the two together.
PT_HD void pixel_shift(int x, int y, int frame, float& rx, float& ry) {
  const float inv_u16 = 1.0f / 65536.0f;
  uint32_t h = hash_u32((uint32_t)x * 1973u + (uint32_t)y * 9277u + (uint32_t)frame * 26699u);
  rx = (float)(h & 0xffffu) * inv_u16;
  ry = (float)(h >> 16) * inv_u16;
}
PT_HD void blue_noise_shifted(float rx, float ry, int s, float& u, float& v) {
  const float inv_plastic = 1.0f / 1.32471795724474602596f;
  const float inv_plastic_sqr = 1.0f / (1.32471795724474602596f * 1.32471795724474602596f);
  float fs = (float)s + 0.5f;
  u = fract_pos(rx + fs * inv_plastic);
  v = fract_pos(ry + fs * inv_plastic_sqr);
}
PT_HD void blue_noise(int x, int y, int s, int frame, float& u, float& v) {
  float rx, ry;
  pixel_shift(x, y, frame, rx, ry);
  blue_noise_shifted(rx, ry, s, u, v);
}

// optix_device.cu:187-190: triangle-filter warp to [-1,1]
// (one square root for both halves: the operand is chosen first)
PT_HD float sample_tri(float u) {
  const bool low = u < 0.5f;
  const float r = fsqrt(2.0f * (low ? u : 1.0f - u));
  return low ? r - 1.0f : 1.0f - r;
}

// optix_device.cu:945-948
PT_HD uint64_t pixel_seed(int frame, uint32_t dst) {
  return ((uint64_t)(int64_t)frame * 9781ULL) ^ ((uint64_t)(int64_t)(int32_t)dst * 6271ULL) ^ 0x853c49e6748fea9bULL;
}
// Start of the PCG stream of a sample chunk that begins at global sample index `begin`.
// begin == 0 is the reference's seed; later chunks get a decorrelated start so that
// sample ranges rendered on different GPUs do not need each other's stream position.
// The start is passed through a 64-bit finaliser: the first PCG output is a function of the state's high
// bits alone, which pixel_seed leaves the same for every pixel of an image (the reference's first draw of a
// stream varies by under 0.01 over a frame), and a render split into one-sample chunks takes only first draws.
PT_HD uint64_t chunk_seed(int frame, uint32_t dst, int begin) {
  uint64_t s = pixel_seed(frame, dst);
  if (begin != 0) {
    s ^= (uint64_t)(int64_t)begin * 0x9E3779B97F4A7C15ULL;
    s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ULL;
    s ^= s >> 27; s *= 0x94D049BB133111EBULL;
    s ^= s >> 31;
  }
  return s;
}

// optix_device.cu:890-900
PT_HD f3 camera_dir(const CameraView& cam, int x, int y, float jx, float jy, int width, int height) {
  float fx = ((float)x + 0.5f + jx) * rcp((float)width);
  float fy = ((float)y + 0.5f + jy) * rcp((float)height);
  fx = clamp01(fx);
  fy = clamp01(fy);
  float dx = 2.0f * fx - 1.0f, dy = 1.0f - 2.0f * fy;
  f3 d = mk3(cam.w[0] + dx * cam.u[0] + dy * cam.v[0], cam.w[1] + dx * cam.u[1] + dy * cam.v[1],
             cam.w[2] + dx * cam.u[2] + dy * cam.v[2]);
  return normalize3(d);
}

// optix_device.cu:902-912 (== src/main.rs:157-173 for patterns 0..3)
PT_HD int bayer_channel_for(int x, int y, int pattern) {
  const int mx = x & 1, my = y & 1;
  switch (pattern) {
    case 0: return (my == 0) ? (mx == 0 ? 0 : 1) : (mx == 0 ? 1 : 2);
    case 1: return (my == 0) ? (mx == 0 ? 2 : 1) : (mx == 0 ? 1 : 0);
    case 2: return (my == 0) ? (mx == 0 ? 1 : 0) : (mx == 0 ? 2 : 1);
    default: return (my == 0) ? (mx == 0 ? 1 : 2) : (mx == 0 ? 0 : 1);
  }
}

// optix_device.cu:211-222
PT_HD void make_onb(f3 n, f3& t, f3& b) {
  if (n.z < -0.9999999f) {
    t = mk3(0.0f, -1.0f, 0.0f);
    b = mk3(-1.0f, 0.0f, 0.0f);
  } else {
    const float a = rcp(1.0f + n.z);
    const float naxy = -n.x * n.y * a;
    t = mk3(1.0f - n.x * n.x * a, naxy, -n.x);
    b = mk3(naxy, 1.0f - n.y * n.y * a, -n.y);
  }
}

}  // namespace pt
