// pt_guiding.h -- path-guiding field lookup and von Mises-Fisher sampling (a13).
// Restates optix/guiding_gpu.cuh:35-118 for the gfx950 shade stage.
#pragma once
#include "pt_types.h"

namespace pt {

// guiding_gpu.cuh:35-45: uniform grid cell -> region id, -1 when the cell is empty
PT_HD int guiding_region_id(const GuideView& G, f3 P) {
  const float rx = (P.x - G.gmin[0]) / G.cell[0];
  const float ry = (P.y - G.gmin[1]) / G.cell[1];
  const float rz = (P.z - G.gmin[2]) / G.cell[2];
  int cx = (int)rx, cy = (int)ry, cz = (int)rz;
  cx = cx > G.res[0] - 1 ? G.res[0] - 1 : cx; cx = cx < 0 ? 0 : cx;
  cy = cy > G.res[1] - 1 ? G.res[1] - 1 : cy; cy = cy < 0 ? 0 : cy;
  cz = cz > G.res[2] - 1 ? G.res[2] - 1 : cz; cz = cz < 0 ? 0 : cz;
  const uint32_t id = G.grid[(cz * G.res[1] + cy) * G.res[0] + cx];
  return id == 0xFFFFFFFFu ? -1 : (int)id;
}

// guiding_gpu.cuh:60-78
PT_HD f3 vmf_sample(f3 mu, float kappa, float ux, float uy) {
  float cos_theta;
  if (kappa <= 1e-4f) {
    cos_theta = 1.f - 2.f * ux;
  } else {
    const float term = fexp(-2.f * kappa);
    cos_theta = 1.f + flog(term + (1.f - term) * ux) * rcp(kappa);
  }
  const float sin_theta = fsqrt(fmaxf(0.f, 1.f - cos_theta * cos_theta));
  const f3 local = mk3(cos2pi(uy) * sin_theta, sin2pi(uy) * sin_theta, cos_theta);
  const f3 w = normalize_chk(mu);
  const f3 up = fabsf(w.z) < 0.9999999f ? mk3(0.f, 0.f, 1.f) : mk3(1.f, 0.f, 0.f);
  const f3 u_axis = normalize_chk(cross(up, w));
  const f3 v_axis = cross(w, u_axis);
  return mk3(u_axis.x * local.x + v_axis.x * local.y + w.x * local.z,
             u_axis.y * local.x + v_axis.y * local.y + w.y * local.z,
             u_axis.z * local.x + v_axis.z * local.y + w.z * local.z);
}

// guiding_gpu.cuh:80-84.  kappa / (4 pi sinh kappa) * exp(kappa mu.w) is evaluated in the
// algebraically identical form kappa / (2 pi (1 - e^{-2 kappa})) * exp(kappa (mu.w - 1)),
// which stays finite where sinhf/expf of the literal form overflow.
PT_HD float vmf_pdf(f3 mu, float kappa, f3 wi) {
  if (kappa <= 1e-4f) return 1.f / (4.f * kPi);
  const float norm = kappa * rcp(2.f * kPi * (1.f - fexp(-2.f * kappa)));
  return norm * fexp(kappa * (dot(mu, wi) - 1.f));
}

PT_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// guiding_gpu.cuh:86-106 with ch = -1 (the only value the live path passes, optix_device.cu:615)
PT_HD int guiding_choose_lobe(const GuideLobe* L, uint32_t lobe_num, float u) {
  float sum = 0.f;
  for (uint32_t i = 0; i < lobe_num; ++i) sum += L[i].weight;
  if (sum <= 0.f) return -1;
  const float r = u * sum;
  float acc = 0.f;
  for (uint32_t i = 0; i < lobe_num; ++i) {
    acc += L[i].weight;
    if (r <= acc) return (int)i;
  }
  return (int)(lobe_num - 1);
}

// guiding_gpu.cuh:112-118
PT_HD float guiding_mixture_pdf(const GuideLobe* L, uint32_t lobe_num, f3 wi) {
  float sum = 0.f, pdf = 0.f;
  for (uint32_t i = 0; i < lobe_num; ++i) {
    const float w = L[i].weight;
    sum += w;
    pdf += w * vmf_pdf(ld3(L[i].mu), L[i].kappa, wi);
  }
  return sum > 0.f ? pdf / sum : 0.f;
}

}  // namespace pt
