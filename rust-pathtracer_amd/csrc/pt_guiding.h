// pt_guiding.h -- path-guiding field lookup and von Mises-Fisher sampling (a13).
// The arithmetic of optix/guiding_gpu.cuh:35-118 for the gfx950 shade stage, on records that carry what the
// reference recomputes at every query (pt_types.h: GuideLobeDev): one pass over a region's lobes picks the
// lobe from running weight sums, one pass evaluates the mixture pdf with one exp2 per lobe.
#pragma once
#include "pt_types.h"

namespace pt {

PT_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// guiding_gpu.cuh:35-45: uniform grid cell -> region id, -1 when the cell is empty
PT_HD int guiding_region_id(const GuideView& G, f3 P) {
  const float rx = (P.x - G.gmin[0]) * G.inv_cell[0];
  const float ry = (P.y - G.gmin[1]) * G.inv_cell[1];
  const float rz = (P.z - G.gmin[2]) * G.inv_cell[2];
  int cx = (int)rx, cy = (int)ry, cz = (int)rz;
  cx = cx > G.res[0] - 1 ? G.res[0] - 1 : cx; cx = cx < 0 ? 0 : cx;
  cy = cy > G.res[1] - 1 ? G.res[1] - 1 : cy; cy = cy < 0 ? 0 : cy;
  cz = cz > G.res[2] - 1 ? G.res[2] - 1 : cz; cz = cz < 0 ? 0 : cz;
  const uint32_t id = G.grid[(cz * G.res[1] + cy) * G.res[0] + cx];
  return id == 0xFFFFFFFFu ? -1 : (int)id;
}

// The same lookup returning the cell's region record itself (lobe_num == 0: no region or no lobes)
PT_HD GuideRegionDev guiding_cell(const GuideView& G, f3 P) {
  const float rx = (P.x - G.gmin[0]) * G.inv_cell[0];
  const float ry = (P.y - G.gmin[1]) * G.inv_cell[1];
  const float rz = (P.z - G.gmin[2]) * G.inv_cell[2];
  int cx = (int)rx, cy = (int)ry, cz = (int)rz;
  cx = cx > G.res[0] - 1 ? G.res[0] - 1 : cx; cx = cx < 0 ? 0 : cx;
  cy = cy > G.res[1] - 1 ? G.res[1] - 1 : cy; cy = cy < 0 ? 0 : cy;
  cz = cz > G.res[2] - 1 ? G.res[2] - 1 : cz; cz = cz < 0 ? 0 : cz;
  return G.cells[(cz * G.res[1] + cy) * G.res[0] + cx];
}

// guiding_gpu.cuh:60-78 with e^(-2 kappa), 1 / kappa and the normalised axis taken from the record
PT_HD f3 vmf_sample(const GuideLobeDev& lb, float ux, float uy) {
  float cos_theta;
  if (lb.kappa <= 1e-4f) {
    cos_theta = 1.f - 2.f * ux;
  } else {
    cos_theta = 1.f + flog(lb.term + (1.f - lb.term) * ux) * lb.inv_kappa;
  }
  const float sin_theta = fsqrt(fmaxf(0.f, 1.f - cos_theta * cos_theta));
  const f3 local = mk3(cos2pi(uy) * sin_theta, sin2pi(uy) * sin_theta, cos_theta);
  const f3 w = ld3(lb.axis);
  const f3 up = fabsf(w.z) < 0.9999999f ? mk3(0.f, 0.f, 1.f) : mk3(1.f, 0.f, 0.f);
  const f3 u_axis = normalize_chk(cross(up, w));
  const f3 v_axis = cross(w, u_axis);
  return mk3(u_axis.x * local.x + v_axis.x * local.y + w.x * local.z,
             u_axis.y * local.x + v_axis.y * local.y + w.y * local.z,
             u_axis.z * local.x + v_axis.z * local.y + w.z * local.z);
}

// guiding_gpu.cuh:86-106 with ch = -1 (the only value the live path passes, optix_device.cu:615): the first
// lobe whose running weight sum reaches u * total.  The sums are the reference's own float additions, done once.
PT_HD int guiding_choose_lobe(const GuideLobeDev* L, uint32_t lobe_num, float sum, float u) {
  if (sum <= 0.f) return -1;
  const float r = u * sum;
  for (uint32_t i = 0; i < lobe_num; ++i)
    if (r <= L[i].cdf) return (int)i;
  return (int)(lobe_num - 1);
}

// guiding_gpu.cuh:80-84, 112-118.  kappa / (4 pi sinh kappa) * exp(kappa mu.w) in the algebraically identical form
// kappa / (2 pi (1 - e^{-2 kappa})) * exp(kappa (mu.w - 1)), which stays finite where sinhf / expf of the literal
// form overflow; the factor in front (times the lobe's weight) is the record's c, the exponential one exp2.
PT_HD float guiding_mixture_pdf(const GuideLobeDev* L, uint32_t lobe_num, float sum, f3 wi) {
  float pdf = 0.f;
  for (uint32_t i = 0; i < lobe_num; ++i) {
    const float e = L[i].kappa <= 1e-4f ? 1.0f : fexp2(L[i].k2 * (dot(ld3(L[i].mu), wi) - 1.f));
    pdf += L[i].c * e;
  }
  return sum > 0.f ? pdf * rcp(sum) : 0.f;
}

// The two passes over a region's lobes with their loads issued together.  A lobe record is four 16-byte words
// (mu kappa | cdf c k2 term | rgb 1/kappa | axis weight); regions of up to kGuideBatch lobes -- what trainers
// produce -- read the second word of all their lobes at once for the choice and the first word of all of them at
// once for the mixture pdf, instead of one dependent load per lobe and pass (each a round trip to L2 from a kernel
// that otherwise never leaves the CU).  Loads past the region's last lobe re-read that lobe and are ignored.
constexpr uint32_t kGuideBatch = 4;
struct GuideBatch {
  float cdf[kGuideBatch], c[kGuideBatch], k2[kGuideBatch];  // per lobe: running weight sum, weight x normalisation, kappa x log2(e)
};
// the second word of every lobe of the region (lobe_num in 1..kGuideBatch), all loads in flight together
PT_HD void guiding_load_batch(const GuideLobeDev* L, uint32_t lobe_num, GuideBatch& gb) {
  const q4* w = reinterpret_cast<const q4*>(L);
  q4 b[kGuideBatch];
#pragma unroll
  for (uint32_t i = 0; i < kGuideBatch; ++i) b[i] = w[4u * (i < lobe_num ? i : lobe_num - 1u) + 1u];
#pragma unroll
  for (uint32_t i = 0; i < kGuideBatch; ++i) { gb.cdf[i] = b[i].x; gb.c[i] = b[i].y; gb.k2[i] = b[i].z; }
}
// guiding_choose_lobe on a loaded batch
PT_HD int guiding_choose_lobe_batch(const GuideBatch& gb, uint32_t lobe_num, float sum, float u) {
  if (sum <= 0.f) return -1;
  const float r = u * sum;
  int idx = (int)lobe_num - 1;
#pragma unroll
  for (int i = (int)kGuideBatch - 2; i >= 0; --i)  // descending, so that the first lobe whose running sum reaches r wins
    if ((uint32_t)i + 1u < lobe_num && r <= gb.cdf[i]) idx = i;
  return idx;
}
// the first word (mu, kappa) of every lobe of the region, all loads in flight together ...
struct GuideAxes {
  q4 a[kGuideBatch];
};
PT_HD void guiding_load_axes(const GuideLobeDev* L, uint32_t lobe_num, GuideAxes& ga) {
  const q4* w = reinterpret_cast<const q4*>(L);
#pragma unroll
  for (uint32_t i = 0; i < kGuideBatch; ++i) ga.a[i] = w[4u * (i < lobe_num ? i : lobe_num - 1u)];
}
// ... and guiding_mixture_pdf on what was loaded
PT_HD float guiding_mixture_pdf_batch(const GuideAxes& ga, const GuideBatch& gb, uint32_t lobe_num, float sum, f3 wi) {
  float pdf = 0.f;
#pragma unroll
  for (uint32_t i = 0; i < kGuideBatch; ++i) {
    const q4 a = ga.a[i];
    const float e = a.w <= 1e-4f ? 1.0f : fexp2(gb.k2[i] * (dot(mk3(a.x, a.y, a.z), wi) - 1.f));
    if (i < lobe_num) pdf += gb.c[i] * e;
  }
  return sum > 0.f ? pdf * rcp(sum) : 0.f;
}

// ---- training statistics binned on the device
// The trainer's field as the binning kernel needs it: bounds, resolution, and (separately) the axis of every lobe.
struct BinField {
  float bmin[3], bmax[3];
  int res[3];
};
constexpr int kBinLobes = 6, kBinWords = 8, kBinShift = 24;
// Records the trainer uses: weight in [2^-25, 2^20], no colour component beyond 2^20 in magnitude -- what the table's
// 2^-24 steps and 64-bit sums hold exactly (a bin takes 2^19 records of the largest weight before it could wrap).
constexpr float kTrainMaxWeight = 1048576.0f, kTrainMinWeight = 2.98023223876953125e-8f;
PT_HD bool train_weight_usable(float weight, const float contrib[3]) {
  if (!(weight >= kTrainMinWeight) || !(weight <= kTrainMaxWeight)) return false;  // a NaN fails both
  for (int k = 0; k < 3; ++k)
    if (!(fabsf(contrib[k]) <= kTrainMaxWeight)) return false;
  return true;
}

// One ring record -> the (region, lobe) it trains and its eight fixed-point addends; false when the trainer would
// drop it (torn or empty slot, outside the field, weight or colour outside the range the table holds exactly).  The assignment repeats the
// trainer's own float arithmetic operation for operation (no fused multiply-adds), so both agree on every record.
PT_HD bool bin_train_record(const TrainRec& rec, const BinField& f, const float* axes, uint32_t& bin, long long add[kBinWords]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (rec.is_delta == 0xffffffffu || (rec.is_delta & ~1u) != train_digest(rec)) return false;
  int c[3];
  for (int k = 0; k < 3; ++k) {
    const float ext = f.bmax[k] - f.bmin[k];
    if (!(rec.position[k] >= f.bmin[k] && rec.position[k] <= f.bmax[k]) || !(ext > 0.f)) return false;  // a NaN is outside
    c[k] = (int)((rec.position[k] - f.bmin[k]) / ext * (float)f.res[k]);
    if (c[k] >= f.res[k]) c[k] = f.res[k] - 1;
  }
  const float weight = (rec.contrib[0] + rec.contrib[1] + rec.contrib[2]) / 3.0f;
  if (!train_weight_usable(weight, rec.contrib)) return false;
  const uint32_t region = (uint32_t)((c[2] * f.res[1] + c[1]) * f.res[0] + c[0]);
  const float* mu = axes + (size_t)region * kBinLobes * 3;
  int best = 0;
  float best_d = -2.f;
  for (int l = 0; l < kBinLobes; ++l) {
    const float d = rec.dir_in[0] * mu[3 * l] + rec.dir_in[1] * mu[3 * l + 1] + rec.dir_in[2] * mu[3 * l + 2];
    if (d > best_d) { best_d = d; best = l; }
  }
  bin = region * kBinLobes + (uint32_t)best;
  const double scale = (double)(1ll << kBinShift);
  for (int k = 0; k < 3; ++k) {
    add[k] = (long long)__builtin_rint((double)weight * (double)rec.dir_in[k] * scale);
    add[4 + k] = (long long)__builtin_rint((double)rec.contrib[k] * scale);
  }
  add[3] = (long long)__builtin_rint((double)weight * scale);
  add[7] = 1;
  return true;
}

// ---- host side: snapshot (optix/guiding_gpu.h:7-22 layouts) -> device records
inline GuideLobeDev make_guide_lobe(const GuideLobe& l, float cdf) {
  GuideLobeDev d;
  for (int k = 0; k < 3; ++k) { d.mu[k] = l.mu[k]; d.rgb[k] = l.rgb[k]; }
  d.kappa = l.kappa;
  d.weight = l.weight;
  d.cdf = cdf;
  const double kp = (double)l.kappa;
  if (l.kappa <= 1e-4f) {
    d.c = (float)((double)l.weight / (4.0 * 3.14159265358979323846));
    d.term = 1.0f;
    d.inv_kappa = 0.0f;
  } else {
    d.c = (float)((double)l.weight * kp / (2.0 * 3.14159265358979323846 * (1.0 - exp(-2.0 * kp))));
    d.term = (float)exp(-2.0 * kp);
    d.inv_kappa = (float)(1.0 / kp);
  }
  d.k2 = (float)(kp * 1.4426950408889634);
  const double len2 = (double)l.mu[0] * l.mu[0] + (double)l.mu[1] * l.mu[1] + (double)l.mu[2] * l.mu[2];
  const double inv = len2 > 0.0 ? 1.0 / sqrt(len2) : 0.0;  // normalize_chk: a zero axis stays zero
  for (int k = 0; k < 3; ++k) d.axis[k] = (float)((double)l.mu[k] * inv);
  return d;
}
// regions[i] = {lobe range, total weight}; lobes in the snapshot's order
inline void make_guide_records(const GuideRegion* regions, uint32_t nr, const GuideLobe* lobes, uint32_t nl,
                               GuideRegionDev* out_regions, GuideLobeDev* out_lobes) {
  for (uint32_t i = 0; i < nl; ++i) out_lobes[i] = make_guide_lobe(lobes[i], 0.0f);
  for (uint32_t r = 0; r < nr; ++r) {
    GuideRegionDev d{regions[r].lobe_ofs, regions[r].lobe_num, 0.0f, 0.0f};
    if ((uint64_t)d.lobe_ofs + d.lobe_num > nl) d.lobe_num = d.lobe_ofs < nl ? nl - d.lobe_ofs : 0u;  // stay inside the array
    float acc = 0.0f;
    for (uint32_t i = 0; i < d.lobe_num; ++i) {  // the float additions of guiding_gpu.cuh:93-103, in their order
      acc += lobes[d.lobe_ofs + i].weight;
      out_lobes[d.lobe_ofs + i].cdf = acc;
    }
    d.sum = acc;
    out_regions[r] = d;
  }
}
// cells[c] = the record of the region grid[c] names, region id kept in pad; an empty cell has no lobes
inline void make_guide_cells(const uint32_t* grid, size_t n_cells, const GuideRegionDev* regions, uint32_t nr,
                             GuideRegionDev* cells) {
  for (size_t c = 0; c < n_cells; ++c) {
    GuideRegionDev d{0u, 0u, 0.0f, u2f(grid[c])};
    if (grid[c] < nr) {
      d = regions[grid[c]];
      d.pad = u2f(grid[c]);
    }
    cells[c] = d;
  }
}

}  // namespace pt
