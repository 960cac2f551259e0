// pgl_stub.cpp -- stand-in for the reference's OpenPGL bridge (optix/openpgl_bridge.cpp).
//
// NOT OpenPGL.  Intel's Open Path Guiding Library is an un-vendored dependency of the
// reference (optix/CMakeLists.txt:19-25) and is not available to this build, so this file
// keeps the bridge's C ABI (optix/openpgl_bridge.h:7-33) and ownership rules -- opaque
// handles, library-owned snapshot arrays valid until the next snapshot -- and trains a much
// simpler field on the host: a fixed grid of regions over the scene bounds, each holding up
// to six von Mises-Fisher lobes fitted by weighted moments.  What it shares with the bridge:
//   * sample weight = mean of the RGB weight, pdf 1, distance 0 (openpgl_bridge.cpp:47-58)
//   * per-lobe RGB from the sample colours assigned to the lobe with the largest
//     dot(direction, mu) in the sample's region (openpgl_bridge.cpp:95-119).  The bridge stores
//     their SUM, and the integrator multiplies the path throughput by it (optix_device.cu:629, 653-654):
//     with real training data that multiplies radiance by the number of samples per lobe.  (It never
//     shows in the reference because its training ring stays empty.)  Here the MEAN colour is stored.
//   * snapshot layout pgl_region / pgl_lobe with lobe_ofs / lobe_num (126-193)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_guiding.h"

namespace {

constexpr int kLobes = 6;

struct StubDevice {
  int threads;
};
struct StubSample {
  float pos[3], dir[3], weight, rgb[3];
  int is_delta;
};
struct StubSamples {
  std::vector<StubSample> items;
};
struct StubLobe {
  double r[3] = {0, 0, 0};  // weighted sum of directions
  double w = 0;             // sum of weights
  double rgb[3] = {0, 0, 0};
  double count = 0;         // samples assigned
  float mu[3] = {0, 0, 1};  // current mean direction used for assignment
};
struct StubRegion {
  StubLobe lobes[kLobes];
};
struct StubField {
  float bmin[3], bmax[3];
  int res[3];
  std::vector<StubRegion> regions;
  bool trained = false;
};

std::vector<pgl_region> g_regions;
std::vector<pgl_lobe> g_lobes;

int field_res() {
  const char* v = getenv("PT_PGL_STUB_RES");
  int r = v ? atoi(v) : 4;
  return r < 1 ? 1 : (r > 64 ? 64 : r);
}

// the assignment directions follow the accumulated moments
void refresh_axes(StubField& f) {
  for (StubRegion& reg : f.regions)
    for (StubLobe& lb : reg.lobes) {
      const double len = std::sqrt(lb.r[0] * lb.r[0] + lb.r[1] * lb.r[1] + lb.r[2] * lb.r[2]);
      if (lb.w > 0 && len > 1e-12)
        for (int k = 0; k < 3; ++k) lb.mu[k] = (float)(lb.r[k] / len);
    }
}

}  // namespace

extern "C" {

pgl_dev_t pgl_dev_create(int num_threads) { return new StubDevice{num_threads}; }
void pgl_dev_destroy(pgl_dev_t dev) { delete static_cast<StubDevice*>(dev); }

pgl_field_t pgl_field_create(pgl_dev_t, const float world_min[3], const float world_max[3]) {
  StubField* f = new StubField();
  const int r = field_res();
  for (int k = 0; k < 3; ++k) {
    f->bmin[k] = world_min[k];
    f->bmax[k] = world_max[k];
    f->res[k] = r;
  }
  f->regions.resize((size_t)r * r * r);
  static const float axes[kLobes][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (StubRegion& reg : f->regions)
    for (int l = 0; l < kLobes; ++l) memcpy(reg.lobes[l].mu, axes[l], sizeof(float) * 3);
  return f;
}
void pgl_field_destroy(pgl_field_t field) { delete static_cast<StubField*>(field); }

pgl_samples_t pgl_samples_create(void) { return new StubSamples(); }
void pgl_samples_destroy(pgl_samples_t s) { delete static_cast<StubSamples*>(s); }

void pgl_samples_add_surface(pgl_samples_t handle, const float pos[3], const float dir_in[3], const float weight_rgb[3],
                             int is_delta) {
  StubSamples* s = static_cast<StubSamples*>(handle);
  StubSample it;
  memcpy(it.pos, pos, sizeof(float) * 3);
  memcpy(it.dir, dir_in, sizeof(float) * 3);
  memcpy(it.rgb, weight_rgb, sizeof(float) * 3);
  it.weight = (weight_rgb[0] + weight_rgb[1] + weight_rgb[2]) / 3.0f;
  it.is_delta = is_delta;
  s->items.push_back(it);
}

// Many records at once, in the layout guiding_map_train_samples returns (src/guiding.rs:196-219 adds them one
// by one): what a host uses that has gathered the rings of several GPUs.
void pt_pgl_samples_add_batch(pgl_samples_t handle, const TrainSample* records, uint32_t count) {
  if (!handle || !records) return;
  for (uint32_t i = 0; i < count; ++i)
    pgl_samples_add_surface(handle, records[i].position, records[i].dir_in, records[i].contrib, (int)records[i].is_delta);
}

void pgl_field_update(pgl_field_t field_handle, pgl_samples_t samples_handle) {
  StubField* f = static_cast<StubField*>(field_handle);
  StubSamples* s = static_cast<StubSamples*>(samples_handle);
  if (!f || !s) return;
  for (const StubSample& it : s->items) {
    int c[3];
    bool inside = true;
    for (int k = 0; k < 3; ++k) {
      const float ext = f->bmax[k] - f->bmin[k];
      if (!(it.pos[k] >= f->bmin[k] && it.pos[k] <= f->bmax[k]) || !(ext > 0.f)) { inside = false; break; }  // a NaN is outside
      c[k] = (int)((it.pos[k] - f->bmin[k]) / ext * (float)f->res[k]);
      if (c[k] >= f->res[k]) c[k] = f->res[k] - 1;
    }
    if (!inside || !pt::train_weight_usable(it.weight, it.rgb)) continue;  // the same records the binning kernel uses
    StubRegion& reg = f->regions[((size_t)c[2] * f->res[1] + c[1]) * f->res[0] + c[0]];
    int best = 0;
    float best_d = -2.f;
    for (int l = 0; l < kLobes; ++l) {
      const float* mu = reg.lobes[l].mu;
      const float d = it.dir[0] * mu[0] + it.dir[1] * mu[1] + it.dir[2] * mu[2];
      if (d > best_d) { best_d = d; best = l; }
    }
    StubLobe& lb = reg.lobes[best];
    for (int k = 0; k < 3; ++k) {
      lb.r[k] += (double)it.weight * it.dir[k];
      lb.rgb[k] += it.rgb[k];
    }
    lb.w += it.weight;
    lb.count += 1.0;
  }
  refresh_axes(*f);
  f->trained = true;
  s->items.clear();  // openpgl_bridge.cpp:122-123
}

// ---- training from statistics binned on the GPU (no counterpart in the reference, which moves the whole ring to
// the host and adds its records one call at a time, src/guiding.rs:194-220).  pgl_field_update only ever needs,
// per region and lobe, the sums over the samples assigned to it; those sums are taken on the device
// (pt_kernels.hip: k_bin_train) as 64-bit fixed-point integers, so they do not depend on the order of addition and
// the tables of several GPUs add up exactly.
//   bins[(region * PT_PGL_BIN_LOBES + lobe) * PT_PGL_BIN_WORDS + k]:
//   k = 0..2 sum of weight * dir_in, 3 sum of weight, 4..6 sum of the sample colour (all x 2^24), 7 number of samples
int pt_pgl_field_layout(pgl_field_t field_handle, float bmin[3], float bmax[3], int res[3], float* lobe_axes, uint32_t* regions) {
  StubField* f = static_cast<StubField*>(field_handle);
  if (!f) return -1;
  for (int k = 0; k < 3; ++k) {
    if (bmin) bmin[k] = f->bmin[k];
    if (bmax) bmax[k] = f->bmax[k];
    if (res) res[k] = f->res[k];
  }
  if (regions) *regions = (uint32_t)f->regions.size();
  if (lobe_axes)  // the directions samples are assigned by, [region][lobe][3]
    for (size_t r = 0; r < f->regions.size(); ++r)
      for (int l = 0; l < kLobes; ++l) memcpy(lobe_axes + (r * kLobes + l) * 3, f->regions[r].lobes[l].mu, sizeof(float) * 3);
  return 0;
}

void pt_pgl_field_update_binned(pgl_field_t field_handle, const long long* bins) {
  StubField* f = static_cast<StubField*>(field_handle);
  if (!f || !bins) return;
  static_assert(PT_PGL_BIN_LOBES == kLobes, "bin table and field disagree on the lobes per region");
  const double inv = 1.0 / (double)(1ll << PT_PGL_BIN_SHIFT);
  for (size_t r = 0; r < f->regions.size(); ++r)
    for (int l = 0; l < kLobes; ++l) {
      const long long* b = bins + (r * kLobes + l) * PT_PGL_BIN_WORDS;
      StubLobe& lb = f->regions[r].lobes[l];
      for (int k = 0; k < 3; ++k) {
        lb.r[k] += (double)b[k] * inv;
        lb.rgb[k] += (double)b[4 + k] * inv;
      }
      lb.w += (double)b[3] * inv;
      lb.count += (double)b[7];
    }
  refresh_axes(*f);
  f->trained = true;
}

void pgl_field_snapshot(pgl_field_t field_handle, const struct pgl_region** regions, uint32_t* region_count,
                        const struct pgl_lobe** lobes, uint32_t* lobe_count) {
  g_regions.clear();
  g_lobes.clear();
  StubField* f = static_cast<StubField*>(field_handle);
  if (f) {
    for (int z = 0; z < f->res[2]; ++z)
      for (int y = 0; y < f->res[1]; ++y)
        for (int x = 0; x < f->res[0]; ++x) {
          const StubRegion& reg = f->regions[((size_t)z * f->res[1] + y) * f->res[0] + x];
          pgl_region out;
          const int c[3] = {x, y, z};
          for (int k = 0; k < 3; ++k) {
            const float step = (f->bmax[k] - f->bmin[k]) / (float)f->res[k];
            out.bmin[k] = f->bmin[k] + step * (float)c[k];
            out.bmax[k] = f->bmin[k] + step * (float)(c[k] + 1);
          }
          out.lobe_ofs = (uint32_t)g_lobes.size();
          double wsum = 0;
          for (const StubLobe& lb : reg.lobes) wsum += lb.w;
          uint32_t n = 0;
          for (const StubLobe& lb : reg.lobes) {
            if (!(lb.w > 0)) continue;
            const double len = std::sqrt(lb.r[0] * lb.r[0] + lb.r[1] * lb.r[1] + lb.r[2] * lb.r[2]);
            double rbar = len / lb.w;
            if (rbar > 0.999) rbar = 0.999;
            double kappa = rbar * (3.0 - rbar * rbar) / (1.0 - rbar * rbar);  // Banerjee et al. 2005
            if (kappa < 0.01) kappa = 0.01;
            if (kappa > 100.0) kappa = 100.0;
            pgl_lobe l;
            for (int k = 0; k < 3; ++k) {
              l.mu[k] = lb.mu[k];
              l.rgb[k] = (float)(lb.rgb[k] / std::max(lb.count, 1.0));
            }
            l.kappa = (float)kappa;
            l.weight = (float)(lb.w / wsum);
            g_lobes.push_back(l);
            n++;
          }
          out.lobe_num = n;
          g_regions.push_back(out);
        }
  }
  if (regions) *regions = g_regions.data();
  if (region_count) *region_count = (uint32_t)g_regions.size();
  if (lobes) *lobes = g_lobes.data();
  if (lobe_count) *lobe_count = (uint32_t)g_lobes.size();
}

}  // extern "C"
