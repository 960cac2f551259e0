// guiding_host.h -- host-side guiding loop, C++ restatement of src/guiding.rs.
//
// RAII wrappers over the pgl_* handles (guiding.rs:71-143) and GuidingState with new /
// collect_train_samples / process_batch (guiding.rs:146-247), calling the same C ABI.
#pragma once
#include <cstdint>

#include "pt_api.h"

namespace host {

struct Device {  // guiding.rs:71-93
  pgl_dev_t p = nullptr;
  explicit Device(int threads) : p(pgl_dev_create(threads)) {}
  ~Device() { if (p) pgl_dev_destroy(p); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
};
struct Field {  // guiding.rs:96-118
  pgl_field_t p = nullptr;
  Field(const Device& d, const float mn[3], const float mx[3]) : p(pgl_field_create(d.p, mn, mx)) {}
  ~Field() { if (p) pgl_field_destroy(p); }
  Field(const Field&) = delete;
  Field& operator=(const Field&) = delete;
};
struct Samples {  // guiding.rs:121-143
  pgl_samples_t p = nullptr;
  Samples() : p(pgl_samples_create()) {}
  ~Samples() { if (p) pgl_samples_destroy(p); }
  Samples(const Samples&) = delete;
  Samples& operator=(const Samples&) = delete;
};

class GuidingState {  // guiding.rs:146-248
 public:
  GuidingState(bool enabled, const float scene_min[3], const float scene_max[3], const int grid_res[3],
               uint32_t update_interval, uint32_t train_capacity)
      : enabled_(enabled), update_interval_(update_interval ? update_interval : 1) {
    guiding_set_enabled(enabled ? 1 : 0);
    guiding_set_train_buffer_size(train_capacity);
    guiding_set_grid_res(grid_res[0], grid_res[1], grid_res[2]);
    if (enabled) {
      dev_ = new Device(0);
      if (dev_->p) field_ = new Field(*dev_, scene_min, scene_max);
      samples_ = new Samples();
    }
  }
  ~GuidingState() {
    delete samples_;
    delete field_;
    delete dev_;
  }
  bool enabled() const { return enabled_; }

  void process_batch() {  // guiding.rs:222-247
    if (!enabled_) return;
    collect_train_samples();
    frame_idx_++;
    if (frame_idx_ % update_interval_ == 0 && field_ && field_->p && samples_ && samples_->p) {
      pgl_field_update(field_->p, samples_->p);
      const pgl_region* regions = nullptr;
      const pgl_lobe* lobes = nullptr;
      uint32_t region_count = 0, lobe_count = 0;
      pgl_field_snapshot(field_->p, &regions, &region_count, &lobes, &lobe_count);
      guiding_upload_snapshot(regions, region_count, lobes, lobe_count);
    }
  }

 private:
  void collect_train_samples() {  // guiding.rs:194-220
    if (!enabled_ || !samples_ || !samples_->p) return;
    const TrainSample* ptr = nullptr;
    uint32_t count = 0;
    guiding_map_train_samples(&ptr, &count);
    if (ptr && count > 0)
      for (uint32_t i = 0; i < count; ++i)
        pgl_samples_add_surface(samples_->p, ptr[i].position, ptr[i].dir_in, ptr[i].contrib, (int)ptr[i].is_delta);
    guiding_reset_train_write_idx();
  }
  bool enabled_;
  Device* dev_ = nullptr;
  Field* field_ = nullptr;
  Samples* samples_ = nullptr;
  uint32_t update_interval_;
  uint32_t frame_idx_ = 0;
};

}  // namespace host
