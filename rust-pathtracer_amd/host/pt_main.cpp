// pt_main.cpp -- C++ host driver, a restatement of the reference's Rust binary
// (src/main.rs:437-564): same default parameters, same call sequence against the same C
// ABI, same output files (pt.exr, pt_bayer.exr) and the same progress lines on stdout.
// The Rust toolchain is not available to this build; a Rust host binds the identical
// symbols through the extern "C" block it already has (src/main.rs:9-32).
//
// Differences from main.rs: the render parameters, which main.rs hard-codes
// (src/main.rs:478-484), can be overridden on the command line; the demosaic runs on the
// GPU through pt_demosaic_f32 (--cpu-demosaic is not offered: there is no CPU path); ray
// throughput is printed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "guiding_host.h"
#include "pt_api.h"

static bool arg_value(const char* arg, const char* key, std::string& out) {
  const size_t n = strlen(key);
  if (strncmp(arg, key, n) == 0 && arg[n] == '=') { out = arg + n + 1; return true; }
  return false;
}

int main(int argc, char** argv) {
  // src/main.rs:478-484
  int w = 1920, h = 1440, spp = 512, max_depth = 32, pattern = 0, min_adaptive_samples = 32;
  float noise_threshold = 0.01f;  // 0 disables adaptive sampling
  bool fast_demosaic = false;     // Cargo feature `fast-demosaic` (src/main.rs:403-415)
  // src/main.rs:439-465
  bool guiding = false;
  uint32_t update_interval = 1;
  int grid_res[3] = {1, 1, 1};
  std::string out_rgb = "pt.exr", out_bayer = "pt_bayer.exr", out_png, v;  // --png=FILE: an 8-bit sRGB copy of the RGB render
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    if (!strcmp(a, "--guiding")) guiding = true;
    else if (arg_value(a, "--guide-update-interval", v)) update_interval = (uint32_t)atoi(v.c_str());
    else if (arg_value(a, "--guide-grid", v)) {
      int x, y, z;
      if (sscanf(v.c_str(), "%d%*[xX,]%d%*[xX,]%d", &x, &y, &z) == 3) { grid_res[0] = x; grid_res[1] = y; grid_res[2] = z; }
      else if (sscanf(v.c_str(), "%d", &x) == 1) { grid_res[0] = grid_res[1] = grid_res[2] = x; }
    }
    else if (arg_value(a, "--width", v)) w = atoi(v.c_str());
    else if (arg_value(a, "--height", v)) h = atoi(v.c_str());
    else if (arg_value(a, "--spp", v)) spp = atoi(v.c_str());
    else if (arg_value(a, "--max-depth", v)) max_depth = atoi(v.c_str());
    else if (arg_value(a, "--noise-threshold", v)) noise_threshold = (float)atof(v.c_str());
    else if (arg_value(a, "--min-adaptive", v)) min_adaptive_samples = atoi(v.c_str());
    else if (arg_value(a, "--pattern", v)) pattern = atoi(v.c_str());
    else if (arg_value(a, "--gpus", v)) pt_set_device_count(atoi(v.c_str()));  // sample shards, one GPU each
    else if (arg_value(a, "--demosaic", v)) fast_demosaic = (v == "bilinear");
    else if (arg_value(a, "--mode", v)) pt_set_mode(v == "strict" ? PT_MODE_STRICT : PT_MODE_CONVERGE);
    else if (arg_value(a, "--pipeline", v)) pt_set_pipeline(v == "staged" ? PT_PIPELINE_STAGED : PT_PIPELINE_FUSED);
    else if (arg_value(a, "--scene", v)) {
      int n = 0, sub = 3;
      if (v == "cornell") pt_scene_cornell();
      else if (sscanf(v.c_str(), "instanced:%d:%d", &n, &sub) >= 1) pt_scene_synthetic_instanced(42u, n, sub);  // the same picture as instances
      else if (sscanf(v.c_str(), "synthetic:%d:%d", &n, &sub) >= 1) pt_scene_synthetic(42u, n, sub);
      else if (pt_scene_load_obj(v.c_str()) < 0) return 2;  // anything else is the path of an OBJ file
    }
    else if (arg_value(a, "--out", v)) out_rgb = v;
    else if (arg_value(a, "--out-bayer", v)) out_bayer = v;
    else if (arg_value(a, "--png", v)) out_png = v;
    else { fprintf(stderr, "unknown argument %s\n", a); return 2; }
  }
  const float scene_min[3] = {-1.0f, -1.0f, 0.0f}, scene_max[3] = {1.0f, 1.0f, 2.0f};
  const uint32_t train_cap = guiding ? (1u << 18) : 0u;  // src/main.rs:468 (ring only when guiding is on)
  host::GuidingState guiding_state(guiding, scene_min, scene_max, grid_res, update_interval, train_cap);

  const size_t rgb_bytes = (size_t)w * h * 3 * sizeof(float), bayer_bytes = (size_t)w * h * sizeof(float);
  float* rgb = static_cast<float*>(optix_alloc_host(rgb_bytes));      // src/main.rs:489-493
  float* bayer = static_cast<float*>(optix_alloc_host(bayer_bytes));

  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };

  const auto t0 = clk::now();
  if (optix_render_rgb(w, h, spp, max_depth, noise_threshold, min_adaptive_samples, rgb) != 0) abort();  // :495-507
  const double t_rgb = secs(t0, clk::now());
  guiding_state.process_batch();  // :510

  const auto t1 = clk::now();
  if (optix_render_bayer_f32(w, h, spp, max_depth, noise_threshold, min_adaptive_samples, pattern, bayer) != 0) abort();
  const double t_raw = secs(t1, clk::now());
  guiding_state.process_batch();  // :528
  optix_synchronize();            // :530-532

  const float exposure = 1.0f;
  printf("Exposure multiplier=%.6f\n", exposure);
  if (pt_write_exr(out_rgb.c_str(), w, h, rgb, exposure) == 0) printf("Saved %s (render %.3fs)\n", out_rgb.c_str(), t_rgb);
  else fprintf(stderr, "Failed to save %s\n", out_rgb.c_str());
  if (!out_png.empty()) {
    if (pt_write_png(out_png.c_str(), w, h, rgb, exposure, 8) == 0) printf("Saved %s\n", out_png.c_str());
    else fprintf(stderr, "Failed to save %s\n", out_png.c_str());
  }

  const auto t2 = clk::now();
  std::vector<float> demosaiced((size_t)w * h * 3);
  if (pt_demosaic_f32(bayer, w, h, pattern, fast_demosaic ? 1 : 0, demosaiced.data()) != 0) abort();  // :541-543
  const double t_demosaic = secs(t2, clk::now());
  printf("Bayer render: %.3fs\n", t_raw);
  printf("Demosaic (%s): %.3fs\n", fast_demosaic ? "bilinear" : "AMaZE", t_demosaic);
  if (pt_write_exr(out_bayer.c_str(), w, h, demosaiced.data(), exposure) == 0) printf("Saved %s\n", out_bayer.c_str());
  else fprintf(stderr, "Failed to save %s\n", out_bayer.c_str());

  uint64_t rr = 0, rs = 0;
  pt_get_ray_counts(&rr, &rs);
  printf("Rays: %llu radiance + %llu shadow, %.1f Mrays/s over both renders\n", (unsigned long long)rr,
         (unsigned long long)rs, (double)(rr + rs) / (t_rgb + t_raw) * 1e-6);

  optix_free_host(rgb);  // :560-563
  optix_free_host(bayer);
  return 0;
}
