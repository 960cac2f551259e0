// valu_issue.hip -- how many SIMD cycles does a wave64 v_fma_f32 take on gfx950, as a function of waves per SIMD?
// Each wave runs kIter x 32 independent scalar (not packed: build with -fno-slp-vectorize) FMAs, 8 chains x 4
// unrolled.  One workgroup of 256 x (waves per SIMD) threads per CU -- a dynamic LDS request of more than half
// the CU's 160 KB keeps a second workgroup off the CU -- so every SIMD hosts exactly that many waves (8 per SIMD:
// two workgroups of 1024 threads with 64 KB each).  Prints TFLOP/s and, from s_memtime, the SIMD cycles one
// wave-instruction takes: wave lifetime / (waves per SIMD x instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIter = 65536;

extern __shared__ float g_pad[];
typedef float v2f __attribute__((ext_vector_type(2)));
// the same loop with v_pk_fma_f32: two FMAs per lane per instruction
__global__ void __launch_bounds__(1024) k_pk_fma(float* out, unsigned long long* cycles, float a, float b) {
  if (a == 12345.f) g_pad[threadIdx.x] = b;
  const v2f va = {a, a}, vb = {b, b};
  v2f x[8];
  for (int k = 0; k < 8; ++k) x[k] = v2f{(float)threadIdx.x + k, (float)threadIdx.x - k};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIter; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], va, vb);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int k = 0; k < 8; ++k) acc += x[k].x + x[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0;
    cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0;
  }
}
__global__ void __launch_bounds__(1024) k_fma(float* out, unsigned long long* cycles, float a, float b) {
  if (a == 12345.f) g_pad[threadIdx.x] = b;  // keep the LDS request
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIter; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if ((threadIdx.x & 63) == 0) {
    cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0;      // shader clock ticks
    cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0;  // 100 MHz ticks
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  float* out;
  unsigned long long* cyc;
  const int max_blocks = cus * 8;
  CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, (size_t)max_blocks * 8 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int packed = 0; packed < 2; ++packed) {
    auto kern = packed ? k_pk_fma : k_fma;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int wps : {1, 2, 4, 8}) {
      const int per_cu = wps == 8 ? 2 : 1, threads = 256 * wps / per_cu, blocks = cus * per_cu;
      const size_t lds = wps == 8 ? 64 * 1024 : 96 * 1024;
      kern<<<blocks, threads, lds>>>(out, cyc, 1.0001f, 0.5f);  // warm-up
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) kern<<<blocks, threads, lds>>>(out, cyc, 1.0001f, 0.5f);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      const size_t waves = (size_t)blocks * threads / 64;
      std::vector<unsigned long long> h(2 * waves);
      CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double ticks = 0, real = 0;
      for (size_t i = 0; i < waves; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
      ticks /= waves;
      real /= waves;
      const double insts = (double)kIter * 32;                      // FMA instructions per wave
      const double flops = (double)waves * 64 * insts * 2 * (packed ? 2 : 1);
      printf("%s waves/SIMD %d: %.3f ms, %.1f TFLOP/s, in-kernel clock %.2f GHz, %.2f shader cycles per wave-instruction per SIMD\n",
             packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, flops / (ms * 1e-3) / 1e12, ticks / real * 0.1, ticks / (wps * insts));
    }
  }
  return 0;
}
