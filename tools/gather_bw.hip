// gather_bw.hip -- how fast can MI355X fetch random 128-byte lines?  The ceiling for BVH traversal of a scene
// that does not fit in the caches: every lane walks a dependent chain of random lines (the next line's index
// comes from the data just loaded, as a child pointer does), reading 80 of the line's 128 bytes as five
// 16-byte words, with as many waves resident as the traversal kernels have.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_bw.hip -o tools/gather_bw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
  return x;
}
__global__ void __launch_bounds__(256) k_fill(uint4* p, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t h = mix((uint32_t)i * 2654435761u + 12345u);
    p[i] = uint4{h, mix(h), mix(h + 1u), mix(h + 2u)};
  }
}
template <int WORDS>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ p, uint32_t nlines, int steps, uint32_t* out, int lds_pad) {
  extern __shared__ uint32_t pad[];
  if (lds_pad < 0) pad[threadIdx.x] = 0;  // keeps the dynamic LDS (which sets the occupancy) alive
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) % nlines;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    const uint4* line = p + (size_t)idx * 8u;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < WORDS; ++k) { const uint4 q = line[k]; v ^= q.x ^ q.y ^ q.z ^ q.w; }
    acc += v;
    idx = (uint32_t)(((uint64_t)mix(v + (uint32_t)s) * nlines) >> 32);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
  const int steps = 64;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out; CK(hipMalloc(&out, 4));
  // one argument: a single footprint in MB, 7 waves per SIMD, one launch per word count (for counter calibration)
  const bool single = argc > 1;
  std::vector<size_t> sizes_mb = {2, 16, 128, 1152, 8192};
  if (single) sizes_mb = {(size_t)atoll(argv[1])};
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20;
    uint4* p; CK(hipMalloc(&p, bytes));
    k_fill<<<cus * 8, 256>>>(p, bytes / 16);
    CK(hipDeviceSynchronize());
    const uint32_t nlines = (uint32_t)(bytes / 128);
    for (int waves : {2, 4, 7, 8}) {           // waves per SIMD: blocks of 256 threads per CU
      if (single && waves != 7) continue;
      const size_t lds = (160u * 1024u / waves) & ~255u;   // as much LDS as lets exactly `waves` blocks fit
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int words : {5, 8}) {
        float best = 1e30f;
        for (int rep = 0; rep < (single ? 1 : 3); ++rep) {
          CK(hipEventRecord(a));
          if (words == 5) k_chase<5><<<cus * waves, 256, lds>>>(p, nlines, steps, out, 0);
          else k_chase<8><<<cus * waves, 256, lds>>>(p, nlines, steps, out, 0);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const double lines = (double)cus * waves * 256 * steps;
        printf("footprint %5zu MB  %d waves/SIMD  %d x 16 B of each line: %7.3f ms  %6.2f G lines/s  %6.2f TB/s of 128-B lines  (%.0f ns per step per lane; %.0f lines = %.1f MB per launch)\n",
               mb, waves, words, best, lines / best / 1e6, lines * 128 / best / 1e9, best * 1e6 / steps, lines, lines * 128 / 1e6);
      }
    }
    CK(hipFree(p));
  }
  return 0;
}
