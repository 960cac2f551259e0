// pt_types.h -- vector type, math wrappers and shared POD layouts for the gfx950 kernels.
//
// The integrator's device functions are written against this header only, so the same
// sources also compile with a host C++ compiler for CPU-side debugging of the device
// logic (tests/hostcheck); the shipped library never runs them on the CPU.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#define PT_D __device__ __forceinline__
#else
#define PT_HD inline
#define PT_D inline
#endif

#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define PT_ON_DEVICE 1
#else
#define PT_ON_DEVICE 0
#endif

// Section counters for the profiling build (-DPT_PROFILE): how many times a wave executes a
// section and with how many active lanes.  Compiled out of the shipped library.
#if defined(PT_PROFILE) && defined(__HIPCC__)
__shared__ unsigned int g_prof_lds[64];
#endif
#if defined(PT_PROFILE) && PT_ON_DEVICE
#define PT_PROF(k)                                                                       \
  do {                                                                                   \
    const unsigned long long m_ = __ballot(1);                                           \
    if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) {                          \
      atomicAdd(&g_prof_lds[2 * (k)], 1u);                                               \
      atomicAdd(&g_prof_lds[2 * (k) + 1], (unsigned)__popcll(m_));                       \
    }                                                                                    \
  } while (0)
// Time sections: shader-clock ticks a wave spends between PT_TIC and PT_TOC(k), summed over waves into
// g_prof_lds[40 + k].  (With several waves per SIMD a wave's ticks include the turns of its neighbours, so read
// the sums as shares of the total, not as instruction counts.)
#define PT_TIC(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#define PT_TOC(name, k)                                                                  \
  do {                                                                                   \
    const unsigned long long d_ = __builtin_amdgcn_s_memtime() - name;                   \
    const unsigned long long m_ = __ballot(1);                                           \
    if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) atomicAdd(&g_prof_lds[40 + (k)], (unsigned)d_); \
  } while (0)
#else
#define PT_PROF(k) do { } while (0)
#define PT_TIC(name) do { } while (0)
#define PT_TOC(name, k) do { } while (0)
#endif
enum { PT_T_LOOP = 0, PT_T_REGEN, PT_T_TRACE, PT_T_SHADE, PT_T_ENQUEUE, PT_T_SHADOW, PT_T_EPILOGUE, PT_T_NODE, PT_T_TRI, PT_T_COUNT };
// profiling sections
enum { PF_BOUNCE = 0, PF_REGEN, PF_NODE, PF_TRI, PF_SHADE, PF_GLASS, PF_OPAQUE, PF_NEE, PF_SPEC, PF_DIFFUSE,
       PF_SH_NODE, PF_SH_TRI, PF_SHADOW, PF_POP, PF_SH_POP, PF_EMIS_MIS, PF_RR, PF_FALLBACK, PF_COUNT };

namespace pt {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 1.0f / kPi;

struct f3 {
  float x, y, z;
};
PT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_HD f3 mk3(float s) { return f3{s, s, s}; }
PT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
PT_HD f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
PT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD f3& operator+=(f3& a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_HD float comp(f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
PT_HD float max3(f3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }

// ---- fast scalar math: one hardware instruction each on gfx950 (v_rcp/v_sqrt/v_rsq/
// v_sin/v_cos/v_exp/v_log, all about 1 ulp), libm on the host.  The reference itself is
// built with --use_fast_math (optix/CMakeLists.txt:60,78), so approximate transcendental
// rates are the intended precision class.
PT_HD float rcp(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
PT_HD float fsqrt(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
PT_HD float rsqrt(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
// sin/cos of 2*pi*u: v_sin_f32/v_cos_f32 take their argument in revolutions.
PT_HD float sin2pi(float u) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_sinf(u);
#else
  return sinf(2.0f * kPi * u);
#endif
}
PT_HD float cos2pi(float u) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_cosf(u);
#else
  return cosf(2.0f * kPi * u);
#endif
}
PT_HD float fexp(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
#else
  return expf(x);
#endif
}
PT_HD float fexp2(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_exp2f(x);
#else
  return exp2f(x);
#endif
}
PT_HD float flog(float x) {
#if PT_ON_DEVICE
  return __builtin_amdgcn_logf(x) * 0.693147180559945309f;
#else
  return logf(x);
#endif
}
PT_HD float pow5(float x) {
  float x2 = x * x;
  return x2 * x2 * x;
}
PT_HD float fract_pos(float x) { return x - floorf(x); }  // == fmodf(x,1) for x >= 0
PT_HD float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

PT_HD float length(f3 v) { return fsqrt(dot(v, v)); }
// optix_device.cu:122-125
PT_HD f3 normalize_chk(f3 v) {
  float l2 = dot(v, v);
  return l2 > 0.f ? v * rsqrt(l2) : mk3(0.f);
}
// optix_device.cu:137-139
// one reciprocal square root in place of a square root and a reciprocal (the two forms agree to an ulp; the guard
// only matters for vectors shorter than 1e-19, which none of the callers makes)
PT_HD f3 normalize3(f3 v) { return v * rsqrt(fmaxf(dot(v, v), 1e-38f)); }

PT_HD uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
PT_HD float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---------------------------------------------------------------------------------
// Shared POD layouts (host builds them, kernels read them).

// 16-byte vector for coalesced dwordx4 traffic.
struct alignas(16) q4 {
  float x, y, z, w;
};
struct alignas(16) u4 {
  uint32_t x, y, z, w;
};

// Compressed 8-wide BVH node, 80 bytes padded to a 128-byte cache line so that a visit touches one line
// (layout after Ylitie, Karras, Laine 2017):
// child boxes are 8-bit offsets on a per-node grid p + q * 2^e.
struct alignas(128) BvhNode {
  float px, py, pz;        // grid origin = node box minimum
  uint8_t ex, ey, ez;      // biased float exponents of the grid step per axis
  uint8_t imask;           // bit s set: slot s holds an internal child
  uint32_t child_base;     // index of the first internal child (children contiguous in slot order)
  uint32_t tri_base;       // index of the first triangle referenced by this node's leaf slots
  uint8_t meta[8];         // per slot: 0 empty; internal (1<<5)|(24+slot); leaf (unary count<<5)|first triangle offset
  uint8_t qlox[8], qloy[8], qloz[8];
  uint8_t qhix[8], qhiy[8], qhiz[8];
  uint8_t pad_[48];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");

// Per-triangle shading record, indexed by the scene's primitive index (96 bytes).
struct alignas(16) ShadeRec {
  float v0[3]; float roughness;
  float v1[3]; float ior;
  float v2[3]; float transmission;
  float ng[3]; uint32_t home_node;   // BVH node whose leaf slots hold this primitive
  float base[3]; uint32_t home_mask;  // triangles of that node in this primitive's plane (bit = offset in the node)
  float emission[3]; uint32_t corner_order;  // input corner held by v0 / v1 / v2, 2 bits each (0x24: as given)
};
static_assert(sizeof(ShadeRec) == 96, "ShadeRec must be 96 bytes");

// optix/guiding_gpu.cuh:8-17
struct GuideRegion { float bmin[3]; float bmax[3]; uint32_t lobe_ofs, lobe_num; };
struct GuideLobe { float mu[3]; float kappa; float weight; float rgb[3]; };
// optix/guiding_gpu.cuh:24-29
struct TrainRec { float position[3]; float dir_in[3]; float contrib[3]; uint32_t is_delta; };
// Ring slots are overwritten without synchronisation, so two vertices written to one slot at the same
// moment can leave a mix of both.  A record in the ring therefore carries, in bits 1..31 of is_delta, a
// digest of its nine payload words (bit 0 is the flag itself); the host drops records whose digest does
// not match.  The digest word of a captured record has bit 0 clear, so it never equals the "never
// written" marker 0xffffffff.
PT_HD uint32_t train_digest(const TrainRec& r) {
  const float w[9] = {r.position[0], r.position[1], r.position[2], r.dir_in[0], r.dir_in[1], r.dir_in[2],
                      r.contrib[0],  r.contrib[1],  r.contrib[2]};
  uint32_t h = 0x9E3779B9u;
  for (int k = 0; k < 9; ++k) {
    h = (h ^ f2u(w[k])) * 0x85EBCA6Bu;
    h ^= h >> 15;
  }
  return h << 1;
}

// The field as the kernels read it (built from the snapshot at upload time, pt_guiding.h: make_guide_lobes).
// A lobe is four 16-byte words with everything that does not depend on the query already worked out: the running
// weight sum for the lobe choice, weight x normalisation and kappa x log2(e) for the mixture pdf, e^(-2 kappa) and
// 1 / kappa for the sampler, the unit axis.  A region is its lobe range and total weight.
struct alignas(16) GuideLobeDev {
  float mu[3]; float kappa;          // as given (the pdf of guiding_gpu.cuh:80-84 uses mu unnormalised)
  float cdf, c, k2, term;            // sum of weights up to and including this lobe; weight * kappa / (2 pi (1 - e^-2k))
                                     // (weight / 4 pi for kappa <= 1e-4); kappa * log2(e); e^(-2 kappa)
  float rgb[3]; float inv_kappa;
  float axis[3]; float weight;       // mu normalised
};
struct alignas(16) GuideRegionDev {
  uint32_t lobe_ofs, lobe_num;
  float sum, pad;                    // total weight of the region's lobes
};
static_assert(sizeof(GuideLobeDev) == 64 && sizeof(GuideRegionDev) == 16, "device guiding records");

// Device view of the guiding field (optix/guiding_gpu.cuh:18-22).
struct GuideView {
  const GuideRegionDev* regions;
  const GuideLobeDev* lobes;
  const uint32_t* grid;          // region id per cell, 0xFFFFFFFF = none (guiding_gpu.cuh:35-45)
  // the same grid with each cell's region record in place (pad = region id as bits; lobe_num = 0 for none):
  // one load instead of two dependent ones on the shading path
  const GuideRegionDev* cells;
  int res[3];
  float gmin[3];
  float cell[3];
  float inv_cell[3];  // 1 / cell (0 where the field is flat along the axis)
};

struct SceneView {
  const BvhNode* nodes;
  // Small trees also exist as wide nodes (pt_bvh.h): wide_src in device memory (null for large trees), and
  // home_wide = per primitive {node, item bits} of where it lives in the wide tree.  Kernels that stage the
  // scene in LDS point `wide` (and home_wide) at their copy; everywhere else `wide` is null and traversal
  // reads the packed nodes.
  const u4* wide;
  const u4* wide_src;
  const uint32_t* home_wide;
  uint32_t wide_groups;      // slot groups to test per wide node: the most any node of the tree uses
  uint32_t wide_one_node;    // the wide tree is its root alone (every leaf slot in one node)
  // 4 x q4 per leaf item (a triangle, or two forming a parallelogram) in BVH leaf order (see the record
  // layout in pt_bvh.h): v0 and the rows that map (p - v0) to the barycentrics of v1, v2 and to the distance
  // from the plane
  const q4* tris;
  const ShadeRec* shade;     // per primitive
  // Scenes made of instances (else null): 6 q4 per instance -- rows of the world -> object matrix, then an optional
  // material (base rgb, roughness | emission rgb, ior | transmission, present flag).  The top-level tree's leaf items
  // are instance records (pt_bvh.h) and the meshes' trees follow it in `nodes`; primitives are in object space.
  const q4* inst_records;
  uint32_t num_tris;         // primitives (shade records)
  uint32_t num_items;        // traversal records
  float vmag;                // largest |coordinate| of any vertex (error bound of the fast triangle test)
  // rectangle light used for next-event estimation (optix_wrapper.cpp:842-845)
  float light_pos[3], light_emit[3], light_normal[3], light_half[2];
};

struct CameraView {
  float eye[3], u[3], v[3], w[3];
};

}  // namespace pt
